// How fast does the chip start workgroups? N workgroups of T threads that each live about L microseconds (s_sleep loop,
// no memory traffic): kernel time vs N * L / slots.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/dispatch_rate tools/dispatch_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out, int ticks) {   // ticks of 10 ns
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
}
int main() {
    unsigned long long* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int threads : {64, 256}) for (int life_us : {0, 2, 10, 40}) for (int n : {32400, 129600}) {
        const int blocks = n * 64 / threads;
        k<<<blocks, threads>>>(out, life_us * 100);
        hipEventRecord(e0);
        k<<<blocks, threads>>>(out, life_us * 100);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%6d waves as %6d x %3d threads, life %2d us: %8.1f us -> %6.1f waves/us started; ideal at 8192 slots %7.1f us\n", n, blocks, threads, life_us,
               ms * 1e3, n / (ms * 1e3), double(n) * life_us / 8192.0);
    }
    return 0;
}
