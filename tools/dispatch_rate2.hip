// Does a mix of short and long waves keep the chip full? 32,400 single-wave workgroups; a share of them (spread evenly,
// or clustered in one range of block ids like the blocks of a frame that cover the mesh) live 10x longer.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(64) void k(unsigned long long* out, int short_ticks, int long_ticks, int mode) {
    const bool heavy = mode == 0 ? false : mode == 1 ? (blockIdx.x % 10 == 0) : (blockIdx.x >= 14000 && blockIdx.x < 14000 + 3240);
    const int ticks = heavy ? long_ticks : short_ticks;
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t0;
}
int main() {
    unsigned long long* out; hipMalloc(&out, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int n = 32400;
    for (int mode : {0, 1, 2}) {
        const int s = 600, l = 6000;      // 6 us and 60 us
        k<<<n, 64>>>(out, s, l, mode);
        hipEventRecord(e0);
        k<<<n, 64>>>(out, s, l, mode);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double total = mode == 0 ? n * 6.0 : (n * 0.9 * 6.0 + n * 0.1 * 60.0);
        printf("mode %d (%s): %7.1f us; ideal at 8192 slots %6.1f us\n", mode, mode == 0 ? "all 6 us" : mode == 1 ? "every 10th wave 60 us" : "3240 consecutive waves 60 us", ms * 1e3, total / 8192.0);
    }
    return 0;
}
