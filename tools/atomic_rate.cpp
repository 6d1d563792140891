// How fast are same-address returning atomics from many waves? (one atomic per wave, like the level appends)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_atomic(unsigned* ctr, unsigned* out, int shards, int stride_words, int work) {
    const unsigned wave = blockIdx.x;
    float x = float(threadIdx.x);
    for (int i = 0; i < work; ++i) x = x * 1.0001f + 0.5f;       // some independent work per wave
    unsigned base = 0;
    if (threadIdx.x == 0) base = atomicAdd(&ctr[(wave % shards) * stride_words], 64u);
    base = __shfl(base, 0);
    if (x == 12345.678f) out[0] = 1;
    out[(size_t(wave) * 64 + threadIdx.x)] = base + threadIdx.x;   // dependent store, like the append
}
__global__ void k_plain(unsigned* ctr, unsigned* out, int work) {
    const unsigned wave = blockIdx.x;
    float x = float(threadIdx.x);
    for (int i = 0; i < work; ++i) x = x * 1.0001f + 0.5f;
    if (x == 12345.678f) out[0] = 1;
    out[(size_t(wave) * 64 + threadIdx.x)] = wave * 64 + threadIdx.x;
}
int main() {
    const int waves = 65536;
    unsigned *ctr, *out;
    hipMalloc(&ctr, 1 << 20); hipMalloc(&out, size_t(waves) * 64 * 4);
    hipMemset(ctr, 0, 1 << 20);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    auto run = [&](const char* name, auto f) {
        f(); hipDeviceSynchronize();
        hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-40s %8.1f us per launch, %6.2f ns per wave\n", name, ms * 100, ms * 1e5 / waves);
    };
    run("no atomic", [&] { k_plain<<<waves, 64>>>(ctr, out, 0); });
    for (int shards : {1, 2, 8, 64, 1024}) {
        char nm[64]; snprintf(nm, 64, "atomic, %d counters (256 B apart)", shards);
        run(nm, [&] { k_atomic<<<waves, 64>>>(ctr, out, shards, 64, 0); });
    }
    run("atomic, 1 counter, 2000 fma of work", [&] { k_atomic<<<waves, 64>>>(ctr, out, 1, 64, 2000); });
    run("no atomic, 2000 fma of work", [&] { k_plain<<<waves, 64>>>(ctr, out, 2000); });
    return 0;
}
