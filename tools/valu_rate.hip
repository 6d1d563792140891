// Issue rate of the VALU instruction classes the traversal kernels are made of, per SIMD, with W waves per SIMD:
// every wave runs ITER x 8 independent dependent-chains of one instruction. Prints shader cycles per wave64 instruction
// per SIMD (s_memtime ticks of one wave / instructions issued by all W waves of its SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/valu_rate tools/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define ITER 4096
template <int OP>
__global__ __launch_bounds__(64) void k(float* out, unsigned long long* ticks, float a, float b) {
    float x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7;
    const double da = 1.0000001, db = 0.5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITER; ++i) {
#define STEP(x)                                                                                                  \
        if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));                         \
        else if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(a));                                \
        else if (OP == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(b));                                \
        else if (OP == 3) asm volatile("v_min_f32 %0, %0, %1" : "+v"(x) : "v"(b));                                \
        else if (OP == 4) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b) : "vcc");  \
        else if (OP == 5) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x));          \
        else if (OP == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(x));                                             \
        else if (OP == 8) asm volatile("v_readlane_b32 s20, %0, 3\n v_mov_b32 %0, s20" : "+v"(x) : : "s20");              \
        else if (OP == 9) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));                     \
        else if (OP == 10) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(x), "v"(b) : "vcc");                     \
        else if (OP == 11) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(b));                        \
        else if (OP == 12) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(a), "v"(b));                     \
        else if (OP == 13) asm volatile("v_writelane_b32 %0, %1, 3" : "+v"(x) : "s"(a));                            \
        else if (OP == 14) asm volatile("v_sqrt_f32 %0, %0" : "+v"(x));                                             \
        else if (OP == 15) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(x) : "v"(b) : "vcc");          \
        else if (OP == 16) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));                    \
        else if (OP == 17) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(b));                                 \
        else if (OP == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(x) : "v"(b));                                 \
        else if (OP == 19) asm volatile("v_mov_b32 %0, %1" : "+v"(x) : "v"(b));                                     \
        else if (OP == 20) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x) : "v"(b));                                 \
        else if (OP == 21) asm volatile("v_fma_f32 %0, %0, %1, %2\n s_cselect_b32 s20, s21, s22\n s_cselect_b32 s23, s22, s21" : "+v"(x) : "v"(a), "v"(b) : "s20", "s23");
        if (OP < 22) { STEP(x0) STEP(x1) STEP(x2) STEP(x3) STEP(x4) STEP(x5) STEP(x6) STEP(x7) }
        else {
#define STEP2(d)                                                                                                 \
            if (OP == 22) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(d) : "v"(da), "v"(db));                  \
            else if (OP == 23) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d) : "v"(da), "v"(db));                \
            else if (OP == 24) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d) : "v"(da));                             \
            else if (OP == 25) asm volatile("v_rsq_f64 %0, %0" : "+v"(d));                                           \
            else if (OP == 26) { float f_; asm volatile("v_cvt_f32_f64 %0, %1\n v_cvt_f64_f32 %1, %0" : "=&v"(f_), "+v"(d)); }
            STEP2(d0) STEP2(d1) STEP2(d2) STEP2(d3) STEP2(d4) STEP2(d5) STEP2(d6) STEP2(d7)
        }
    }
    x0 += (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 64 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int OP> void run(const char* name, int inst_per_step, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&ticks, blocks * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 64>>>(out, ticks, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<OP><<<blocks, 64>>>(out, ticks, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), ticks, blocks * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = double(h[blocks / 2]);
    const double inst_wave = double(ITER) * 8 * inst_per_step;
    printf("%-22s W=%d  %.2f ms  ticks/wave %.0f  -> %.2f shader cycles per wave-instruction per SIMD (wall: %.2f at 2.4 GHz)\n", name, waves_per_simd, ms,
           med, med / (inst_wave * waves_per_simd), ms * 1e-3 * 2.4e9 / (inst_wave * waves_per_simd));
    hipFree(out); hipFree(ticks);
}
#include <algorithm>
int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", 1, w); run<1>("v_mul_f32", 1, w); run<2>("v_add_f32", 1, w); run<3>("v_min_f32", 1, w);
        run<4>("v_cmp+v_cndmask", 2, w); run<5>("v_mov_dpp quad_perm", 1, w); run<6>("v_rcp_f32", 1, w); run<8>("v_readlane+v_mov", 2, w);
        run<9>("v_min3_f32", 1, w); run<10>("v_cmp_lt_f32", 1, w); run<11>("v_cndmask_b32", 1, w); run<12>("v_fma_f32 (sgpr src)", 1, w);
        run<13>("v_writelane_b32", 1, w); run<14>("v_sqrt_f32", 1, w); run<15>("v_div_scale_f32", 1, w); run<16>("v_med3_f32", 1, w);
        run<17>("v_max_f32", 1, w); run<18>("v_and_b32", 1, w); run<19>("v_mov_b32", 1, w); run<20>("v_sub_f32", 1, w);
        run<21>("v_fma + 2 s_cselect", 1, w); run<22>("v_pk_fma_f32", 1, w); run<23>("v_fma_f64", 1, w); run<24>("v_mul_f64", 1, w);
        run<25>("v_rsq_f64", 1, w); run<26>("v_cvt_f64_f32+back", 2, w);
    }
    return 0;
}
