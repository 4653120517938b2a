// valurate.hip -- issue cost of FP64 vector instructions on gfx950, in SIMD cycles per wavefront instruction:
// v_fma_f64, v_mul_f64, v_add_f64, v_rcp_f64, v_cvt_i32_f64, v_cndmask_b32, v_min_f64 ...  One wavefront per SIMD slot,
// WAVES wavefronts per SIMD, each running a loop of 16 independent instructions of one kind.
// Build: hipcc -O3 --offload-arch=gfx950 -o valurate valurate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(double *out, int iters, double seed) {
    double v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = seed + q * 0.125 + threadIdx.x * 1e-3;
    const double c = seed * 0.5 + 1.0, d = seed * 0.25 + 0.5;
    int w[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) w[q] = threadIdx.x + q;
    const int wc = iters;
    for (int it = 0; it < iters; ++it) {
#define FMA(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[q]) : "v"(c), "v"(d));
#define MUL(q) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[q]) : "v"(c));
#define ADD(q) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[q]) : "v"(c));
#define RCP(q) asm volatile("v_rcp_f64 %0, %0" : "+v"(v[q]));
#define MIN(q) asm volatile("v_min_f64 %0, %0, %1" : "+v"(v[q]) : "v"(c));
#define CND(q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[q]) : "v"(wc));
#define CVT(q) { int t_; asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(t_) : "v"(v[q])); asm volatile("" :: "v"(t_)); }
#define CMP(q) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(v[q]), "v"(c) : "vcc");
#define RCP32(q) { float f_ = (float)q + 1.5f; asm volatile("v_rcp_f32 %0, %0" : "+v"(f_)); asm volatile("" :: "v"(f_)); }
#define LDEXP(q) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(v[q]));
// DEPENDENT chains (round 6): every instruction reads the result of the one before it -- what a wavefront that has its SIMD to itself pays
// per instruction of an exact tanh / log evaluation; two / four chains interleaved
#define DFMA(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[0]) : "v"(c), "v"(d));
#define DFMA2(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[(q) & 1]) : "v"(c), "v"(d));
#define DFMA4(q) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[(q) & 3]) : "v"(c), "v"(d));
#define DADD(q) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[0]) : "v"(c));
#define DCND(q) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(w[0]) : "v"(wc));
        if (KIND == 0) { REP16(FMA) }
        if (KIND == 1) { REP16(MUL) }
        if (KIND == 2) { REP16(ADD) }
        if (KIND == 3) { REP16(RCP) }
        if (KIND == 4) { REP16(MIN) }
        if (KIND == 5) { REP16(CND) }
        if (KIND == 6) { REP16(CVT) }
        if (KIND == 7) { REP16(CMP) }
        if (KIND == 8) { REP16(RCP32) }
        if (KIND == 9) { REP16(LDEXP) }
        if (KIND == 10) { REP16(DFMA) }
        if (KIND == 11) { REP16(DFMA2) }
        if (KIND == 12) { REP16(DFMA4) }
        if (KIND == 13) { REP16(DADD) }
        if (KIND == 14) { REP16(DCND) }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += v[q] + w[q];
    if (s == 12345.678) out[0] = s;
}

int main() {
    double *out;
    CHK(hipMalloc(&out, 8));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    const char *names[15] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_min_f64", "v_cndmask_b32", "v_cvt_i32_f64", "v_cmp_lt_f64", "v_rcp_f32", "v_ldexp_f64",
                             "v_fma_f64 dependent chain", "v_fma_f64 two dependent chains interleaved", "v_fma_f64 four dependent chains interleaved",
                             "v_add_f64 dependent chain", "v_cndmask_b32 dependent chain"};
    void (*kerns[15])(double *, int, double) = {rate_kernel<0>, rate_kernel<1>, rate_kernel<2>, rate_kernel<3>, rate_kernel<4>, rate_kernel<5>, rate_kernel<6>, rate_kernel<7>, rate_kernel<8>, rate_kernel<9>,
                                                rate_kernel<10>, rate_kernel<11>, rate_kernel<12>, rate_kernel<13>, rate_kernel<14>};
    const int iters = 20000;
    for (int waves_per_simd : {1, 2, 4}) {
        for (int k = 0; k < 15; ++k) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHK(hipEventRecord(e0));
                hipLaunchKernelGGL(kerns[k], dim3(256 * waves_per_simd), dim3(256), 0, 0, out, iters, 1.0);  // 256 threads = one wavefront per SIMD of a CU
                CHK(hipEventRecord(e1));
                CHK(hipEventSynchronize(e1));
                float ms;
                CHK(hipEventElapsedTime(&ms, e0, e1));
                best = best < ms ? best : ms;
            }
            const double insts_per_simd = (double)iters * 16 * waves_per_simd;
            printf("{\"waves_per_simd\": %d, \"instruction\": \"%s\", \"ms\": %.3f, \"ns_per_wave_instruction_per_simd\": %.3f}\n", waves_per_simd, names[k], best, best * 1e6 / insts_per_simd);
            fflush(stdout);
        }
    }
    return 0;
}
