// overlap.hip -- does deeper prefetch help when each 6-segment row also carries ~N dependent FP64 FMAs per
// element (the product-sum kernel's situation)?  Sequential pattern, 512 B segments, 4 waves / workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int DEPTH>
__global__ void __launch_bounds__(256) k(const double* __restrict__ src, double* __restrict__ dst, int rows, int iters, int fmas) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t tile = blockIdx.x;
    const double* s = src + tile * (size_t)rows * 6 * 64 + lane;
    double* d = dst + tile * (size_t)rows * 6 * 64 + lane;
    for (int it = 0; it < iters; ++it) {
        double buf[DEPTH][6];
#pragma unroll
        for (int p = 0; p < DEPTH - 1; ++p) {
            int r = wave + p * nw;
#pragma unroll
            for (int q = 0; q < 6; ++q) buf[p][q] = r < rows ? s[((size_t)r * 6 + q) * 64] : 0.0;
        }
        int slot = 0;
        for (int r = wave; r < rows; r += nw) {
            // prefetch row r + (DEPTH-1)*nw into the free slot
            const int rp = r + (DEPTH - 1) * nw;
            const int ps = (slot + DEPTH - 1) % DEPTH;
#pragma unroll
            for (int p = 0; p < DEPTH; ++p)
                if (p == ps) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) buf[p][q] = rp < rows ? s[((size_t)rp * 6 + q) * 64] : 0.0;
                }
#pragma unroll
            for (int p = 0; p < DEPTH; ++p)
                if (p == slot) {
#pragma unroll
                    for (int q = 0; q < 6; ++q) {
                        double x = buf[p][q];
                        for (int f = 0; f < fmas; ++f) x = __builtin_fma(x, 0.999999, 1e-9);
                        d[((size_t)r * 6 + q) * 64] = x;
                    }
                }
            slot = (slot + 1) % DEPTH;
        }
        __syncthreads();
    }
}

template <int DEPTH>
void run(double* a, double* c, int tiles, int rows, int iters, int fmas) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<DEPTH>, dim3(tiles), dim3(256), 0, 0, a, c, rows, 1, fmas);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<DEPTH>, dim3(tiles), dim3(256), 0, 0, a, c, rows, iters, fmas);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bytes = 2.0 * 8.0 * (double)tiles * rows * 6 * 64 * iters;
    printf("fmas/elem %4d  prefetch depth %d : %8.1f GB/s   (%.1f ms)\n", fmas, DEPTH, bytes / ms / 1e6, ms);
}

int main() {
    const int tiles = 1024, rows = 5000, iters = 4;
    const size_t n = (size_t)tiles * rows * 6 * 64;
    double *a, *c; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&c, n * 8));
    CK(hipMemset(a, 0, n * 8)); CK(hipMemset(c, 0, n * 8));
    for (int f : {0, 60, 120, 180, 240}) { run<1>(a, c, tiles, rows, iters, f); run<2>(a, c, tiles, rows, iters, f); run<3>(a, c, tiles, rows, iters, f); run<4>(a, c, tiles, rows, iters, f); }
    return 0;
}
