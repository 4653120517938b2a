// segcopy.hip -- what HBM gives the message traffic of the streamed BP kernels, pattern by pattern.
// A "tile" owns nseg segments of 512 bytes (64 syndromes x one double: one edge of the code); a workgroup per tile copies its
// segments src -> dst, 8 bytes per lane per access like the kernels do, reading / writing them in sequence or through a
// permutation (gather / scatter), singly or in runs of `run` neighbouring segments (a column's or a row's entries stored together).
// Build: hipcc -O3 --offload-arch=gfx950 -o segcopy segcopy.hip      Run on an MI355X: ./segcopy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int U>
__global__ void __launch_bounds__(768) segcopy_kernel(const double *src, double *dst, const int *__restrict__ rmap, const int *__restrict__ wmap, int nseg, int passes) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nwaves = blockDim.x >> 6;  // (maps through the scalar cache)
    const size_t base = (size_t)blockIdx.x * (size_t)nseg * 64;
    for (int p = 0; p < passes; ++p) {
        const double *s = (p & 1) ? dst : src;
        double *d = (p & 1) ? const_cast<double *>(src) : dst;
        for (int e0 = wave * U; e0 < nseg; e0 += nwaves * U) {
            double v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int e = e0 + u < nseg ? e0 + u : nseg - 1;
                const int r = rmap ? rmap[e] : e;
                v[u] = __builtin_nontemporal_load(s + base + (size_t)r * 64 + lane);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (e0 + u >= nseg) break;
                const int w = wmap ? wmap[e0 + u] : e0 + u;
                __builtin_nontemporal_store(v[u], d + base + (size_t)w * 64 + lane);
            }
        }
        __syncthreads();
    }
}

static std::vector<int> make_map(int nseg, int run, unsigned seed) {  // runs of `run` neighbouring segments, the runs shuffled
    const int nruns = nseg / run;
    std::vector<int> order(nruns);
    std::iota(order.begin(), order.end(), 0);
    std::mt19937 g(seed);
    std::shuffle(order.begin(), order.end(), g);
    std::vector<int> m((size_t)nseg);
    for (int q = 0; q < nruns; ++q)
        for (int k = 0; k < run; ++k) m[(size_t)q * run + k] = order[q] * run + k;
    for (int e = nruns * run; e < nseg; ++e) m[(size_t)e] = e;
    return m;
}

int main(int argc, char **argv) {
    const int tiles = argc > 1 ? atoi(argv[1]) : 512, nseg = argc > 2 ? atoi(argv[2]) : 30000, passes = argc > 3 ? atoi(argv[3]) : 8;
    const bool quick = argc > 4;  // only sequential / sequential and scattered / scattered
    const size_t bytes = (size_t)tiles * nseg * 512;
    double *a, *b;
    CHK(hipMalloc(&a, bytes));
    CHK(hipMalloc(&b, bytes));
    CHK(hipMemset(a, 1, bytes));
    CHK(hipMemset(b, 2, bytes));
    int *maps[4];  // 0: none; single segments; runs of 3; runs of 6
    const int runs[4] = {0, 1, 3, 6};
    for (int q = 1; q < 4; ++q) {
        std::vector<int> m = make_map(nseg, runs[q], 17u + q);
        CHK(hipMalloc(&maps[q], sizeof(int) * (size_t)nseg));
        CHK(hipMemcpy(maps[q], m.data(), sizeof(int) * (size_t)nseg, hipMemcpyHostToDevice));
    }
    maps[0] = nullptr;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    printf("{\"tiles\": %d, \"segments_per_tile\": %d, \"bytes_per_array\": %zu, \"passes\": %d}\n", tiles, nseg, bytes, passes);
    const char *names[4] = {"sequential", "scattered 512 B", "scattered runs of 3 (1.5 KiB)", "scattered runs of 6 (3 KiB)"};
    for (int unroll : {6, 12})
        for (int r = 0; r < 4; ++r)
            for (int w = 0; w < 4; ++w) {
                if (quick && (unroll != 6 || r != w || r > 1)) continue;
                float best = 1e30f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHK(hipEventRecord(e0));
                    if (unroll == 6) hipLaunchKernelGGL(segcopy_kernel<6>, dim3(tiles), dim3(768), 0, 0, a, b, maps[r], maps[w], nseg, passes);
                    else hipLaunchKernelGGL(segcopy_kernel<12>, dim3(tiles), dim3(768), 0, 0, a, b, maps[r], maps[w], nseg, passes);
                    CHK(hipGetLastError());
                    CHK(hipEventRecord(e1));
                    CHK(hipEventSynchronize(e1));
                    float ms;
                    CHK(hipEventElapsedTime(&ms, e0, e1));
                    best = std::min(best, ms);
                }
                const double tbps = 2.0 * (double)bytes * passes / (best * 1e-3) / 1e12;
                printf("{\"segments_in_flight_per_wavefront\": %d, \"read\": \"%s\", \"write\": \"%s\", \"ms\": %.3f, \"TB_per_s\": %.3f}\n", unroll, names[r], names[w], best, tbps);
                fflush(stdout);
            }
    return 0;
}
