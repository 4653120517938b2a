// membench.hip -- what HBM bandwidth do the BP kernel's two access patterns reach as a function of the
// contiguous segment size (= 8 bytes x syndromes per tile)?  Standalone diagnostic (not part of the library).
//   pattern S: per wave, read ROW contiguous segments of array A, write ROW contiguous segments of C  (check pass)
//   pattern G: per wave, read COL segments of C at pseudo-random edge positions of its tile, write them to A at
//              the same positions                                                                      (bit pass)
// Tiles are [E][SEG bytes] regions; one workgroup per tile at a time, like the real kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)

template <int VEC>  // doubles per lane: 1 -> 512 B segments, 2 -> 1 KiB, 4 -> 2 KiB
struct V { double v[VEC]; };

template <int VEC, bool GATHER, int DEPTH>
__global__ void __launch_bounds__(512) k(const double* __restrict__ src, double* __restrict__ dst, int E, int iters) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const size_t tile = blockIdx.x;
    const size_t seg = 64 * VEC;  // doubles per segment
    const V<VEC>* s = reinterpret_cast<const V<VEC>*>(src + tile * (size_t)E * seg) + lane;
    V<VEC>* d = reinterpret_cast<V<VEC>*>(dst + tile * (size_t)E * seg) + lane;
    constexpr int GRP = GATHER ? 3 : 6;
    for (int it = 0; it < iters; ++it) {
        for (int g0 = wave * DEPTH; g0 * GRP < E; g0 += nw * DEPTH) {
            V<VEC> buf[DEPTH][GRP];
            int pos[DEPTH][GRP];
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
#pragma unroll
                for (int q = 0; q < GRP; ++q) {
                    int e = (g0 + u) * GRP + q;
                    if (GATHER) e = (int)(((unsigned)e * 2654435761u + (unsigned)it * 40503u) % (unsigned)E);
                    pos[u][q] = e < E ? e : 0;
                    buf[u][q] = s[(size_t)pos[u][q] * 64];
                }
#pragma unroll
            for (int u = 0; u < DEPTH; ++u)
#pragma unroll
                for (int q = 0; q < GRP; ++q) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) buf[u][q].v[v] += 1.0;
                    d[(size_t)pos[u][q] * 64] = buf[u][q];
                }
        }
        __syncthreads();
    }
}

template <int VEC, bool GATHER, int DEPTH>
void run(double* a, double* c, size_t total_doubles, int E, int waves, int iters) {
    const size_t seg = 64 * VEC;
    const int tiles = (int)(total_doubles / ((size_t)E * seg));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<VEC, GATHER, DEPTH>), dim3(tiles), dim3(waves * 64), 0, 0, a, c, E, 1);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<VEC, GATHER, DEPTH>), dim3(tiles), dim3(waves * 64), 0, 0, a, c, E, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double bytes = 2.0 * 8.0 * (double)tiles * E * seg * iters;
    printf("seg %4zu B  %s depth %d waves %2d tiles %5d : %8.1f GB/s\n", seg * 8, GATHER ? "gather/scatter" : "sequential    ", DEPTH, waves, tiles, bytes / ms / 1e6);
}

int main() {
    const size_t total = (size_t)2 << 30;  // doubles per array: 16 GiB each
    double *a, *c;
    CK(hipMalloc(&a, total * 8)); CK(hipMalloc(&c, total * 8));
    CK(hipMemset(a, 0, total * 8)); CK(hipMemset(c, 0, total * 8));
    const int E = 30000, it = 6;
    for (int waves : {4, 8}) {
        run<1, false, 1>(a, c, total, E, waves, it); run<1, false, 2>(a, c, total, E, waves, it);
        run<2, false, 1>(a, c, total, E, waves, it); run<2, false, 2>(a, c, total, E, waves, it);
        run<4, false, 1>(a, c, total, E, waves, it);
        run<1, true, 1>(a, c, total, E, waves, it); run<1, true, 2>(a, c, total, E, waves, it); run<1, true, 4>(a, c, total, E, waves, it);
        run<2, true, 1>(a, c, total, E, waves, it); run<2, true, 2>(a, c, total, E, waves, it); run<2, true, 4>(a, c, total, E, waves, it);
        run<4, true, 1>(a, c, total, E, waves, it); run<4, true, 2>(a, c, total, E, waves, it);
    }
    return 0;
}
