// dma_test.hip -- validates the LDS-DMA idiom the BP kernel relies on (gfx950):
//   buffer_load_dwordx4 ... offen lds : lane l's 16 bytes land at LDS[m0 + 16*l]; counted vmcnt waits.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("%s: %s\n",#x,hipGetErrorString(e)); exit(1);} }while(0)
extern __shared__ __attribute__((aligned(16))) unsigned char dyn_lds[];
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory");
}
// each wave: rows of 6 segments (512 B each = 64 doubles); ring of 3 slots; out[row][k][lane] = in[row][k][lane] + 1
__global__ void __launch_bounds__(256) k(const double* A, double* C, int rows) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nw = blockDim.x >> 6;
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, rows * 6 * 512, 0x00020000);
    const unsigned ring = (unsigned)(uintptr_t)dyn_lds + wave * 3 * 3072;
    const double* ringp = (const double*)(dyn_lds + wave * 3 * 3072);
    auto issue = [&](int row, int slot) {
        for (int c = 0; c < 3; ++c) dma16(ra, lane * 16, (unsigned)(row * 6 + 2 * c) * 512u, ring + slot * 3072 + c * 1024);
    };
    int nrows = 0;
    for (int r = wave; r < rows; r += nw) nrows++;
    for (int p = 0; p < 3 && p < nrows; ++p) issue(wave + p * nw, p);
    for (int idx = 0; idx < nrows; ++idx) {
        const int r = wave + idx * nw;
        const int slot = idx % 3;
        const int later = (nrows - 1 - idx) < 2 ? (nrows - 1 - idx) : 2;  // DMA rows issued after this one
        // conservative exact count: DMAs of later rows (3 each) + stores issued after DMA(r)
        // here: simply wait by case
        if (idx >= 3 && idx + 2 < nrows) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        double v[6];
        for (int q = 0; q < 6; ++q) v[q] = ringp[slot * 384 + q * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (idx + 3 < nrows) issue(r + 3 * nw, slot);
        for (int q = 0; q < 6; ++q) C[((size_t)r * 6 + q) * 64 + lane] = v[q] + 1.0;
        (void)later;
    }
}
int main() {
    const int rows = 5000; const size_t n = (size_t)rows * 6 * 64;
    std::vector<double> h(n); for (size_t i = 0; i < n; ++i) h[i] = (double)i;
    double *a, *c; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&c, n * 8));
    CK(hipMemcpy(a, h.data(), n * 8, hipMemcpyHostToDevice)); CK(hipMemset(c, 0, n * 8));
    for (int trial = 0; trial < 3; ++trial) {
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 4 * 3 * 3072, 0, a, c, rows);
        CK(hipDeviceSynchronize());
        std::vector<double> o(n); CK(hipMemcpy(o.data(), c, n * 8, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0; for (size_t i = 0; i < n; ++i) if (o[i] != h[i] + 1.0) { if (!bad) first = i; bad++; }
        printf("trial %d: %zu mismatches of %zu%s\n", trial, bad, n, bad ? "" : "  PASS");
        if (bad) printf("  first at %zu: got %.1f want %.1f (row %zu seg %zu lane %zu)\n", first, o[first], h[first] + 1, first / 384, (first / 64) % 6, first % 64);
    }
    return 0;
}
