// bp_flood_lane_kernel.h -- the flooding schedule (bp.hpp:192-325) for the rows a streamed first pass left: a workgroup per syndrome
// Part of libldpc_hip.so (translation unit tu_stream.hip).
#pragma once

#include "bp_device_common.h"

// A 64-syndrome tile moves 64 lanes' messages whatever the number of lanes still decoding.  After the first pass of a two-pass decode
// (host_stream.h: decode_stream_repacked) the rows still running are a minority that thins out fast -- at the headline's early-exit point
// 16.7 % after iteration 7, 1.9 % after 8, 0.2 % after 9 -- yet, compacted into dense tiles, they keep EVERY tile of the second pass alive
// for three more rounds (a tile with one live lane moves all 64), and the one hopeless syndrome of the batch keeps its tile in the
// per-pass rounds for 40 more.  Here each such row is decoded by itself: a workgroup takes a row, copies its bit->check messages out of
// the first pass's tile (lane by lane) into a row-major array [nnz] that lives in L2, and runs the iterations with lane = node -- check
// pass (lane = check), bit pass (lane = bit), syndrome test, two workgroup barriers an iteration -- through the SAME per-node routines
// as the tile kernels (check_row / check_row_streamed / bit_column on a plain-array buffer type): same operations, same order, same
// bits.  A row that converges leaves at once; the hopeless one costs max_iter x ~35 us.  Rows and their count are the device's
// (the first pass's list of unconverged rows): nothing waits for the host.
struct FloodLaneArgs {
    int32_t m, n, nnz, max_iter, it_start;
    double ms_scaling_factor;
    const int32_t *row_ptr, *col_idx, *col_ptr, *csc_edge;
    const double *llr0;
    const double *A_tiles;      // [tiles][nnz][64] tanh(b2c / 2) | b2c after it_start iterations (the first pass's array)
    const int32_t *pos;         // a row's place in A_tiles' tiles (tile pos / 64, lane pos % 64), by the row's own number; nullptr: the number itself
    const int32_t *rows;        // the rows still decoding: their numbers in the caller's arrays
    const unsigned *count_dev;  // [0]: how many
    double *A, *C;              // [gridDim.x][nnz] each: a workgroup's row-major message arrays
    const uint8_t *synd;        // the caller's [batch][m]
    uint8_t *decoding;          // the caller's arrays, indexed by the rows' own numbers
    double *llr;
    int32_t *iters;
    uint8_t *conv;
};

struct RowMsgBuf {  // the message-array interface of the per-node routines (MsgBufT) on a plain row-major array
    double *p;
    __device__ __forceinline__ double ld(int, int edge) const { return p[edge]; }
    __device__ __forceinline__ void st(int, int edge, double x) const { p[edge] = x; }
};

template <int METHOD, int MATH, int DR, int DC>
__global__ void __launch_bounds__(512) bp_flood_lane_kernel(const FloodLaneArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fl_lds[];
    uint8_t *hard = fl_lds;                        // [n] this iteration's hard decisions
    uint8_t *sy = fl_lds + ((a.n + 15) & ~15);     // [m] syndrome bytes
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    const int tid = threadIdx.x, T = blockDim.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = tid; q < 256; q += T) log_tab[q] = ldpc_math::k_log_tab[q];
    const RowMsgBuf Ab{a.A + (size_t)blockIdx.x * (size_t)nnz}, Cb{a.C + (size_t)blockIdx.x * (size_t)nnz};
    const int64_t count = (int64_t)a.count_dev[0];
    for (int64_t r = blockIdx.x; r < count; r += gridDim.x) {
        const int64_t b = a.rows[r];
        __syncthreads();  // (the previous row's arrays and flags are done with)
        {
            const int64_t q = a.pos ? (int64_t)a.pos[b] : b;
            const double *from = a.A_tiles + ((size_t)(q >> 6) * (size_t)nnz) * LDPC_WAVE + (size_t)(q & 63);
            for (int e = tid; e < nnz; e += T) Ab.p[e] = from[(size_t)e * LDPC_WAVE];
            for (int i = tid; i < m; i += T) sy[i] = a.synd[b * m + i];
            for (int j = tid; j < n; j += T) hard[j] = 0;
        }
        __syncthreads();
        int it = a.it_start;
        bool converged = false;
        while (it < a.max_iter) {
            ++it;
            const double alpha = METHOD == LDPC_HIP_MINIMUM_SUM ? ((a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor) : 0.0;
            // ---- check pass (bp.hpp:201-273): lane = check ----
            for (int i = tid; i < m; i += T) {
                const int rs = a.row_ptr[i], d = a.row_ptr[i + 1] - rs;
                const bool neg = sy[i] != 0;          // bp.hpp:213
                const int parity = sy[i] & 1;         // bp.hpp:236
                if (d <= DR) {
                    double cur[DR];
#pragma unroll
                    for (int k = 0; k < DR; ++k)
                        if (k < d) cur[k] = Ab.p[rs + k];
                    check_row<METHOD, MATH, DR, RowMsgBuf>(cur, d, rs, neg, parity, alpha, Cb, 0, log_tab);
                } else {
                    check_row_streamed<METHOD, MATH, RowMsgBuf>(d, rs, neg, parity, alpha, Ab, Cb, 0, log_tab);
                }
            }
            __syncthreads();
            // ---- bit pass (bp.hpp:276-298 and 311-318): lane = bit ----
            for (int j = tid; j < n; j += T) {
                const int cs = a.col_ptr[j], d = a.col_ptr[j + 1] - cs;
                const double prior = a.llr0[j];
                double llr;
                if (d <= DC) {
                    int e[DC];
                    double c[DC];
#pragma unroll
                    for (int k = 0; k < DC; ++k)
                        if (k < d) { e[k] = a.csc_edge[cs + k]; c[k] = Cb.p[e[k]]; }
                    llr = bit_column<METHOD, MATH, DC, RowMsgBuf>(c, e, d, prior, Ab, 0, true);
                } else {  // the reference's two sweeps through memory
                    double temp = prior;
                    for (int k = 0; k < d; ++k) {
                        const int ee = a.csc_edge[cs + k];
                        Ab.p[ee] = temp;
                        temp += Cb.p[ee];
                    }
                    llr = temp;
                    double sfx = 0.0;
                    for (int k = d - 1; k >= 0; --k) {
                        const int ee = a.csc_edge[cs + k];
                        Ab.p[ee] = edge_form<METHOD, MATH>(Ab.p[ee] + sfx);
                        sfx += Cb.p[ee];
                    }
                }
                hard[j] = llr <= 0 ? 1 : 0;  // bp.hpp:290
                if (a.llr) a.llr[b * n + j] = llr;
            }
            __syncthreads();
            // ---- syndrome test (bp.hpp:292-294, 300-308): a byte > 1 never matches ----
            int bad = 0;
            for (int i = tid; i < m; i += T) {
                int cand = 0;
                for (int g = a.row_ptr[i]; g < a.row_ptr[i + 1]; ++g) cand ^= hard[a.col_idx[g]];
                bad |= cand != (int)sy[i];
            }
            if (!__syncthreads_or(bad)) { converged = true; break; }
        }
        for (int j = tid; j < n; j += T) a.decoding[b * n + j] = hard[j];
        if (tid == 0) {
            if (a.iters) a.iters[b] = converged ? it : a.max_iter;
            if (a.conv) a.conv[b] = converged ? 1 : 0;
        }
    }
}
