// osd_kernels.h -- ordered-statistics decoding of the rows BP left unconverged (osd.hpp:103-187)
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

#define OSD_PIECE 9  // blocked elimination of osd_big_kernel: planes per table round
// workgroup barrier that orders LDS traffic only: global loads and stores in flight stay in flight (__syncthreads() waits for them)
#define OSD_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define OSD_BLOCK_ROWS 2048  // blocked elimination: rows the workgroup keeps in registers (eight per thread)
#define OSD_TRIP 16  // osd_big_kernel: columns per trip when a candidate is weighed

#ifdef LDPC_HIP_OSD_CLOCKS  // profiling aid (tools/osd_phase_clocks.py): cycles per phase of osdw_reg_kernel, summed over wavefronts
__device__ unsigned long long osd_phase_clocks[16];
#define OSD_CLK(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); \
                           if (lane == 0) atomicAdd(&osd_phase_clocks[slot], now_ - clk_); clk_ = now_; } while (0)
#define OSD_CLK_START() unsigned long long clk_ = __builtin_readcyclecounter()
#ifndef LDPC_HIP_OSD_CLOCK_MASK  // which osd_big_kernel probes are live (a probe serialises: few at a time perturb least); time goes to the next live probe
#define LDPC_HIP_OSD_CLOCK_MASK 0xffff
#endif
#define OSD_WG_CLK(slot) do { if ((LDPC_HIP_OSD_CLOCK_MASK >> (slot)) & 1) { const unsigned long long now_ = __builtin_readcyclecounter(); \
                              if (tid == 0) atomicAdd(&osd_phase_clocks[slot], now_ - clk_); clk_ = now_; } } while (0)
#else
#define OSD_CLK(slot) do { } while (0)
#define OSD_CLK_START() do { } while (0)
#define OSD_WG_CLK(slot) do { } while (0)
#endif

// ---- OSD-0 (osd.hpp:110-117 = sort.hpp:48-62 + gf2sparse_linalg.hpp:298-401, 237-288) -------------
// One wavefront per syndrome that BP left unconverged.  The reference sorts the columns by ascending
// log-ratio (glibc qsort: stable, so ties keep ascending index), runs a greedy column-ordered Gaussian
// elimination on a linked-list matrix until the syndrome is in the span of the pivots, and solves on
// the pivot columns.  That solution is unique given the column order (the reference's min-row-weight
// pivoting only picks which ROW carries a pivot), so here the augmented matrix [H | s] lives bit-packed
// in LDS (lane l owns rows l, l+64, ...), columns are visited in rank order and eliminated
// Gauss-Jordan style with wave ballots.  All LDS traffic is wave-private: no workgroup barriers.
struct OsdArgs {
    int32_t m, n, words;  // words = ceil((n + 1) / 64): n matrix bits + the syndrome bit per row
    int64_t batch;
    const int32_t *row_ptr, *col_idx;
    const uint64_t *packed;  // [m][words] H bit-packed by rows (bit n, the syndrome's place, clear): register kernels
    const uint8_t *synd;   // [batch][m]
    const double *llr;     // [batch][n]  BP posteriors
    const uint8_t *conv;   // [batch]     1 = BP converged: row left untouched
    uint8_t *decoding;     // [batch][n]  in: BP decisions, out: OSD solution for unconverged rows
    int32_t lds_per_wave;  // bytes
    int32_t method, order; // osdw_kernel: 2 = exhaustive (OSD_E), 3 = combination sweep (OSD_CS); order > 0
    int32_t kwords;        // osdw_reg_kernel: ceil((n - rank H) / 64), at least 1
    int32_t rank;          // register kernels: rank of H over GF(2)
    const double *wt;      // [n] log(1 / p_j): the weight of bit j in a candidate (osd.hpp:134, 173)
    const int32_t *list;   // rows BP left unconverged, any order (osd_collect_kernel)
    unsigned *counters;    // [0] number of entries of `list`, [1] next entry to hand out (both zeroed before the collect)
    const uint16_t *ell;   // osd0_flat_kernel: [m][8] the columns of a row's entries, 0xffff behind them (rows of up to eight entries)
    uint8_t *status;       // [batch] or nullptr: osd0_reg_kernel says itself what became of a row (1 solved, 2 outside the image: osd_status_kernel's
                           // answer, read off the eliminated matrix); nullptr in the second pass over corrected syndromes, whose rows keep their 2
};

// (osd_collect_kernel -- the list of rows BP left unconverged -- lives in io_kernels.h: the streamed and serial hosts use it too)

// What became of every row that went through OSD: status[b] = 1 if the returned x satisfies H x = s, 2 if it does not.
// The latter happens exactly when s lies outside the image of H (rank-deficient H with a syndrome that violates a
// dependency among the checks, e.g. toric-code X checks with a measurement error): no x solves such a system.  The
// reference then still returns a vector -- the solution of the subsystem formed by ITS pivot rows, and which rows those are
// is decided by the sparsity heuristic of its linked-list elimination (fewest entries across L and U, ties by current
// position; gf2sparse_linalg.hpp:327-340), which a bit-packed elimination does not re-enact.  The device returns the
// solution of the subsystem of its own pivot rows (first candidate row in sorted-column order) and says so here.
// One workgroup per listed row; the status array is zeroed beforehand (0 = BP converged, OSD not run).
__global__ void __launch_bounds__(256) osd_status_kernel(const OsdArgs a, uint8_t *status) {
    __shared__ int bad;
    const unsigned count = a.counters[0];
    for (unsigned r = blockIdx.x; r < count; r += gridDim.x) {
        const int64_t b = a.list[r];
        if (threadIdx.x == 0) bad = 0;
        __syncthreads();
        int mine = 0;
        for (int i = threadIdx.x; i < a.m; i += blockDim.x) {
            unsigned par = a.synd[b * a.m + i] ? 1u : 0u;  // a non-zero byte is a one (gf2sparse_linalg.hpp:309)
            for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) par ^= a.decoding[b * a.n + a.col_idx[e]] & 1u;
            mine |= (int)par;
        }
        if (mine) atomicOr(&bad, 1);
        __syncthreads();
        if (threadIdx.x == 0) status[b] = bad ? 2 : 1;
        __syncthreads();
    }
}

// Rows for the persistent workers (wavefronts, or workgroups in osd_big_kernel).  Worker g of W starts with entry g of the list --
// no atomic: a visit to the work counter costs ~1 us under load and one word serves ~88 of them per us, which for the few
// hundred listed rows of a small batch was most of the kernel (8 192 idle wavefronts queueing for one word: ~100 us) -- and
// workers beyond the list leave at once.  Entries W, W + 1, ... are handed out through the counter; none exist when W >= count.
__device__ __forceinline__ unsigned osd_list_count(const OsdArgs &a) {
    return __hip_atomic_load(&a.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ int64_t osd_first_row(const OsdArgs &a, unsigned worker) {
    return worker < osd_list_count(a) ? (int64_t)a.list[worker] : -1;
}
// next unconverged row for this wavefront, -1 when the list is exhausted
__device__ __forceinline__ int64_t osd_next_row(const OsdArgs &a, int lane, unsigned workers) {
    const unsigned count = osd_list_count(a);
    if (workers >= count) return -1;  // (wave-uniform)
    unsigned idx = 0;
    if (lane == 0) idx = atomicAdd(&a.counters[1], 1u);
    idx = workers + (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
    return idx < count ? (int64_t)a.list[idx] : -1;
}
#define OSD_WAVE_WORKER() (blockIdx.x * (blockDim.x >> 6) + (unsigned)wave)
#define OSD_WAVE_WORKERS() (gridDim.x * (blockDim.x >> 6))

__device__ __forceinline__ bool osd_less(double a, int ia, double b, int ib) {
    const bool na = a != a, nb = b != b;
    if (na || nb) return na == nb ? ia < ib : nb;  // numbers before NaNs (reference order undefined for NaN)
    if (a < b) return true;
    if (a > b) return false;
    return ia < ib;  // stable: ties in ascending index, as glibc's merge-sort qsort leaves them
}

__global__ void __launch_bounds__(256) osd0_kernel(const OsdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = a.m, n = a.n, W = a.words;
    unsigned char *base = osd_lds + (size_t)wave * a.lds_per_wave;
    volatile uint64_t *mat = reinterpret_cast<volatile uint64_t *>(base);                  // [m][W]
    volatile double *keys = reinterpret_cast<volatile double *>(base + (size_t)m * W * 8);  // [n]
    volatile int32_t *order = reinterpret_cast<volatile int32_t *>(base + (size_t)m * W * 8 + (size_t)n * 8);  // [n]
    volatile int32_t *pivot_col = order + n;                                                // [m]
    volatile uint8_t *x = reinterpret_cast<volatile uint8_t *>(const_cast<int32_t *>(pivot_col + m));  // [n]

    const int sw = n >> 6;
    const uint64_t sbit = 1ull << (n & 63);
    for (int64_t b = osd_first_row(a, OSD_WAVE_WORKER()); b >= 0; b = osd_next_row(a, lane, OSD_WAVE_WORKERS())) {
    for (int i = lane; i < m; i += 64) {
        for (int w = 0; w < W; ++w) mat[(size_t)i * W + w] = 0;
        for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) {
            const int c = a.col_idx[e];
            mat[(size_t)i * W + (c >> 6)] = mat[(size_t)i * W + (c >> 6)] | (1ull << (c & 63));
        }
        if (a.synd[b * m + i]) mat[(size_t)i * W + sw] = mat[(size_t)i * W + sw] | sbit;  // `if (i)`, gf2sparse_linalg.hpp:309
        pivot_col[i] = -1;
    }
    for (int j = lane; j < n; j += 64) { keys[j] = a.llr[b * n + j]; x[j] = 0; }
    __builtin_amdgcn_wave_barrier();
    // soft_decision_col_sort: rank of column i = number of columns that sort before it
    for (int i = lane; i < n; i += 64) {
        const double ki = keys[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += osd_less(keys[j], j, ki, i) ? 1 : 0;
        order[r] = i;
    }
    __builtin_amdgcn_wave_barrier();

    const int max_rank = m < n ? m : n;
    int rank = 0;
    for (int t = 0; t < n && rank < max_rank; ++t) {
        const int c = order[t];
        const int cw = c >> 6;
        const uint64_t cb = 1ull << (c & 63);
        // first unpivoted row with a one in column c
        int p = -1;
        for (int i0 = 0; i0 < m && p < 0; i0 += 64) {
            const int i = i0 + lane;
            const bool cand = i < m && pivot_col[i] < 0 && (mat[(size_t)i * W + cw] & cb);
            const uint64_t mask = __ballot(cand);
            if (mask) p = i0 + __builtin_ctzll(mask);
        }
        if (p < 0) continue;
        for (int i = lane; i < m; i += 64)
            if (i != p && (mat[(size_t)i * W + cw] & cb))
                for (int w = 0; w < W; ++w) mat[(size_t)i * W + w] = mat[(size_t)i * W + w] ^ mat[(size_t)p * W + w];
        if (lane == 0) pivot_col[p] = c;
        ++rank;
        __builtin_amdgcn_wave_barrier();
        // stop once the syndrome is in the span of the pivots (gf2sparse_linalg.hpp:373-383)
        bool pending = false;
        for (int i0 = 0; i0 < m && !pending; i0 += 64) {
            const int i = i0 + lane;
            pending = __ballot(i < m && pivot_col[i] < 0 && (mat[(size_t)i * W + sw] & sbit)) != 0;
        }
        if (!pending) break;
    }
    for (int i = lane; i < m; i += 64)
        if (pivot_col[i] >= 0 && (mat[(size_t)i * W + sw] & sbit)) x[pivot_col[i]] = 1;
    __builtin_amdgcn_wave_barrier();
    for (int j = lane; j < n; j += 64) a.decoding[b * n + j] = x[j];
    __builtin_amdgcn_wave_barrier();
    }  // next row
}

// ---- register-resident elimination for small matrices ------------------------------------------------------
// For m <= 64 R rows and n + 1 <= 64 W bits the augmented matrix of a syndrome fits in the wavefront's registers:
// lane l holds rows l, l + 64, ... (R of them, W words each).  A pivot step is then a ballot (first unpivoted row
// with the bit), W x v_readlane (broadcast of the pivot row) and predicated XORs -- no LDS traffic at all; only the
// column order goes through LDS once.  R and W are template bounds, so every register index is a compile-time one.
template <int R, int W>
struct OsdRows {
    uint64_t w[R][W];  // H (bit j = column j); bits >= n stay clear
    uint32_t s[R];     // the syndrome column of [H | s], kept apart so that no step has to index a word by n / 64
    uint64_t sm[R];    // the same column as wave-uniform lane masks -- the form the steps with SS = true keep up to date INSTEAD of s[]: its
                       // updates and the early-stop test are then scalar instructions that hang on no vector result (osd0_reg_kernel)
    int32_t pcol[R];   // pivot column carried by the row, -1: none
    uint64_t unp[R];   // wave-uniform: lanes whose row r carries no pivot yet
};

__device__ __forceinline__ uint64_t osd_readlane64(uint64_t v, int src_lane) {  // src_lane wave-uniform
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src_lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src_lane);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ double osd_readlane_f64(double v, int src_lane) {
    return __builtin_bit_cast(double, osd_readlane64(__builtin_bit_cast(uint64_t, v), src_lane));
}
// [H | s] of batch row b into registers (bit n of a row = its syndrome byte != 0, gf2sparse_linalg.hpp:309); H comes
// bit-packed (a.packed), so a row is W independent loads rather than a walk over its CSR entries
template <int R, int W>
__device__ __forceinline__ void osd_load_rows(const OsdArgs &a, int64_t b, int lane, OsdRows<R, W> &rows) {
    const int m = a.m;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = r * 64 + lane;
        rows.pcol[r] = -1;
        rows.unp[r] = ~0ull;
#pragma unroll
        for (int w = 0; w < W; ++w) rows.w[r][w] = (i < m && w < a.words) ? a.packed[(size_t)i * a.words + w] : 0ull;
        rows.s[r] = (i < m && a.synd[b * m + i]) ? 1u : 0u;
        rows.sm[r] = __ballot(rows.s[r] != 0);
    }
}

// osd_less as one unsigned compare: numbers ascending (-0.0 == +0.0), NaNs after every number
__device__ __forceinline__ uint64_t osd_sort_key(double x) {  // (selects, no branches: a lone wavefront pays for every jump)
    uint64_t u = __builtin_bit_cast(uint64_t, x);
    u = x == 0.0 ? 0ull : u;  // -0.0 ties with +0.0 (neither a < b nor a > b in the comparator)
    const uint64_t k = (u >> 63) ? ~u : u | (1ull << 63);
    return x != x ? ~0ull : k;
}

// soft_decision_col_sort (sort.hpp:48-62) with the keys in registers: order[rank of column i] = i
struct OsdNoRanks {};  // (osd_sort_columns without the inverse of the order)
template <int W, bool FLAT = false, class OrderPtr, class RankPtr = OsdNoRanks>
__device__ __forceinline__ void osd_sort_columns(const double *llr_row, int n, int lane, OrderPtr order, RankPtr rankv = RankPtr()) {
    uint64_t key[W];
    int rk[W];
    if constexpr (FLAT) {  // all loads in flight at once (places behind the row read its last entry), then the keys
        double val[W];
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const int j = q * 64 + lane;
            val[q] = llr_row[j < n ? j : n - 1];
        }
#pragma unroll
        for (int q = 0; q < W; ++q) key[q] = q * 64 + lane < n ? osd_sort_key(val[q]) : ~0ull;
    } else {  // (hoisting the loads costs osdw_reg_kernel<2, 4> eight registers and with them its fifth wavefront per SIMD: 92 -> 100 VGPRs)
#pragma unroll
        for (int q = 0; q < W; ++q) key[q] = q * 64 + lane < n ? osd_sort_key(llr_row[q * 64 + lane]) : ~0ull;
    }
#pragma unroll
    for (int q = 0; q < W; ++q) rk[q] = 0;
    // Every key goes round once (two v_readlane) and every lane counts it against its own keys.  A lone wavefront -- the rows that need OSD in a
    // batch of 8 192 are a few hundred, one per SIMD -- issues an instruction every ~8 cycles whatever it depends on, so what counts there is the
    // number of instructions: the loop one key at a time took 29 a key = 240 cycles, 44 % of an OSD-0 row on the BB [[144,12,12]] code
    // (tools/osd_phase_clocks.py --bb --osd0).  FLAT (osd0_reg_kernel up to four groups): the places are compile-time constants (64 x W keys
    // of straight-line code: 8 - 40 KB) -- the lane of a v_readlane and the mask of the lanes above it are literals then: 12 instructions a key
    // (4 compares, 2 carries, 2 v_readlane, select + add, and + or).  Places behind column n - 1 hold the largest key and never count before
    // a real column: a smaller index only helps among EQUAL keys (NaN columns), and theirs is the larger -- so FLAT visits whole eights.
    // Elsewhere (osdw_reg_kernel, eight groups) the loop stays: unrolling it costs registers, and those kernels live on occupancy.
    auto one_key = [&](int q, int l) __attribute__((always_inline)) {
        const uint64_t kj = osd_readlane64(key[q], l);
        // column q * 64 + l sorts before column q2 * 64 + lane: smaller key, ties by index -- and which index is smaller is
        // known at compile time unless both sit in the same group of 64, where it is a lane mask
#pragma unroll
        for (int q2 = 0; q2 < W; ++q2) {
            if (q < q2) rk[q2] += kj <= key[q2] ? 1 : 0;
            else if (q > q2) rk[q2] += kj < key[q2] ? 1 : 0;
            else {
                const uint64_t above = l >= 63 ? 0ull : ~0ull << (l + 1);  // lanes whose index is larger than l
                const uint64_t before = __ballot(kj < key[q2]) | (__ballot(kj <= key[q2]) & above);
                rk[q2] += __builtin_amdgcn_inverse_ballot_w64(before) ? 1 : 0;
            }
        }
    };
#pragma unroll
    for (int q = 0; q < W; ++q) {
        const int cnt = n - q * 64 < 64 ? n - q * 64 : 64;
        if constexpr (FLAT) {
#pragma unroll
            for (int l0 = 0; l0 < 64; l0 += 8) {
                if (l0 < cnt) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) one_key(q, l0 + u);
                }
            }
        } else {
            for (int l = 0; l < cnt; ++l) one_key(q, l);
        }
    }
#pragma unroll
    for (int q = 0; q < W; ++q)
        if (q * 64 + lane < n) {
            order[rk[q]] = q * 64 + lane;
            if constexpr (!__is_same(RankPtr, OsdNoRanks)) rankv[q * 64 + lane] = (uint16_t)rk[q];
        }
}

// One pivot step for column c, whose bit lives in word CW (compile-time) of a row; false: no unpivoted row has the bit.
// Everything that steers the step is scalar: `has` (which lanes' rows carry the bit) comes out of a compare, the set
// of unpivoted rows is kept as one 64-bit lane mask per register row, the pivot row is fetched with v_readlane from
// the right register under a scalar branch, and the XOR runs under the execution mask of the rows that have the bit.
template <int R, int W, int CW, bool SS = false>
__device__ __forceinline__ bool osd_pivot_step(OsdRows<R, W> &rows, int c, int lane) {
    uint64_t has[R];
    const uint32_t cb = 1u << (c & 31);
    if (c & 32) {
#pragma unroll
        for (int r = 0; r < R; ++r) has[r] = __ballot(((uint32_t)(rows.w[r][CW] >> 32) & cb) != 0);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) has[r] = __ballot(((uint32_t)rows.w[r][CW] & cb) != 0);
    }
    int p_lane = -1, p_r = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {  // first unpivoted row (ascending row index) with a one in column c; rows >= m are zero
        const uint64_t mask = has[r] & rows.unp[r];
        if (p_lane < 0 && mask) { p_lane = __builtin_ctzll(mask); p_r = r; }
    }
    if (p_lane < 0) return false;
    uint64_t prow[W];
    uint32_t psy = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) prow[w] = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (p_r == r) {
#pragma unroll
            for (int w = 0; w < W; ++w) prow[w] = osd_readlane64(rows.w[r][w], p_lane);
            if (SS) psy = (uint32_t)((rows.sm[r] >> p_lane) & 1ull);
            else psy = (uint32_t)__builtin_amdgcn_readlane((int)rows.s[r], p_lane);
            rows.unp[r] &= ~(1ull << p_lane);
            rows.pcol[r] = lane == p_lane ? c : rows.pcol[r];
            has[r] &= ~(1ull << p_lane);  // the pivot row keeps its own bit
        }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (SS && psy) rows.sm[r] ^= has[r];
        if (__builtin_amdgcn_inverse_ballot_w64(has[r])) {
#pragma unroll
            for (int w = 0; w < W; ++w) rows.w[r][w] ^= prow[w];
            if (!SS) rows.s[r] ^= psy;
        }
    }
    return true;
}

// Greedy elimination over the sorted columns (gf2sparse_linalg.hpp:132-226 / 298-401), rows fully reduced.
// EARLY_STOP: leave as soon as the syndrome lies in the span of the pivots (fast_solve, :373-383).  Returns the rank reached.
// The column order is read 64 entries at a time (one per lane) and handed out with v_readlane.
template <int R, int W, bool EARLY_STOP, bool SS = false, class OrderPtr>
__device__ __forceinline__ int osd_eliminate(OsdRows<R, W> &rows, OrderPtr order, int max_rank, int n, int lane) {
    // max_rank = rank H (the host knows it): once that many pivots exist every unpivoted row is zero, and the columns
    // not visited yet are non-pivot columns whatever they hold
    int rank = 0;
    for (int t0 = 0; t0 < n && rank < max_rank; t0 += 64) {
        const int mine = t0 + lane < n ? order[t0 + lane] : 0;
        const int cnt = n - t0 < 64 ? n - t0 : 64;
        bool done = false;
        for (int tt = 0; tt < cnt && rank < max_rank; ++tt) {
            const int c = __builtin_amdgcn_readlane(mine, tt);
            const int cw = c >> 6;  // scalar: a scalar branch picks the specialisation
            bool found = false;
            if (cw == 0) found = osd_pivot_step<R, W, 0, SS>(rows, c, lane);
            if constexpr (W > 1) { if (cw == 1) found = osd_pivot_step<R, W, 1, SS>(rows, c, lane); }
            if constexpr (W > 2) { if (cw == 2) found = osd_pivot_step<R, W, 2, SS>(rows, c, lane); }
            if constexpr (W > 3) { if (cw == 3) found = osd_pivot_step<R, W, 3, SS>(rows, c, lane); }
            if constexpr (W > 4) { if (cw == 4) found = osd_pivot_step<R, W, 4, SS>(rows, c, lane); }
            if constexpr (W > 5) { if (cw == 5) found = osd_pivot_step<R, W, 5, SS>(rows, c, lane); }
            if constexpr (W > 6) { if (cw == 6) found = osd_pivot_step<R, W, 6, SS>(rows, c, lane); }
            if constexpr (W > 7) { if (cw == 7) found = osd_pivot_step<R, W, 7, SS>(rows, c, lane); }
            if (!found) continue;
            ++rank;
            if (EARLY_STOP) {
                bool pending = false;
#pragma unroll
                for (int r = 0; r < R; ++r)
                    pending |= ((SS ? rows.sm[r] : __ballot(rows.s[r] != 0)) & rows.unp[r]) != 0;
                if (!pending) { done = true; break; }
            }
        }
        if (done) break;
    }
    return rank;
}

// OSD-0 with the matrix in registers (same result as osd0_kernel; chosen by the host when m <= 64 R, n + 1 <= 64 W)
template <int R, int W>
__global__ void __launch_bounds__(256) osd0_reg_kernel(const OsdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    typedef __attribute__((address_space(3))) int32_t lds_i32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = a.n;
    volatile lds_i32 *order = (volatile lds_i32 *)((__attribute__((address_space(3))) unsigned char *)osd_lds + wave * a.lds_per_wave);  // [n]
    for (int64_t b = osd_first_row(a, OSD_WAVE_WORKER()); b >= 0; b = osd_next_row(a, lane, OSD_WAVE_WORKERS())) {
        OSD_CLK_START();
        OsdRows<R, W> rows;
        osd_load_rows<R, W>(a, b, lane, rows);
        OSD_CLK(0);
        osd_sort_columns<W, (W <= 4)>(a.llr + b * n, n, lane, order);
        __builtin_amdgcn_wave_barrier();
        OSD_CLK(1);
        const int rank_reached = osd_eliminate<R, W, true, true>(rows, order, a.rank, n, lane);
        (void)rank_reached;
        OSD_CLK(2);
#ifdef LDPC_HIP_OSD_CLOCKS
        if (lane == 0) { atomicAdd(&osd_phase_clocks[8], (unsigned long long)rank_reached); atomicAdd(&osd_phase_clocks[9], 1ull); }
#endif
        if (a.status) {
            // H x = s for the x written below <=> no row without a pivot keeps a syndrome bit: the pivot rows are reduced against every pivot
            // column (x satisfies them by construction), the others have lost all their pivot-column bits and x is zero elsewhere.  The
            // elimination left either because of exactly that (fast_solve's early stop) or with rank H pivots / every column visited.
            bool pending = false;
#pragma unroll
            for (int r = 0; r < R; ++r) pending |= (rows.sm[r] & rows.unp[r]) != 0;
            if (lane == 0) a.status[b] = pending ? 2 : 1;
        }
        // x = 0 except on the pivot columns, where it is the reduced syndrome bit of the pivot's row (lu_solve, :237-288): put together in
        // LDS, where the column order lay (a wavefront's LDS instructions execute in the order issued; zeros and ones straight to global
        // memory needed a release fence in between -- a round trip to L2 -- and left as single bytes), then whole rows leave
        typedef __attribute__((address_space(3))) uint8_t lds_u8;
        volatile lds_u8 *xl = (volatile lds_u8 *)order;
        for (int j = lane; j < n; j += 64) xl[j] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (rows.pcol[r] >= 0 && __builtin_amdgcn_inverse_ballot_w64(rows.sm[r])) xl[rows.pcol[r]] = 1;
        for (int j = lane; j < n; j += 64) a.decoding[b * n + j] = xl[j];
        __builtin_amdgcn_wave_barrier();
        OSD_CLK(6);
    }
}

// ---- OSD-0 for small matrices with the COLUMNS PERMUTED into their sorted order (round 6) -----------------------------------------
// What osd0_reg_kernel pays per pivot is ~100 instructions and a dozen jumps, most of them because the column it visits sits in a
// word and a half-word only known at run time (a scalar branch per word, another per half, the pivot row fetched whole) -- and a lone
// wavefront (a batch of 8 192 leaves a few hundred rows for OSD: one per SIMD) issues an instruction every ~8 cycles whatever it depends
// on, so the kernel ends with its slowest row: up to rank H pivots.  Here the matrix is BUILT with column t = the t-th column of the
// order: a row's entries come as eight 16-bit column numbers (OsdArgs::ell, one 16-byte load), each is looked up in the inverse of the
// order (the ranks the sort produces anyway) and ORed into the row's bits in LDS; the rows then live in registers as D dwords with the
// visit running over dword d = 0 .. D - 1 (compile-time) and bit b (a scalar shift).  A step is two compares per register row, the
// scalar pivot search, the pivot row's REMAINING dwords d .. D - 1 (earlier columns are never looked at again), XORs under the mask of
// the rows that have the bit; the syndrome column and the rows without a pivot are scalar lane masks as in osd0_reg_kernel: ~75
// instructions, half a dozen jumps.  Same pivots as osd0_reg_kernel (first row without a pivot, ascending, that has the bit), same x,
// same status.  m <= 64 R (R <= 2), n <= 32 D, rows of at most eight entries.
__host__ __device__ inline size_t osd_flat_lds_bytes(int n, int R, int D) {
    const size_t n4 = ((size_t)n + 3) & ~(size_t)3;
    return ((n4 * 4 + n4 * 2 + n4 + (size_t)64 * R * (D | 1) * 4) + 15) & ~(size_t)15;  // order, ranks, x, the rows (odd stride: no bank conflicts)
}

template <int R, int D>
__global__ void __launch_bounds__(256) osd0_flat_kernel(const OsdArgs a) {
    static_assert(R >= 1 && R <= 2 && D >= 1 && D <= 8, "up to 128 rows, 256 columns");
    constexpr int W = (D + 1) / 2;  // groups of 64 columns in the sort
    constexpr int DS = D | 1;
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    typedef __attribute__((address_space(3))) int32_t lds_i32;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = a.n, m = a.m;
    const int n4 = (n + 3) & ~3;
    auto base = (__attribute__((address_space(3))) unsigned char *)osd_lds + wave * a.lds_per_wave;
    volatile lds_i32 *order = (volatile lds_i32 *)base;                    // [n] column with rank t
    volatile lds_u16 *rankv = (volatile lds_u16 *)(base + n4 * 4);         // [n] rank of column j
    volatile lds_u8 *xl = (volatile lds_u8 *)(base + n4 * 6);              // [n] the solution, before it leaves
    volatile lds_u32 *P = (volatile lds_u32 *)(base + n4 * 7);             // [64 R][DS] the rows, columns in sorted order
    for (int64_t b = osd_first_row(a, OSD_WAVE_WORKER()); b >= 0; b = osd_next_row(a, lane, OSD_WAVE_WORKERS())) {
        OSD_CLK_START();
        uint64_t sm[R], unp[R];  // wave-uniform: rows whose (reduced) syndrome bit is set / that carry no pivot yet
        uint4 ell[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = r * 64 + lane;
            sm[r] = __ballot(i < m && a.synd[b * m + i] != 0);  // (a non-zero byte is a one, gf2sparse_linalg.hpp:309)
            unp[r] = ~0ull;                                      // (rows >= m are empty: never candidates)
            ell[r] = i < m ? reinterpret_cast<const uint4 *>(a.ell)[i] : make_uint4(~0u, ~0u, ~0u, ~0u);
#pragma unroll
            for (int dd = 0; dd < D; ++dd) P[(r * 64 + lane) * DS + dd] = 0;
        }
        OSD_CLK(0);
        osd_sort_columns<W, true>(a.llr + b * n, n, lane, order, rankv);
        __builtin_amdgcn_wave_barrier();
        OSD_CLK(1);
        // the rows with their columns in sorted order: bit t of a row = its entry in the column of rank t
        // (a wavefront's LDS instructions execute in the order issued: the ranks are there, the zeros are there)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t e4[4] = {ell[r].x, ell[r].y, ell[r].z, ell[r].w};
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t c = (e4[k >> 1] >> ((k & 1) * 16)) & 0xffffu;
                if (c != 0xffffu) {
                    const uint32_t t = rankv[c];
                    __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t *)&P[(r * 64 + lane) * DS + (t >> 5)], 1u << (t & 31), __ATOMIC_RELAXED,
                                          __HIP_MEMORY_SCOPE_WAVEFRONT);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        uint32_t w[R][D];
        int pc[R];  // sorted position of the pivot column the row carries, -1: none
#pragma unroll
        for (int r = 0; r < R; ++r) {
            pc[r] = -1;
#pragma unroll
            for (int dd = 0; dd < D; ++dd) w[r][dd] = P[(r * 64 + lane) * DS + dd];
        }
        OSD_CLK(3);
        // greedy elimination over the sorted columns (gf2sparse_linalg.hpp:298-401): stops with rank H pivots, or as soon as the syndrome
        // lies in the span of the pivots (fast_solve, :373-383)
        int rank = 0;
        bool done = a.rank <= 0 || ((sm[0] & unp[0]) | (R > 1 ? sm[R - 1] & unp[R - 1] : 0ull)) == 0;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int nb = n - 32 * d < 32 ? n - 32 * d : 32;
            for (int bit = 0; bit < nb && !done; ++bit) {
                const uint32_t mask = 1u << bit;
                uint64_t has[R];
#pragma unroll
                for (int r = 0; r < R; ++r) has[r] = __ballot((w[r][d] & mask) != 0);
                const uint64_t c0 = has[0] & unp[0], c1 = R > 1 ? has[R - 1] & unp[R - 1] : 0ull;
                if ((c0 | c1) == 0) continue;
                uint32_t prow[D];
                uint32_t psy;
                const int t = 32 * d + bit;
                // (written out per register row that can carry the pivot with everything behind it, the two cases cost 20 register moves to join
                // again: the common tail below with its scalar selects is the shorter form -- ~75 instructions a pivot)
                if (R == 1 || c0 != 0) {
                    const int p = __builtin_ctzll(c0);
#pragma unroll
                    for (int dd = d; dd < D; ++dd) prow[dd] = (uint32_t)__builtin_amdgcn_readlane((int)w[0][dd], p);
                    psy = (uint32_t)((sm[0] >> p) & 1ull);
                    unp[0] &= ~(1ull << p);
                    has[0] &= ~(1ull << p);  // the pivot row keeps its own bit
                    pc[0] = lane == p ? t : pc[0];
                } else {
                    const int p = __builtin_ctzll(c1);
#pragma unroll
                    for (int dd = d; dd < D; ++dd) prow[dd] = (uint32_t)__builtin_amdgcn_readlane((int)w[R - 1][dd], p);
                    psy = (uint32_t)((sm[R - 1] >> p) & 1ull);
                    unp[R - 1] &= ~(1ull << p);
                    has[R - 1] &= ~(1ull << p);
                    pc[R - 1] = lane == p ? t : pc[R - 1];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    if (psy) sm[r] ^= has[r];
                    if (__builtin_amdgcn_inverse_ballot_w64(has[r])) {
#pragma unroll
                        for (int dd = d; dd < D; ++dd) w[r][dd] ^= prow[dd];
                    }
                }
                ++rank;
                done = rank >= a.rank || ((sm[0] & unp[0]) | (R > 1 ? sm[R - 1] & unp[R - 1] : 0ull)) == 0;
            }
        }
        OSD_CLK(2);
#ifdef LDPC_HIP_OSD_CLOCKS
        if (lane == 0) { atomicAdd(&osd_phase_clocks[8], (unsigned long long)rank); atomicAdd(&osd_phase_clocks[9], 1ull); }
#endif
        if (a.status) {  // (as osd0_reg_kernel: H x = s <=> no row without a pivot keeps a syndrome bit)
            const bool pending = ((sm[0] & unp[0]) | (R > 1 ? sm[R - 1] & unp[R - 1] : 0ull)) != 0;
            if (lane == 0) a.status[b] = pending ? 2 : 1;
        }
        // x = 0 except on the pivot columns, where it is the reduced syndrome bit of the pivot's row (lu_solve, :237-288)
        for (int j = lane; j < n; j += 64) xl[j] = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (pc[r] >= 0 && __builtin_amdgcn_inverse_ballot_w64(sm[r])) xl[order[pc[r]]] = 1;
        for (int j = lane; j < n; j += 64) a.decoding[b * n + j] = xl[j];
        __builtin_amdgcn_wave_barrier();
        OSD_CLK(6);
    }
}

// ---- higher-order OSD (osd.hpp:119-187): OSD_E / OSD_CS, one wavefront per unconverged syndrome ---------------
// After the column sort the matrix is brought to REDUCED row echelon form over the sorted columns (no early
// stop).  Then no candidate needs a solve of its own: flipping the non-pivot columns F changes the solution on
// the pivot column of row r by XOR_{f in F} R[r][f] (R = the reduced matrix), so a candidate is the OSD-0
// solution, a mask over the non-pivot columns, and one parity per pivot row.  Candidates are spread over the
// lanes; each lane adds up its candidate's weight in ascending bit order exactly as the reference does
// (sequential FP64 sum of log(1/p_j) over the support), and the first strictly lightest candidate wins.
struct OsdCandidate {
    uint64_t mask;  // chosen columns among the first 64 non-pivot columns (sorted order)
    int32_t single; // a chosen non-pivot column beyond the first 64 (OSD_CS weight-one strings, pairs that reach past column 63), else -1
    int32_t single2; // the second such column of a pair, else -1
    bool valid;
};

// Pair number p of the reference's i-major list of pairs i < j < order (osd.hpp:91-99) -> (i, j).  Row i starts at
// S(i) = i (2 order - 1 - i) / 2.  Small orders walk the rows (a few steps); larger ones -- OSD_CS takes any order up to n - rank, millions of
// pairs on hypergraph-product codes -- invert the triangular number: i = floor((b - sqrt(b^2 - 8 p)) / 2), b = 2 order - 1, and the two loops
// after it make the result exact whatever the rounding of the square root did (each runs at most a step or two).  p < order (order - 1) / 2.
__device__ __forceinline__ void osd_pair_of(long p, int order, int &i, int &j) {
    if (order <= 48) {
        i = 0;
        while (p >= order - 1 - i) { p -= order - 1 - i; ++i; }
        j = i + 1 + (int)p;
        return;
    }
    const double b = 2.0 * (double)order - 1.0;
    double disc = b * b - 8.0 * (double)p;
    if (disc < 0.0) disc = 0.0;
    long r = (long)((b - sqrt(disc)) * 0.5);
    if (r < 0) r = 0;
    if (r > order - 2) r = order - 2;
    const long n2 = 2L * order - 1;
    while (r > 0 && r * (n2 - r) / 2 > p) --r;
    while (r < order - 2 && (r + 1) * (n2 - (r + 1)) / 2 <= p) ++r;
    i = (int)r;
    j = i + 1 + (int)(p - r * (n2 - r) / 2);
}

__device__ __forceinline__ OsdCandidate osd_candidate(int method, int order, int k, long c) {
    OsdCandidate r;
    r.mask = 0;
    r.single = r.single2 = -1;
    r.valid = true;
    const uint64_t kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    if (method == 2) {  // numbers 1 .. 2^order - 1, bit j -> j-th non-pivot column, bits >= k dropped (util.hpp:12-38)
        r.mask = (uint64_t)(c + 1) & kmask;
    } else if (c < k) {  // weight one, every non-pivot column (osd.hpp:84-89)
        if (c < 64) r.mask = 1ull << c; else r.single = (int32_t)c;
    } else {  // pairs (i, j), i < j < order, i-major (osd.hpp:91-99)
        int i, j;
        osd_pair_of(c - k, order, i, j);
        if (j >= k) r.valid = false;  // past the candidate string in the reference
        else {  // (any osd_order <= k: columns below 64 go into the mask, the others are named)
            if (i < 64) r.mask |= 1ull << i; else r.single = i;
            if (j < 64) r.mask |= 1ull << j; else if (r.single < 0) r.single = j; else r.single2 = j;
        }
    }
    return r;
}

__global__ void __launch_bounds__(256) osdw_kernel(const OsdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int m = a.m, n = a.n, W = a.words;
    unsigned char *base = osd_lds + (size_t)wave * a.lds_per_wave;
    volatile uint64_t *mat = reinterpret_cast<volatile uint64_t *>(base);                         // [m][W]
    volatile uint64_t *T = mat + (size_t)m * W;                                                   // [m]
    volatile double *keys = reinterpret_cast<volatile double *>(const_cast<uint64_t *>(T + m));   // [n] log-ratios, later weights
    volatile int32_t *order = reinterpret_cast<volatile int32_t *>(const_cast<double *>(keys + n));  // [n]
    volatile int32_t *code = order + n;       // [n] pivot column: its row; non-pivot column: -1 - position among the non-pivots
    volatile int32_t *npcol = code + n;       // [n] non-pivot columns in sorted order
    volatile int32_t *pivot_col = npcol + n;  // [m]

    const int sw = n >> 6;
    const uint64_t sbit = 1ull << (n & 63);
    for (int64_t b = osd_first_row(a, OSD_WAVE_WORKER()); b >= 0; b = osd_next_row(a, lane, OSD_WAVE_WORKERS())) {
    for (int i = lane; i < m; i += 64) {
        for (int w = 0; w < W; ++w) mat[(size_t)i * W + w] = 0;
        for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) {
            const int c = a.col_idx[e];
            mat[(size_t)i * W + (c >> 6)] = mat[(size_t)i * W + (c >> 6)] | (1ull << (c & 63));
        }
        if (a.synd[b * m + i]) mat[(size_t)i * W + sw] = mat[(size_t)i * W + sw] | sbit;
        pivot_col[i] = -1;
    }
    for (int j = lane; j < n; j += 64) { keys[j] = a.llr[b * n + j]; code[j] = INT32_MIN; }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < n; i += 64) {  // soft_decision_col_sort (sort.hpp:48-62)
        const double ki = keys[i];
        int r = 0;
        for (int j = 0; j < n; ++j) r += osd_less(keys[j], j, ki, i) ? 1 : 0;
        order[r] = i;
    }
    __builtin_amdgcn_wave_barrier();

    // rref over the sorted columns (gf2sparse_linalg.hpp:132-226), rows fully reduced
    const int max_rank = m < n ? m : n;
    int rank = 0;
    for (int t = 0; t < n && rank < max_rank; ++t) {
        const int c = order[t];
        const int cw = c >> 6;
        const uint64_t cb = 1ull << (c & 63);
        int p = -1;
        for (int i0 = 0; i0 < m && p < 0; i0 += 64) {
            const int i = i0 + lane;
            const bool cand = i < m && pivot_col[i] < 0 && (mat[(size_t)i * W + cw] & cb);
            const uint64_t mask = __ballot(cand);
            if (mask) p = i0 + __builtin_ctzll(mask);
        }
        if (p < 0) continue;
        for (int i = lane; i < m; i += 64)
            if (i != p && (mat[(size_t)i * W + cw] & cb))
                for (int w = 0; w < W; ++w) mat[(size_t)i * W + w] = mat[(size_t)i * W + w] ^ mat[(size_t)p * W + w];
        if (lane == 0) { pivot_col[p] = c; code[c] = p; }
        ++rank;
        __builtin_amdgcn_wave_barrier();
    }
    // non-pivot columns in sorted order (`cols[rank ..]`, gf2sparse_linalg.hpp:210-224)
    int k = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const int c = t < n ? order[t] : 0;
        const bool np = t < n && code[c] < 0;
        const uint64_t mask = __ballot(np);
        if (np) {
            const int q = k + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
            npcol[q] = c;
            code[c] = -1 - q;
        }
        k += __builtin_popcountll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    const int k64 = k < 64 ? k : 64;
    for (int r = lane; r < m; r += 64) {  // the reduced matrix on the first 64 non-pivot columns, one word per row
        uint64_t t = 0;
        for (int q = 0; q < k64; ++q) {
            const int c = npcol[q];
            t |= ((mat[(size_t)r * W + (c >> 6)] >> (c & 63)) & 1ull) << q;
        }
        T[r] = t;
    }
    for (int j = lane; j < n; j += 64) keys[j] = a.wt[j];
    __builtin_amdgcn_wave_barrier();

    // weight of a candidate: sum over its support in ascending bit order (osd.hpp:171-176)
    auto bit_of = [&](const OsdCandidate &cd, int i) -> bool {
        const int cdi = code[i];
        if (cdi >= 0) {
            uint64_t v = (mat[(size_t)cdi * W + sw] >> (n & 63)) ^ (uint64_t)__builtin_popcountll(T[cdi] & cd.mask);
            if (cd.single >= 0) {
                const int c = npcol[cd.single];
                v ^= mat[(size_t)cdi * W + (c >> 6)] >> (c & 63);
            }
            if (cd.single2 >= 0) {
                const int c = npcol[cd.single2];
                v ^= mat[(size_t)cdi * W + (c >> 6)] >> (c & 63);
            }
            return (v & 1ull) != 0;
        }
        const int q = -1 - cdi;
        return (q < 64 && ((cd.mask >> (q & 63)) & 1ull)) || q == cd.single || q == cd.single2;  // (q & 63: the shift is defined whichever way the test is compiled)
    };
    auto weight_of = [&](const OsdCandidate &cd) -> double {
        double acc = 0;
        for (int i = 0; i < n; ++i)
            if (bit_of(cd, i)) acc += keys[i];
        return acc;
    };
    OsdCandidate none;
    none.mask = 0; none.single = none.single2 = -1; none.valid = true;
    const double w0 = weight_of(none);  // the OSD-0 solution (osd.hpp:131-136)
    const long ncand = a.method == 2 ? (1L << a.order) - 1 : (long)k + (long)a.order * (a.order - 1) / 2;
    double best_w = w0;
    long best_c = -1;
    for (long c0 = 0; c0 < ncand; c0 += 64) {
        const long c = c0 + lane;
        if (c < ncand) {
            const OsdCandidate cd = osd_candidate(a.method, a.order, k, c);
            if (cd.valid) {
                const double w = weight_of(cd);
                if (w < best_w) { best_w = w; best_c = c; }  // strict: the first lightest candidate stays (osd.hpp:177)
            }
        }
    }
    // across lanes: lightest, then earliest
    for (int off = 32; off > 0; off >>= 1) {
        const double ow = __shfl_xor(best_w, off);
        const long oc = ((long)__shfl_xor((int)(best_c >> 32), off) << 32) | (unsigned)__shfl_xor((int)(best_c & 0xffffffff), off);
        const bool mine_set = best_c >= 0, other_set = oc >= 0;
        if (other_set && (!mine_set || ow < best_w || (ow == best_w && oc < best_c))) { best_w = ow; best_c = oc; }
    }
    OsdCandidate win = none;
    if (best_c >= 0) win = osd_candidate(a.method, a.order, k, best_c);
    for (int j = lane; j < n; j += 64) a.decoding[b * n + j] = bit_of(win, j) ? 1 : 0;
    __builtin_amdgcn_wave_barrier();
    }  // next row
}


// ---- higher-order OSD with the elimination in registers (m <= 256, n <= 511) --------------------------------------
// As osdw_kernel, but (i) the reduced row echelon form is computed in registers (osd_eliminate), and (ii) what a
// candidate needs is laid out per COLUMN once: for column i a record {weight log(1 / p_i), S_i, T_i[0 .. KW)} with
// KW = ceil(k / 64), k = n - rank, such that the candidate flipping the set M of non-pivot columns has
//     x_i = parity(T_i & M) ^ S_i
// -- for a pivot column T is its row of the reduced matrix restricted to the non-pivot columns (bit q = the q-th
// non-pivot column in sorted order) and S the reduced syndrome bit; for the q-th non-pivot column T = 1 << q, S = 0.
// The candidate strings of the reference are of two shapes only: ONE non-pivot column q (OSD_CS, osd.hpp:84-89) -- x_i
// = bit q of T_i ^ S_i, and with lane = q - 64 v one broadcast read of word v per column serves 64 candidates -- or a
// mask inside the first 64 non-pivot columns (OSD_CS pairs, osd.hpp:91-99; every OSD_E string, util.hpp:12-38) -- one
// AND + popcount of word 0.  Weighing is one pass of broadcast LDS reads in ascending bit order, accumulated
// sequentially as the reference does (osd.hpp:171-176); no dependent look-ups, no per-candidate solve.
// acc[r] |= (bit `cbit` of word CW of row r) << qq for the R rows of this lane; cbit, qq scalar
template <int R, int W, int CW>
__device__ __forceinline__ void osd_gather_step(const OsdRows<R, W> &rows, uint32_t (&acc)[R], int cbit, int qq) {
    if (cbit & 32) {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] |= (((uint32_t)(rows.w[r][CW] >> 32) >> (cbit & 31)) & 1u) << qq;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] |= (((uint32_t)rows.w[r][CW] >> cbit) & 1u) << qq;
    }
}

template <int R, int W>
__global__ void __launch_bounds__(256) osdw_reg_kernel(const OsdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) int32_t lds_i32;
    typedef __attribute__((address_space(3))) uint32_t lds_u32;
    typedef __attribute__((address_space(3))) uint64_t lds_u64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int n = a.n, KW = a.kwords, RS = a.kwords + 2;  // RS: 64-bit words of a column record
    lds_u8 *base = (lds_u8 *)osd_lds + wave * a.lds_per_wave;
    volatile lds_u64 *rec = (volatile lds_u64 *)base;              // [n][RS]  {weight, S, T[KW]}
    volatile lds_i32 *order = (volatile lds_i32 *)(rec + (size_t)n * RS);  // [n]
    volatile lds_i32 *colQ = order + n;                            // [n] -2: unseen, -1: pivot column, q >= 0: the q-th non-pivot column
    volatile lds_i32 *npcol = colQ + n;                            // [64 KW]
    for (int64_t b = osd_first_row(a, OSD_WAVE_WORKER()); b >= 0; b = osd_next_row(a, lane, OSD_WAVE_WORKERS())) {
        OSD_CLK_START();
        OsdRows<R, W> rows;
        osd_load_rows<R, W>(a, b, lane, rows);
        OSD_CLK(0);
        osd_sort_columns<W>(a.llr + b * n, n, lane, order);
        OSD_CLK(1);
        for (int j = lane; j < n; j += 64) { colQ[j] = -2; rec[(size_t)j * RS] = __builtin_bit_cast(uint64_t, a.wt[j]); }
        __builtin_amdgcn_wave_barrier();
        osd_eliminate<R, W, false>(rows, order, a.rank, n, lane);
        OSD_CLK(2);
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (rows.pcol[r] >= 0) {
                colQ[rows.pcol[r]] = -1;
                rec[(size_t)rows.pcol[r] * RS + 1] = rows.s[r];
            }
        __builtin_amdgcn_wave_barrier();
        int k = 0;  // non-pivot columns in sorted order (`cols[rank ..]`, gf2sparse_linalg.hpp:210-224)
        for (int t0 = 0; t0 < n; t0 += 64) {
            const int t = t0 + lane;
            const int c = t < n ? order[t] : 0;
            const bool np = t < n && colQ[c] == -2;
            const uint64_t mask = __ballot(np);
            if (np) {
                const int q = k + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                colQ[c] = q;
                rec[(size_t)c * RS + 1] = 0;
                for (int v = 0; v < KW; ++v) rec[(size_t)c * RS + 2 + v] = (q >> 6) == v ? 1ull << (q & 63) : 0ull;
                if (q < 64 * KW) npcol[q] = c;
            }
            k += __builtin_popcountll(mask);
        }
        __builtin_amdgcn_wave_barrier();
        OSD_CLK(3);
        if (k > 64 * KW) k = 64 * KW;  // cannot happen: the host sized KW from the rank of H
        // the reduced matrix on the non-pivot columns, 32 columns at a time: their numbers sit one per lane and come
        // out with v_readlane, a scalar branch picks the register word, each row adds its bit
        volatile lds_u32 *rec32 = (volatile lds_u32 *)rec;
        for (int v2 = 0; v2 * 32 < k; ++v2) {
            const int cnt = k - v2 * 32 < 32 ? k - v2 * 32 : 32;
            const int mine = lane < cnt ? npcol[v2 * 32 + lane] : 0;
            uint32_t acc[R];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = 0;
            for (int qq = 0; qq < cnt; ++qq) {
                const int c = __builtin_amdgcn_readlane(mine, qq);
                const int cw = c >> 6, cbit = c & 63;
                if (cw == 0) osd_gather_step<R, W, 0>(rows, acc, cbit, qq);
                if constexpr (W > 1) { if (cw == 1) osd_gather_step<R, W, 1>(rows, acc, cbit, qq); }
                if constexpr (W > 2) { if (cw == 2) osd_gather_step<R, W, 2>(rows, acc, cbit, qq); }
                if constexpr (W > 3) { if (cw == 3) osd_gather_step<R, W, 3>(rows, acc, cbit, qq); }
                if constexpr (W > 4) { if (cw == 4) osd_gather_step<R, W, 4>(rows, acc, cbit, qq); }
                if constexpr (W > 5) { if (cw == 5) osd_gather_step<R, W, 5>(rows, acc, cbit, qq); }
                if constexpr (W > 6) { if (cw == 6) osd_gather_step<R, W, 6>(rows, acc, cbit, qq); }
                if constexpr (W > 7) { if (cw == 7) osd_gather_step<R, W, 7>(rows, acc, cbit, qq); }
            }
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (rows.pcol[r] >= 0) rec32[((size_t)rows.pcol[r] * RS + 2) * 2 + v2] = acc[r];
        }
        if ((k & 63) != 0 && (k & 63) <= 32) {  // an odd number of 32-column groups: clear the upper half of the last word
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (rows.pcol[r] >= 0) rec32[((size_t)rows.pcol[r] * RS + 2) * 2 + (k >> 6) * 2 + 1] = 0;
        }
        for (int v = (k + 63) >> 6; v < KW; ++v) {  // k = 0 (KW is at least 1)
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (rows.pcol[r] >= 0) rec[(size_t)rows.pcol[r] * RS + 2 + v] = 0;
        }
        __builtin_amdgcn_wave_barrier();
        OSD_CLK(4);
        // the records are complete and only read from here on: plain (non-volatile) views let the loads of several
        // columns be in flight together
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        const lds_u64 *rec_r = (const lds_u64 *)rec;
        const uint32_t lm_lo = lane < 32 ? 1u << lane : 0u, lm_hi = lane >= 32 ? 1u << (lane - 32) : 0u;
        // (`acc += flip ? weight : 0.0` adds the same numbers as the reference's conditional sum: acc starts at +0.0
        // and x + 0.0 == x bit for bit for every x an accumulator that started at +0.0 can hold)
        auto weigh_single = [&](int v) -> double {  // this lane's candidate: the non-pivot column 64 v + lane alone
            double acc = 0;  // added in column order, as the reference does
#pragma unroll 8
            for (int i = 0; i < n; ++i) {
                const uint64_t t = rec_r[(size_t)i * RS + 2 + v];
                const uint32_t flip = (((uint32_t)t & lm_lo) | ((uint32_t)(t >> 32) & lm_hi)) != 0 ? 1u : 0u;
                const double wgt = __builtin_bit_cast(double, rec_r[(size_t)i * RS]);
                acc += (flip ^ (uint32_t)rec_r[(size_t)i * RS + 1]) ? wgt : 0.0;
            }
            return acc;
        };
        auto weigh_mask = [&](uint64_t mask) -> double {  // this lane's candidate: a set of the first 64 non-pivot columns
            double acc = 0;
#pragma unroll 8
            for (int i = 0; i < n; ++i) {
                const uint64_t t = rec_r[(size_t)i * RS + 2] & mask;  // (v_bcnt_u32_b32 adds its second operand: S rides along)
                const int par = __builtin_popcount((uint32_t)(t >> 32)) + (__builtin_popcount((uint32_t)t) + (int)(uint32_t)rec_r[(size_t)i * RS + 1]);
                const double wgt = __builtin_bit_cast(double, rec_r[(size_t)i * RS]);
                acc += (par & 1) ? wgt : 0.0;
            }
            return acc;
        };
        auto pair_mask = [&](long p, bool &valid) -> uint64_t {  // pairs (i, j), i < j < order <= 64, i-major (osd.hpp:91-99)
            int i = 0;
            while (p >= a.order - 1 - i) { p -= a.order - 1 - i; ++i; }
            const int j = i + 1 + (int)p;
            valid = j < k;  // past the candidate string in the reference
            return valid ? (1ull << i) | (1ull << j) : 0ull;
        };
        // osd_order > 64: a pair may sit in any two words of T -- each lane reads the two words of ITS pair (no broadcast any more)
        auto pair_cols = [&](long p, int &i, int &j) -> bool {  // pair number p of the reference's list (i-major, osd.hpp:91-99) -> i < j; false: past the string
            osd_pair_of(p, a.order, i, j);
            return j < k;
        };
        auto weigh_pair = [&](int qi, int qj) -> double {
            double acc = 0;
            const int wi = 2 + (qi >> 6), wj = 2 + (qj >> 6), si = qi & 63, sj = qj & 63;
#pragma unroll 4
            for (int i = 0; i < n; ++i) {
                const uint64_t x = (rec_r[(size_t)i * RS + wi] >> si) ^ (rec_r[(size_t)i * RS + wj] >> sj) ^ rec_r[(size_t)i * RS + 1];
                const double wgt = __builtin_bit_cast(double, rec_r[(size_t)i * RS]);
                acc += (x & 1ull) ? wgt : 0.0;
            }
            return acc;
        };
        double best_w = weigh_mask(0);  // the OSD-0 solution (osd.hpp:131-136)
        long best_c = -1;               // index in the reference's candidate list; -1: the OSD-0 solution
        const uint64_t kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
        const long npairs = (long)a.order * (a.order - 1) / 2;
        if (a.method == 3) {
            for (int v = 0; v * 64 < k; ++v) {
                const double w = weigh_single(v);
                if (v * 64 + lane < k && w < best_w) { best_w = w; best_c = v * 64 + lane; }  // strict: the first lightest candidate stays (osd.hpp:177)
            }
            if (a.order <= 64) {
                for (long p0 = 0; p0 < npairs; p0 += 64) {
                    const long pp = p0 + lane;
                    bool valid = false;
                    const uint64_t mask = pp < npairs ? pair_mask(pp, valid) : 0ull;
                    const double w = weigh_mask(mask);
                    if (valid && w < best_w) { best_w = w; best_c = k + pp; }
                }
            } else {
                for (long p0 = 0; p0 < npairs; p0 += 64) {
                    const long pp = p0 + lane;
                    int qi = 0, qj = 0;
                    const bool valid = pp < npairs && pair_cols(pp, qi, qj);
                    if (__ballot(valid) == 0) continue;  // (a whole round past the string: osd_order > k)
                    const double w = weigh_pair(valid ? qi : 0, valid ? qj : 0);
                    if (valid && w < best_w) { best_w = w; best_c = k + pp; }
                }
            }
        } else {  // numbers 1 .. 2^order - 1 (order <= 24), bits >= k dropped (util.hpp:12-38)
            const long total = (1L << a.order) - 1;
            for (long c0 = 0; c0 < total; c0 += 64) {
                const long c = c0 + lane;
                const double w = weigh_mask((uint64_t)(c + 1) & kmask);
                if (c < total && w < best_w) { best_w = w; best_c = c; }
            }
        }
        OSD_CLK(5);
        for (int off = 32; off > 0; off >>= 1) {  // across lanes: lightest, then earliest
            const double ow = __shfl_xor(best_w, off);
            const long oc = ((long)__shfl_xor((int)(best_c >> 32), off) << 32) | (unsigned)__shfl_xor((int)(best_c & 0xffffffff), off);
            const bool mine_set = best_c >= 0, other_set = oc >= 0;
            if (other_set && (!mine_set || ow < best_w || (ow == best_w && oc < best_c))) { best_w = ow; best_c = oc; }
        }
        const bool single = a.method == 3 && best_c >= 0 && best_c < k;
        const bool far_pair = a.method == 3 && best_c >= k && a.order > 64;
        uint64_t win = 0;
        int wq_i = 0, wq_j = 0;
        if (far_pair) (void)pair_cols(best_c - k, wq_i, wq_j);
        else if (best_c >= 0 && !single) {
            bool valid;
            win = a.method == 3 ? pair_mask(best_c - k, valid) : (uint64_t)(best_c + 1) & kmask;
        }
        for (int j = lane; j < n; j += 64) {
            uint64_t x = rec_r[(size_t)j * RS + 1];
            if (single) x ^= rec_r[(size_t)j * RS + 2 + (best_c >> 6)] >> (best_c & 63);
            else if (far_pair) x ^= (rec_r[(size_t)j * RS + 2 + (wq_i >> 6)] >> (wq_i & 63)) ^ (rec_r[(size_t)j * RS + 2 + (wq_j >> 6)] >> (wq_j & 63));
            else x ^= (uint64_t)__builtin_popcountll(rec_r[(size_t)j * RS + 2] & win);
            a.decoding[b * n + j] = (uint8_t)(x & 1ull);
        }
        __atomic_signal_fence(__ATOMIC_SEQ_CST);
        __builtin_amdgcn_wave_barrier();
        OSD_CLK(6);
    }
}

// ---- OSD with one WORKGROUP per syndrome: [H] in LDS (mid-size matrices) or in a global scratch slot (beyond LDS) ------
// Same results as osd0_kernel / osdw_kernel.  The columns are sorted first (a bitonic network over column numbers comparing
// (key, number), hence stable), and the working copy of H is then built WITH ITS COLUMNS IN THAT ORDER, word-plane major, in LDS
// or in an HBM / MALL slot: plane w holds the sorted columns 64 w .. 64 w + 63 of every row, so "which rows have a one in the next
// 64 columns" is one coalesced read of one plane, and the non-pivot columns come out in candidate order.  LDS holds what is
// touched all the time: the column order, the syndrome column, the pivot columns and -- phase by phase in the same room -- the
// sort keys, the elimination's look-ahead words and combination table, the candidates' column info and staged planes.
// The elimination is blocked (osd_block_eliminate): 64 columns at a time on the look-ahead words alone, then one combined
// update of the rows that took a pivot.  Matrices with more than OSD_BLOCK_ROWS rows keep the one-pivot-per-step loop.
// HIGHER (OSD_E / OSD_CS): no early stop; afterwards the reduced rows are squeezed to the non-pivot columns (T, again
// plane-major, behind the matrix in the slot: one pass, a parallel bit compress per plane) and the candidates are weighed as in
// osdw_reg_kernel: lane = candidate, a task of 64 per wavefront, each wavefront with the T plane it needs staged in LDS.
struct OsdBigArgs {
    OsdArgs o;
    uint64_t *scratch;      // [slots][hwords + kwords][m]
    int64_t slot_stride;    // 64-bit words per slot
    int32_t hwords;         // ceil(n / 64)
    int32_t pow2;           // bitonic size: smallest power of two >= n
    int32_t max_rank;       // rank of H if the host worked it out, else min(m, n)
    int32_t kwords;         // HIGHER: planes of T the slot has room for (>= ceil((n - rank) / 64))
    int32_t extra_off;      // byte offset in LDS of the room the phases share: keys [n] u64 | positions [n] u16 | look [m] u64 + table | {colinfo [n] i16, (8-aligned) plane masks and moves [7][hwords] u64, planes [nplanes][m + 1] u64}
    int32_t mat_off;        // MAT_LDS: byte offset in LDS of the working copy [hwords][m]
    int32_t nplanes;        // HIGHER: wavefronts that weigh candidates, each with its own staged plane [m + 1] u64 in LDS (4, 2 or 1)
    int32_t pbuf_off;       // blocked elimination (m <= OSD_BLOCK_ROWS): byte offset in LDS of the combination table (extra_off + 8 m); -1: one pivot per step
};

// ---- blocked elimination: the 64 columns of a look-ahead block, rows in registers ----------------------------------------------
// The one-pivot-per-step loop below pays, per pivot, two workgroup barriers around a read-modify-write of every hit row in
// L2 / MALL (H in an HBM slot) -- ~800 dependent round trips for a 768 x 1600 matrix.  Blocked: the next 64 sorted columns
// of every row are one word (`look`: a plane of the working copy); the workgroup eliminates on those words alone, thread t holding rows
// t, t + 256, ... in registers, and records for every row r a mask M_r over the block's pivots meaning
//     row_r (after the block) = row_r (at block start) ^ XOR_{j in M_r} pivot_row_j (at block start)
// (taking pivot p's row: M_r ^= M_p ^ {p}).  Per column: every wavefront offers its first unpivoted row with the bit (a ballot
// per register row) together with that row's two words through LDS, ONE barrier, everybody takes the lowest offer.  Then every
// row with a non-empty mask takes ONE combined update over all planes (osd_big_kernel below) -- per block instead of per pivot,
// from a table of the XOR combinations of every four pivot rows -- and its syndrome bit parity(M_r & S), S_j = syndrome bit
// of pivot row j at block start.  Same pivots (first unpivoted row with the bit, columns in sorted order), same reduced matrix:
// tools/proto_blocked_elimination.py checks the algebra against the one-pivot-at-a-time form, the golden fixtures the kernel.
__device__ __forceinline__ unsigned osd_bit_at(uint64_t x, int j) {  // j uniform: one 32-bit shift instead of a 64-bit one
    const unsigned h = j < 32 ? (unsigned)x : (unsigned)(x >> 32);
    return (h >> (j & 31)) & 1u;
}

template <int MQ>
__device__ __forceinline__ int osd_block_eliminate(int tid, int m, int ahead, uint64_t *look, const int16_t *pivcol,
                                                uint16_t *blk_row, uint16_t *blk_col, unsigned long long *xch, int rank, int max_rank) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint64_t L[MQ], M[MQ];
    uint64_t unpm[MQ];  // per register row: the lanes whose row exists and carries no pivot yet (the same for the whole wavefront: scalar registers)
    uint64_t any = 0;
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const int r = q * 256 + tid;
        L[q] = r < m ? look[r] : 0ull;
        M[q] = 0ull;
        const bool unpivoted = r < m && pivcol[r] < 0;
        unpm[q] = __builtin_amdgcn_ballot_w64(unpivoted);
        any |= unpivoted ? L[q] : 0ull;
    }
    // Columns no unpivoted row has a bit in NOW never become pivot columns in this block (a row only changes by taking rows that
    // were unpivoted at block start, and those all have a zero there): they cost no step.  (Keeping that OR current step by step
    // would catch the other non-pivot columns as well, but costs more than the steps it saves: measured.)
    any = (uint64_t)__reduce_or_sync(~0ull, (unsigned)any) | ((uint64_t)__reduce_or_sync(~0ull, (unsigned)(any >> 32)) << 32);
    if (lane == 0 && any) atomicOr(xch + 20, (unsigned long long)any);  // (zeroed by the caller before its last barrier)
    __syncthreads();
    const uint64_t live = __builtin_amdgcn_readfirstlane((unsigned)xch[20]) | ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(xch[20] >> 32)) << 32);
    int np = 0, step = 0;
    // A step is one dependent chain (~8 cycles an instruction, contention or not): every instruction off it counts.  The test
    // "row has the bit" is made once per register row; who offers is then scalar arithmetic on its ballot.
    for (int j = 0; j < ahead && rank < max_rank; ++j) {
        if (!((live >> j) & 1ull)) continue;
        const uint64_t jb = 1ull << j;
        bool h[MQ];
        int qs = -1;
        uint64_t cm = 0;
#pragma unroll
        for (int q = 0; q < MQ; ++q) {
            h[q] = (L[q] & jb) != 0;
            const uint64_t c = __builtin_amdgcn_ballot_w64(h[q]) & unpm[q];
            if (qs < 0 && c) { qs = q; cm = c; }  // this wavefront's offer: its first unpivoted row with the bit (rows ascend in (q, lane))
        }
        // offers, two sets (the next step writes the other one): words 0-1 = the four wavefronts' rows (u32 each, ~0: none),
        // words 2 + 2 w, 3 + 2 w = wavefront w's row in the block's columns and its mask
        unsigned long long *set = xch + (step & 1) * 10;
        ++step;
        if (qs < 0) {
            if (lane == 0) reinterpret_cast<unsigned *>(set)[wave] = ~0u;
        } else if (lane == __builtin_ctzll(cm)) {
            reinterpret_cast<unsigned *>(set)[wave] = (unsigned)(qs * 256 + tid);
#pragma unroll
            for (int q = 0; q < MQ; ++q)
                if (q == qs) { set[2 + 2 * wave] = L[q]; set[3 + 2 * wave] = M[q]; }  // (qs is scalar: one of these runs)
        }
        __syncthreads();
        const uint4 rows = *reinterpret_cast<const uint4 *>(set);
        const unsigned long long l0 = set[2], m0 = set[3], l1 = set[4], m1 = set[5], l2 = set[6], m2 = set[7], l3 = set[8], m3 = set[9];
        const unsigned r0 = __builtin_amdgcn_readfirstlane(rows.x), r1 = __builtin_amdgcn_readfirstlane(rows.y);
        const unsigned r2 = __builtin_amdgcn_readfirstlane(rows.z), r3 = __builtin_amdgcn_readfirstlane(rows.w);
        const unsigned r01 = r0 < r1 ? r0 : r1, r23 = r2 < r3 ? r2 : r3;
        const int p = (int)(r01 < r23 ? r01 : r23);
        if (p < 0) continue;  // (~0: nobody has the bit) not a pivot column
        const int pw = (p >> 6) & 3;  // the wavefront that made the offer
        const uint64_t lp = pw == 0 ? l0 : pw == 1 ? l1 : pw == 2 ? l2 : l3;
        const uint64_t mp = pw == 0 ? m0 : pw == 1 ? m1 : pw == 2 ? m2 : m3;
        const uint64_t take = mp ^ (1ull << np);
        const int pq = p >> 8, pt = p & 255;
        const bool mine = tid == pt;
#pragma unroll
        for (int q = 0; q < MQ; ++q)
            if (h[q] && !(q == pq && mine)) { L[q] ^= lp; M[q] ^= take; }
        if (wave == pw) {
#pragma unroll
            for (int q = 0; q < MQ; ++q)
                if (q == pq) unpm[q] &= ~(1ull << (pt & 63));
        }
        if (tid == 0) { blk_row[np] = (uint16_t)p; blk_col[np] = (uint16_t)j; }  // (the column number is looked up afterwards: no LDS read on this path)
        ++np;
        ++rank;
    }
#pragma unroll
    for (int q = 0; q < MQ; ++q) {
        const int r = q * 256 + tid;
        if (r < m) look[r] = M[q];  // the look-ahead words are spent; the next block gathers its own
    }
    return np | (step << 8);  // pivots made, columns that took a step
}

template <bool HIGHER, bool MAT_LDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) osd_big_kernel(const OsdBigArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    const OsdArgs &a = A.o;
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int m = a.m, n = a.n, HW = A.hwords, P = A.pow2;
    // [pivcol, hits, sy | ord] stay; the room after them (extra_off) belongs to one phase at a time: the sort's keys, the fill's
    // positions, the elimination's look-ahead words and table, the candidates' tables and planes (sized by the host, decode path of bp_hip.hip)
    uint64_t *keys = reinterpret_cast<uint64_t *>(osd_lds + (size_t)A.extra_off);
    // (16-bit tables: m, n < 32768 here -- the host checks --, and the LDS they save is a third resident workgroup)
    uint16_t *ord = reinterpret_cast<uint16_t *>(osd_lds + (size_t)a.lds_per_wave);  // lds_per_wave: bytes before `ord`
    int16_t *pivcol = reinterpret_cast<int16_t *>(osd_lds);            // [m]
    uint16_t *hits = reinterpret_cast<uint16_t *>(pivcol + m);         // [m]
    uint8_t *sy = reinterpret_cast<uint8_t *>(hits + m);               // [m + 1]
    uint64_t *look = reinterpret_cast<uint64_t *>(osd_lds + (size_t)A.extra_off);  // [m] bits of the block's 64 columns, per row
    int16_t *colinfo = reinterpret_cast<int16_t *>(osd_lds + (size_t)A.extra_off);  // [n] pivot column: its row; q-th non-pivot column: -1 - q
    uint64_t *npm = reinterpret_cast<uint64_t *>(osd_lds + (((size_t)A.extra_off + 2 * (size_t)n + 7) & ~(size_t)7));  // [HW] non-pivot positions of every plane, then [HW][6] the compress moves
    uint64_t *planes = npm + 7 * (size_t)HW;  // [4][m + 1] (entry m: the all-zero dummy row)
    __shared__ int sh_row, sh_pivot[3], sh_nhits[3], sh_cnt[4], sh_nact;
    __shared__ uint16_t blk_row[64], blk_col[64];
    __shared__ unsigned long long blk_sy;
    __shared__ __attribute__((aligned(16))) unsigned long long blk_xch[21];  // [20]: OR of the unpivoted rows' look-ahead words
    __shared__ double sh_w[4];
    __shared__ long sh_c[4];
    // MAT_LDS: the working copy of H fits LDS next to everything else (mid-size matrices: a whole workgroup on one
    // syndrome where the one-wavefront kernels would leave a CU with one or two wavefronts); the slot then only holds T
    uint64_t *slot = A.scratch + (int64_t)blockIdx.x * A.slot_stride;
    uint64_t *mat = MAT_LDS ? reinterpret_cast<uint64_t *>(osd_lds + (size_t)A.mat_off) : slot;
    uint64_t *Tm = MAT_LDS ? slot : slot + (int64_t)HW * m;  // [kwords][m]

    for (bool first = true;; first = false) {
        if (tid == 0) {  // (the first row without a visit to the work counter: see osd_first_row)
            const unsigned count = osd_list_count(a);
            unsigned idx = blockIdx.x;
            if (!first) idx = gridDim.x >= count ? count : gridDim.x + atomicAdd(&a.counters[1], 1u);
            sh_row = idx < count ? a.list[idx] : -1;
        }
        __syncthreads();
        const int64_t b = sh_row;
        if (b < 0) return;
        OSD_CLK_START();
        for (int64_t e = tid; e < (int64_t)HW * m; e += T) mat[e] = 0ull;  // (filled after the sort)
        for (int j = tid; j < n; j += T) keys[j] = osd_sort_key(a.llr[b * n + j]);
        for (int j = tid; j < P; j += T) ord[j] = (uint16_t)j;
        __syncthreads();
        // soft_decision_col_sort (sort.hpp:48-62): ascending key, ties by column number; numbers >= n pad the network.
        // A pass compares P / 2 disjoint pairs: pair p = the two positions whose numbers are p with a zero / one inserted at bit
        // log2(j).  A wavefront owns P / 8 consecutive pairs -- for j < P / 4 those are the positions of its own quarter of the
        // array, so such passes (all but six of a 2048-position sort's 66) need no workgroup barrier, only the in-order LDS queue of the
        // wavefront; and a thread's pairs are fetched four at a time (order entries, then the keys they point at) instead of one
        // dependent chain after the other.
        {
            const bool chunked = P >= 512;
            const int per_lane = chunked ? P >> 9 : (P / 2 + T - 1) / T;  // pairs per thread
            for (int k = 2; k <= P; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int e0 = 0; e0 < per_lane; e0 += 4) {
                        int pi[4], pl[4], x[4], y[4];
                        uint64_t kx[4], ky[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int e = e0 + u;
                            const int pr = chunked ? wave * (P >> 3) + lane + 64 * e : tid + T * e;
                            const bool on = e < per_lane && pr < (P >> 1);
                            pi[u] = on ? ((pr & ~(j - 1)) << 1) | (pr & (j - 1)) : -1;
                            pl[u] = pi[u] | j;
                            x[u] = on ? ord[pi[u]] : 0;
                            y[u] = on ? ord[pl[u]] : 0;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            kx[u] = x[u] < n ? keys[x[u]] : ~0ull;
                            ky[u] = y[u] < n ? keys[y[u]] : ~0ull;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool y_first = ky[u] < kx[u] || (ky[u] == kx[u] && y[u] < x[u]);
                            if (pi[u] >= 0 && y_first == ((pi[u] & k) == 0)) { ord[pi[u]] = (uint16_t)y[u]; ord[pl[u]] = (uint16_t)x[u]; }
                        }
                    }
                    const int jn = j > 1 ? j >> 1 : k;  // the next pass's distance (k: the next stage starts there)
                    if (!chunked || j >= (P >> 2) || jn >= (P >> 2) || (k == P && j == 1)) {
                        __syncthreads();
                    } else {
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS operations of a wavefront execute in order; this only pins the compiler
                        __builtin_amdgcn_wave_barrier();
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    }
                }
        }
        // Working copy of H with its COLUMNS IN SORTED ORDER, word-plane major: bit (t & 63) of mat[(t >> 6) * m + i] = H[i][ord[t]].
        // The 64 columns of an elimination block are then one plane (one coalesced read per row instead of 64 scattered ones: the
        // copies of a batch's rows live in HBM / MALL, and bytes moved are what this kernel is bound by), and the non-pivot
        // columns come out in candidate order.  Filled from the CSR form: nnz atomic ORs into the zeroed planes.
        {
            uint16_t *pos = reinterpret_cast<uint16_t *>(osd_lds + (size_t)A.extra_off);  // [n] sorted position of a column (in the keys' room, until the fill is done)
            for (int t = tid; t < n; t += T) pos[ord[t]] = (uint16_t)t;
            __threadfence();
            __syncthreads();
            for (int i = tid; i < m; i += T)
                for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) {
                    const int t = pos[a.col_idx[e]];
                    atomicOr(reinterpret_cast<unsigned long long *>(&mat[(int64_t)(t >> 6) * m + i]), 1ull << (t & 63));
                }
            __threadfence();
            __syncthreads();
        }
        OSD_WG_CLK(1);  // sort + working copy
        for (int i = tid; i < m; i += T) { pivcol[i] = -1; sy[i] = a.synd[b * m + i] ? 1 : 0; }
        if (tid < 3) { sh_nhits[tid] = 0; sh_pivot[tid] = INT32_MAX; }
        __syncthreads();

        int rank = 0;
        if (A.pbuf_off >= 0) {
            // ---- blocked: 64 sorted columns per round (osd_block_eliminate above) ----
            // tbl [16][OSD_PIECE][16]: for the planes of the round in hand, every XOR combination of each group of four pivot rows as
            // they were at block start -- a row then takes ceil(pivots / 4) table entries per plane whatever its mask
            uint64_t *tbl = reinterpret_cast<uint64_t *>(osd_lds + (size_t)A.pbuf_off);
            for (int t = 0; t < n && rank < A.max_rank; t += 64) {
                const int ahead = n - t < 64 ? n - t : 64;
                int pending = 0;
                for (int i = tid; i < m; i += T) {
                    look[i] = mat[(t >> 6) * m + i];  // the block's columns: one plane of the sorted copy
                    if (!HIGHER && pivcol[i] < 0 && sy[i]) pending = 1;
                }
                if (tid == 0) { blk_sy = 0ull; sh_nact = 0; blk_xch[20] = 0ull; }
                if (!HIGHER) {
                    // OSD-0 stops once the syndrome is in the span of the pivots (gf2sparse_linalg.hpp:373-383).  Tested per block:
                    // pivots made past that point have a zero syndrome bit and change no other, so x is the same.
                    if (!__syncthreads_or(pending)) break;
                } else {
                    __syncthreads();
                }
                OSD_WG_CLK(0);  // blocked: the block's plane
                int npv;  // (pivots | steps << 8)
                if (m <= 256) npv = osd_block_eliminate<1>(tid, m, ahead, look, pivcol, blk_row, blk_col, blk_xch, rank, A.max_rank);
                else if (m <= 512) npv = osd_block_eliminate<2>(tid, m, ahead, look, pivcol, blk_row, blk_col, blk_xch, rank, A.max_rank);
                else if (m <= 768) npv = osd_block_eliminate<3>(tid, m, ahead, look, pivcol, blk_row, blk_col, blk_xch, rank, A.max_rank);
                else if (m <= 1024) npv = osd_block_eliminate<4>(tid, m, ahead, look, pivcol, blk_row, blk_col, blk_xch, rank, A.max_rank);
                else if (m <= 1536) npv = osd_block_eliminate<6>(tid, m, ahead, look, pivcol, blk_row, blk_col, blk_xch, rank, A.max_rank);
                else npv = osd_block_eliminate<8>(tid, m, ahead, look, pivcol, blk_row, blk_col, blk_xch, rank, A.max_rank);
                __syncthreads();
                OSD_WG_CLK(6);  // blocked: the block's pivots
                const int nsteps = npv >> 8;
                npv &= 255;
                (void)nsteps;
#ifdef LDPC_HIP_OSD_CLOCKS
                if (tid == 0) atomicAdd(&osd_phase_clocks[2], 1000000ull * nsteps + 1000000000ull * npv);  // (columns, pivots: read off the decimal digits)
#endif
                rank += npv;
                if (npv == 0) continue;
                if (tid < npv && sy[blk_row[tid]]) atomicOr(&blk_sy, 1ull << tid);
                const int groups = (npv + 3) >> 2;
                // OSD-0 never looks at a column again once its block is done: only the planes ahead follow the rows
                const int wfirst = HIGHER ? 0 : (t >> 6) + 1, wspan = HW - wfirst;
                const int rounds = (wspan + OSD_PIECE - 1) / OSD_PIECE, pw = rounds > 0 ? (wspan + rounds - 1) / rounds : 1;  // planes per round, <= OSD_PIECE
                // table of a round: thread (group g, plane q) reads the group's four pivot rows at that plane -- as they are at block
                // start, and a round only changes its own planes: the reads for a round are issued a round ahead
                const int tg = tid / pw, tq = tid - tg * pw;
                uint64_t xt[4] = {0, 0, 0, 0};
                auto table_rows = [&](int w0) {
#pragma unroll
                    for (int bit = 0; bit < 4; ++bit) {
                        const int j = 4 * tg + bit;
                        xt[bit] = (tid < groups * pw && j < npv && w0 + tq < HW) ? mat[(w0 + tq) * m + blk_row[j]] : 0ull;
                    }
                };
                if (wfirst < HW) table_rows(wfirst);
                OSD_LDS_BARRIER();
                const unsigned long long blk_sy_now = blk_sy;  // S: the pivot rows' syndrome bits at block start
                const unsigned long long smask = blk_sy_now;
                // The rows that take any of the block's pivots, listed: for sparse H they are a fraction of the rows, and a wavefront
                // whose lanes each own fixed rows would run the whole update for the few lanes that have one.
#pragma unroll
                for (int k = 0; k < OSD_BLOCK_ROWS / 256; ++k) {
                    const int r = k * 256 + tid;
                    if (k * 256 >= m) break;
                    const uint64_t Mr = r < m ? look[r] : 0ull;
                    const uint64_t am = __builtin_amdgcn_ballot_w64(Mr != 0);
                    if (!am) continue;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&sh_nact, __builtin_popcountll(am));
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (Mr) {
                        hits[base + __builtin_popcountll(am & ((1ull << lane) - 1ull))] = (uint16_t)r;
                        sy[r] ^= (uint8_t)(__builtin_popcountll(Mr & smask) & 1);
                    }
                }
                OSD_LDS_BARRIER();
                const int nact = sh_nact;
                const float inv_nact = 1.0f / (float)(nact > 0 ? nact : 1);
                OSD_WG_CLK(8);  // update: list of the rows
                for (int w0 = wfirst; w0 < HW && nact > 0; w0 += pw) {
                    if (tid < groups * pw) {  // the 16 combinations of the group's four rows
                        const uint64_t *x = xt;
                        uint64_t *e = tbl + (tg * OSD_PIECE + tq) * 16;
                        const uint64_t x01 = x[0] ^ x[1], x23 = x[2] ^ x[3];
                        e[0] = 0ull;        e[1] = x[0];         e[2] = x[1];         e[3] = x01;
                        e[4] = x[2];        e[5] = x[2] ^ x[0];  e[6] = x[2] ^ x[1];  e[7] = x[2] ^ x01;
                        e[8] = x[3];        e[9] = x[3] ^ x[0];  e[10] = x[3] ^ x[1]; e[11] = x[3] ^ x01;
                        e[12] = x23;        e[13] = x23 ^ x[0];  e[14] = x23 ^ x[1];  e[15] = x23 ^ x01;
                    }
                    OSD_WG_CLK(9);  // update: table build
                    OSD_LDS_BARRIER();  // (the previous round's stores may still be in flight: other planes)
                    OSD_WG_CLK(10);  // update: barrier
                    if (w0 + pw < HW) table_rows(w0 + pw);
                    // items (listed row, plane of the round), neighbouring lanes on neighbouring listed rows of one plane; eight at a time:
                    // what each takes from the table first, then only the words that change are read and written back
                    const int pwr = HW - w0 < pw ? HW - w0 : pw, items = nact * pwr;
                    for (int i0 = tid; i0 < items; i0 += 8 * 256) {
                        uint64_t d[8];
                        int at[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int i = i0 + u * 256;
                            d[u] = 0ull;
                            at[u] = 0;
                            if (i < items) {
                                int q = (int)((float)i * inv_nact);  // i / nact, i < 2^14: the float quotient is off by one at most
                                if (q * nact > i) --q;
                                if ((q + 1) * nact <= i) ++q;
                                const int r = hits[i - q * nact];
                                const uint64_t Mr = look[r];
                                const uint64_t *e = tbl + q * 16;
                                uint64_t acc = 0ull;
                                for (int g = 0; g < groups; ++g) acc ^= e[g * (OSD_PIECE * 16) + ((Mr >> (g << 2)) & 15ull)];
                                d[u] = acc;
                                at[u] = (w0 + q) * m + r;  // (plane-major index < 2^24: m, n < 32768)
                            }
                        }
                        uint64_t x[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) x[u] = d[u] ? mat[at[u]] : 0ull;
#pragma unroll
                        for (int u = 0; u < 8; ++u)
                            if (d[u]) mat[at[u]] = x[u] ^ d[u];
                    }
                    OSD_WG_CLK(11);  // update: items
                    // the next round rewrites the table: an LDS-only barrier -- the stores above need not have landed, nobody reads those
                    // planes before the next full barrier (the next round's table is built from planes further on)
                    OSD_LDS_BARRIER();
                    OSD_WG_CLK(12);  // update: closing barrier
                }
                if (tid < npv) pivcol[blk_row[tid]] = (int16_t)ord[t + blk_col[tid]];
                __syncthreads();
                OSD_WG_CLK(7);  // blocked: the combination tables + the combined update of every row
            }
        } else
        // One column per step, two barriers when it yields a pivot, one when not.  The hit counter and the pivot slot
        // exist three times over: step t uses copy t % 3 and, once past its first barrier, re-arms copy (t + 2) % 3,
        // which nobody has touched since step t - 1 and nobody will before step t + 2.
        // Which rows have a one in the column of a step is not read from the planes step by step (a dependent round
        // trip to L2 / MALL each time when H lives in HBM): every 64 steps the plane of the NEXT 64 columns is read into one
        // word per row, and from then on the owner of a row keeps that word current -- a row that takes the pivot row
        // takes the pivot row's word as well.
        for (int t = 0; t < n && rank < A.max_rank; ++t) {
            const int c = ord[t], cur = t % 3, kk = t & 63;
            if (kk == 0) {
                for (int i = tid; i < m; i += T) look[i] = mat[(int64_t)(t >> 6) * m + i];  // (columns in sorted order: a plane)
                __syncthreads();
            }
            int pending = 0;
            for (int i = tid; i < m; i += T) {
                const bool unpivoted = pivcol[i] < 0;
                if ((look[i] >> kk) & 1ull) {
                    hits[atomicAdd(&sh_nhits[cur], 1)] = (uint16_t)i;
                    if (unpivoted) atomicMin(&sh_pivot[cur], i);
                }
                if (!HIGHER && unpivoted && sy[i]) pending = 1;
            }
            if (!HIGHER) {
                // stop once the syndrome is in the span of the pivots (gf2sparse_linalg.hpp:373-383): the test the
                // reference makes after a pivot is made here before the next one -- the same state
                if (!__syncthreads_or(pending)) break;
            } else {
                __syncthreads();
            }
            const int p = sh_pivot[cur], nh = sh_nhits[cur];
            if (tid == 0) { sh_nhits[(t + 2) % 3] = 0; sh_pivot[(t + 2) % 3] = INT32_MAX; }
            if (p == INT32_MAX) continue;  // no unpivoted row has the bit
            const uint8_t psy = sy[p];
            // one (row, 8 planes) piece per thread: the loads are in flight together, and a step with few hit rows
            // still spreads over the workgroup; the pivot row is read in place (nothing writes it in this step)
            const int pieces = (HW + 7) >> 3;
            for (int item = tid; item < nh * pieces; item += T) {
                const int hI = item / pieces, w0 = (item - hI * pieces) << 3;
                const int r = hits[hI];
                if (r == p) continue;
                uint64_t v[8], pw[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const bool in = w0 + q < HW;
                    v[q] = in ? mat[(int64_t)(w0 + q) * m + r] : 0ull;
                    pw[q] = in ? mat[(int64_t)(w0 + q) * m + p] : 0ull;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (pw[q]) mat[(int64_t)(w0 + q) * m + r] = v[q] ^ pw[q];
                if (w0 == 0) sy[r] ^= psy;
            }
            {   // the look-ahead words follow the rows (look[p] itself is read only: row p is not a target)
                const uint64_t plook = look[p];
                for (int i = tid; i < m; i += T)
                    if (i != p && ((look[i] >> kk) & 1ull)) look[i] ^= plook;
            }
            if (tid == 0) pivcol[p] = (int16_t)c;
            ++rank;
            __syncthreads();
        }
        __syncthreads();
        if (!HIGHER) {
            // x = 0 except on the pivot columns, where it is the reduced syndrome bit of the pivot's row (lu_solve, :237-288)
            for (int j = tid; j < n; j += T) a.decoding[b * n + j] = 0;
            __syncthreads();
            for (int i = tid; i < m; i += T)
                if (pivcol[i] >= 0 && sy[i]) a.decoding[b * n + pivcol[i]] = 1;
            __syncthreads();
            continue;
        }

        OSD_WG_CLK(2);  // elimination
        // ---- higher order (osd.hpp:119-187) ----
        for (int j = tid; j < n; j += T) colinfo[j] = INT16_MIN;
        __syncthreads();
        for (int i = tid; i < m; i += T)
            if (pivcol[i] >= 0) colinfo[pivcol[i]] = (int16_t)i;
        __syncthreads();
        int k = 0;  // non-pivot columns in sorted order (`cols[rank ..]`, gf2sparse_linalg.hpp:210-224)
        for (int t0 = 0; t0 < n; t0 += T) {
            const int t = t0 + tid;
            const int c = t < n ? ord[t] : 0;
            const bool np = t < n && colinfo[c] == INT16_MIN;
            const uint64_t mask = __ballot(np);
            if (lane == 0) sh_cnt[wave] = __builtin_popcountll(mask);
            __syncthreads();
            int before = 0, total = 0;
            for (int w = 0; w < 4; ++w) { before += w < wave ? sh_cnt[w] : 0; total += sh_cnt[w]; }
            if (np) {
                const int q = k + before + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                colinfo[c] = (int16_t)(-1 - q);
            }
            if (lane == 0 && (t0 >> 6) + wave < HW) npm[(t0 >> 6) + wave] = mask;  // this wavefront's 64 sorted positions are one plane
            k += total;
            __syncthreads();
        }
        const int KW = (k + 63) >> 6;  // <= A.kwords by the host's sizing
        OSD_WG_CLK(3);  // numbering
        // T: the reduced rows on the non-pivot columns (bit q = the q-th of them), plane v = word v of every row.  With the
        // columns in sorted order that is every plane squeezed to its non-pivot positions (a six-step parallel bit compress,
        // the moves worked out once per plane) and the pieces laid end to end: one read of the matrix.
        for (int w = tid; w < HW; w += T) {
            uint64_t mm = npm[w], mk = ~mm << 1;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                uint64_t mp = mk ^ (mk << 1);
                mp ^= mp << 2; mp ^= mp << 4; mp ^= mp << 8; mp ^= mp << 16; mp ^= mp << 32;
                const uint64_t mv = mp & mm;
                npm[HW + w * 6 + i] = mv;
                mm = (mm ^ mv) | (mv >> (1 << i));
                mk &= ~mp;
            }
        }
        __syncthreads();
        for (int r0 = 0; r0 < m; r0 += 1024) {  // four rows per thread at a time
            int rr[4];
            bool piv[4];
            uint64_t acc[4] = {0, 0, 0, 0}, x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                rr[q] = r0 + q * 256 + tid < m ? r0 + q * 256 + tid : m - 1;
                piv[q] = r0 + q * 256 + tid < m && pivcol[rr[q]] >= 0;
                x[q] = r0 + q * 256 < m ? mat[rr[q]] : 0ull;
            }
            int fill = 0, v = 0;
            for (int w = 0; w < HW; ++w) {
                uint64_t cur[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { cur[q] = x[q]; x[q] = (w + 1 < HW && r0 + q * 256 < m) ? mat[(w + 1) * m + rr[q]] : 0ull; }  // (next plane in flight)
                const uint64_t mask = npm[w];
                const int cnt = __builtin_popcountll(mask);
                if (cnt == 0) continue;
                uint64_t mv[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) mv[i] = npm[HW + w * 6 + i];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    uint64_t y = cur[q] & mask;
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        const uint64_t tt = y & mv[i];
                        y = (y ^ tt) | (tt >> (1 << i));
                    }
                    cur[q] = y;
                    acc[q] |= y << fill;
                }
                if (fill + cnt >= 64) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (r0 + q * 256 + tid < m) Tm[(int64_t)v * m + r0 + q * 256 + tid] = piv[q] ? acc[q] : 0ull;
                        acc[q] = fill ? cur[q] >> (64 - fill) : 0ull;
                    }
                    ++v;
                    fill += cnt - 64;
                } else {
                    fill += cnt;
                }
            }
            if (fill > 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (r0 + q * 256 + tid < m) Tm[(int64_t)v * m + r0 + q * 256 + tid] = piv[q] ? acc[q] : 0ull;
            }
        }
        __syncthreads();
        // x_i of a candidate: pivot column with row r: S_r ^ parity(T_r & candidate); q-th non-pivot column: the candidate's bit q.
        // Weights are added in column order (osd.hpp:171-176); `acc += bit ? w : 0.0` adds the same numbers.
        OSD_WG_CLK(4);  // T planes
        // the weights are the same for every lane, wavefront and workgroup and nobody writes them: through the constant address
        // space they come by scalar loads (a vector load would move 64 copies of each through L1)
        const __attribute__((address_space(4))) double *wt = (const __attribute__((address_space(4))) double *)a.wt;
        // Branch-free: a non-pivot column reads the all-zero dummy row m and adds its own term; for the one-column candidates
        // S is folded into the staged plane.
        // OSD_TRIP columns per trip, written out (the compiler does not unroll these loops on request): a trip waits for an LDS
        // read (column info), a dependent LDS read (plane words) and an L1 / L2 read (weights) in turn, and that wait is what a
        // trip costs -- the more columns share it the better; the additions themselves stay a serial chain in column order.
        auto weigh_single = [&](const uint64_t *pl, int q, bool live) -> double {  // candidate: non-pivot column q alone; pl[r] = T plane q / 64 of row r, XOR all-ones if S_r
            double acc = 0;
            const int sh = q & 63;
            int i = 0;
            for (; i + OSD_TRIP <= n; i += OSD_TRIP) {
                int ci[OSD_TRIP];
                uint64_t t[OSD_TRIP];
                double w[OSD_TRIP];
#pragma unroll
                for (int u = 0; u < OSD_TRIP; ++u) ci[u] = __builtin_amdgcn_readfirstlane(colinfo[i + u]);  // the same for every lane
#pragma unroll
                for (int u = 0; u < OSD_TRIP; ++u) { t[u] = pl[ci[u] >= 0 ? ci[u] : m]; w[u] = wt[i + u]; }
#pragma unroll
                for (int u = 0; u < OSD_TRIP; ++u) {
                    const bool bit = (((t[u] >> sh) & 1ull) != 0) || (live && -1 - ci[u] == q);
                    acc += bit ? w[u] : 0.0;
                }
            }
            for (; i < n; ++i) {
                const int ci = __builtin_amdgcn_readfirstlane(colinfo[i]);
                const bool bit = (((pl[ci >= 0 ? ci : m] >> sh) & 1ull) != 0) || (live && -1 - ci == q);
                acc += bit ? wt[i] : 0.0;
            }
            return acc;
        };
        auto weigh_mask = [&](const uint64_t *pl0, uint64_t mask) -> double {  // candidate: a set of the first 64 non-pivot columns; pl0 = T plane 0
            // (bitwise, not `||`: with the short-circuit form and several columns per trip this compiler drops the weight of
            // the lanes whose bit comes from the second term)
            double acc = 0;
            int i = 0;
            for (; i + OSD_TRIP <= n; i += OSD_TRIP) {
                int ci[OSD_TRIP], r[OSD_TRIP];
                uint64_t t[OSD_TRIP];
                unsigned sv[OSD_TRIP];
                double w[OSD_TRIP];
#pragma unroll
                for (int u = 0; u < OSD_TRIP; ++u) ci[u] = __builtin_amdgcn_readfirstlane(colinfo[i + u]);
#pragma unroll
                for (int u = 0; u < OSD_TRIP; ++u) { r[u] = ci[u] >= 0 ? ci[u] : m; t[u] = pl0[r[u]]; sv[u] = sy[r[u]]; w[u] = wt[i + u]; }
#pragma unroll
                for (int u = 0; u < OSD_TRIP; ++u) {
                    const int qn = ci[u] >= 0 ? 64 : -1 - ci[u];
                    const unsigned own = (unsigned)((mask >> (qn & 63)) & 1ull) & (qn < 64 ? 1u : 0u);
                    const unsigned bit = (((unsigned)__builtin_popcountll(t[u] & mask) + sv[u]) & 1u) | own;
                    acc += bit ? w[u] : 0.0;
                }
            }
            for (; i < n; ++i) {
                const int ci = __builtin_amdgcn_readfirstlane(colinfo[i]);
                const int r = ci >= 0 ? ci : m, qn = ci >= 0 ? 64 : -1 - ci;
                const unsigned own = (unsigned)((mask >> (qn & 63)) & 1ull) & (qn < 64 ? 1u : 0u);
                const unsigned bit = (((unsigned)__builtin_popcountll(pl0[r] & mask) + (unsigned)sy[r]) & 1u) | own;
                acc += bit ? wt[i] : 0.0;
            }
            return acc;
        };
        auto pair_mask = [&](long p, bool &valid) -> uint64_t {  // pairs (i, j), i < j < order <= 64, i-major (osd.hpp:91-99)
            int i = 0;
            while (p >= a.order - 1 - i) { p -= a.order - 1 - i; ++i; }
            const int j = i + 1 + (int)p;
            valid = j < k;
            return valid ? (1ull << i) | (1ull << j) : 0ull;
        };
        const uint64_t kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
        const long npairs = (long)a.order * (a.order - 1) / 2;
        // The winner is the lightest candidate, the earliest of the reference's list among equals, the OSD-0 solution before all of
        // them (osd.hpp:131-136, 171-181: a candidate replaces the incumbent only if strictly lighter) -- a lexicographic minimum, so
        // the candidates can be weighed in any order.  Tasks of 64 candidates, one per wavefront at a time, each wavefront staging the
        // plane it needs in its own buffer and running at its own pace: first the sets of the first `order` non-pivot columns (entry 0
        // = the empty set = the OSD-0 solution, index -1; then the pairs of OSD_CS, index k + pair, or the numbers of OSD_E), then
        // for OSD_CS the single columns, plane by plane (index q).
        // OSD_CS with osd_order > 64: the pairs reach past T plane 0 -- they are weighed plane pair by plane pair further down, and the
        // "sets" here shrink to the empty set (the OSD-0 solution)
        const bool far_pairs = a.method == 3 && a.order > 64;
        const long nsets = (a.method == 3 ? (far_pairs ? 0 : npairs) : (1L << a.order) - 1) + 1;  // (numbers 1 .. 2^order - 1, order <= 24, bits >= k dropped: util.hpp:12-38)
        const long nchunk = (nsets + 63) >> 6, ntask = nchunk + (a.method == 3 ? KW : 0);
        if (tid == 0) sy[m] = 0;  // the dummy row of the non-pivot columns
        __syncthreads();
        double best_w = __builtin_huge_val();
        long best_c = LONG_MAX;  // index in the reference's candidate list; -1: the OSD-0 solution
        auto offer = [&](double w, long c) { if (w < best_w || (w == best_w && c < best_c)) { best_w = w; best_c = c; } };
        {
            uint64_t *pl = planes + (size_t)wave * (m + 1);
            int held = -2;  // what the buffer holds: -1 = T plane 0 as it is (sets), v >= 0 = plane v XOR all-ones where S_r (single columns)
            for (long task0 = 0; task0 < ntask; task0 += A.nplanes) {
                const long task = task0 + __builtin_amdgcn_readfirstlane(wave);
                if (wave >= A.nplanes || task >= ntask) break;  // (fewer buffers than wavefronts where LDS is short: the others wait below)
                const int want = task < nchunk ? -1 : (int)(task - nchunk);
                if (want != held) {
                    if (want < 0) {
                        for (int r = lane; r < m; r += 64) pl[r] = KW > 0 ? Tm[r] : 0ull;
                    } else {
                        for (int r = lane; r < m; r += 64) pl[r] = Tm[(int64_t)want * m + r] ^ (sy[r] ? ~0ull : 0ull);
                    }
                    if (lane == 0) pl[m] = 0;
                    held = want;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS operations of a wavefront execute in order; this only pins the compiler
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                if (want < 0) {
                    const long e = task * 64 + lane;
                    bool valid = e < nsets;
                    uint64_t mask = 0;
                    long idx = -1;
                    if (valid && e >= 1) {
                        if (a.method == 3) { mask = pair_mask(e - 1, valid); idx = k + e - 1; }
                        else { mask = (uint64_t)e & kmask; idx = e - 1; }
                    }
                    const double w = weigh_mask(pl, mask);
                    if (valid) offer(w, idx);
                } else {
                    const int q = 64 * want + lane;
                    const bool live = q < k;
                    const double w = weigh_single(pl, q, live);
                    if (live) offer(w, q);
                }
            }
            if (far_pairs && wave < A.nplanes) {
                // pairs (i, j), i < j < min(osd_order, k), block by block of T planes (vi <= vj): plane vi staged in this wavefront's
                // buffer, plane vj read from the T array itself -- column by column both words are the same for all 64 lanes (one
                // broadcast LDS read, one single-line memory read); a lane's pair picks its two bits.  Index in the reference's list:
                // k + i (order - 1) - i (i - 1) / 2 + (j - i - 1) (i-major, osd.hpp:91-99).
                const int reach = a.order < k ? a.order : k;
                const int P = (reach + 63) >> 6;
                long task = 0;
                for (int vi = 0; vi < P; ++vi) {
                    const int cnt_i = reach - 64 * vi < 64 ? reach - 64 * vi : 64;
                    for (int vj = vi; vj < P; ++vj) {
                        const int cnt_j = reach - 64 * vj < 64 ? reach - 64 * vj : 64;
                        const int npb = vi == vj ? cnt_i * (cnt_i - 1) / 2 : cnt_i * cnt_j;
                        for (int e0 = 0; e0 < npb; e0 += 64, ++task) {
                            if ((int)(task % A.nplanes) != wave) continue;
                            if (held != -3 - vi) {
                                for (int r = lane; r < m; r += 64) pl[r] = Tm[(int64_t)vi * m + r];
                                if (lane == 0) pl[m] = 0;
                                held = -3 - vi;
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                            }
                            const int e = e0 + lane;
                            const bool valid = e < npb;
                            int ii = 0, jj = 1;
                            if (valid) {
                                if (vi != vj) { ii = e / cnt_j; jj = e - ii * cnt_j; }
                                else { int pp = e; while (pp >= cnt_i - 1 - ii) { pp -= cnt_i - 1 - ii; ++ii; } jj = ii + 1 + pp; }
                            }
                            const int qi = 64 * vi + ii, qj = 64 * vj + jj;
                            const uint64_t *Tj = Tm + (int64_t)vj * m;
                            double acc = 0;
                            for (int i = 0; i < n; ++i) {
                                const int ci = __builtin_amdgcn_readfirstlane(colinfo[i]);
                                unsigned bit;
                                if (ci >= 0) bit = (unsigned)(((pl[ci] >> ii) ^ (Tj[ci] >> jj) ^ (uint64_t)sy[ci]) & 1ull);
                                else bit = (unsigned)(-1 - ci == qi) | (unsigned)(-1 - ci == qj);
                                acc += bit ? wt[i] : 0.0;
                            }
                            if (valid) offer(acc, (long)k + (long)qi * (a.order - 1) - (long)qi * (qi - 1) / 2 + (qj - qi - 1));
                        }
                    }
                }
            }
        }
        OSD_WG_CLK(5);  // weighing
        // lightest, then earliest: across the lanes of a wavefront, then across the wavefronts
        for (int off = 32; off > 0; off >>= 1) {
            const double ow = __shfl_xor(best_w, off);
            const long oc = ((long)__shfl_xor((int)(best_c >> 32), off) << 32) | (unsigned)__shfl_xor((int)(best_c & 0xffffffff), off);
            offer(ow, oc);
        }
        __syncthreads();
        if (lane == 0) { sh_w[wave] = best_w; sh_c[wave] = best_c; }
        __syncthreads();
        best_w = sh_w[0]; best_c = sh_c[0];
        for (int w = 1; w < 4; ++w) offer(sh_w[w], sh_c[w]);
        const bool single = a.method == 3 && best_c >= 0 && best_c < k;
        const bool far_win = far_pairs && best_c >= k;
        uint64_t win = 0;
        int wq_i = 0, wq_j = 0;
        if (far_win) {
            osd_pair_of(best_c - k, a.order, wq_i, wq_j);
        } else if (best_c >= 0 && !single) {
            bool valid;
            win = a.method == 3 ? pair_mask(best_c - k, valid) : (uint64_t)(best_c + 1) & kmask;
        }
        const uint64_t *plw = single ? Tm + (int64_t)(best_c >> 6) * m : Tm;
        for (int j = tid; j < n; j += T) {
            const int ci = colinfo[j];
            bool bit;
            if (far_win) {
                if (ci >= 0) bit = (((Tm[(int64_t)(wq_i >> 6) * m + ci] >> (wq_i & 63)) ^ (Tm[(int64_t)(wq_j >> 6) * m + ci] >> (wq_j & 63)) ^ (uint64_t)sy[ci]) & 1ull) != 0;
                else bit = -1 - ci == wq_i || -1 - ci == wq_j;
            } else if (ci >= 0) {
                const uint64_t tw = KW > 0 ? plw[ci] : 0ull;
                bit = single ? ((((tw >> (best_c & 63)) & 1ull) != 0) != (sy[ci] != 0))
                             : (((__builtin_popcountll(tw & win) + (int)sy[ci]) & 1) != 0);
            } else {
                const int q = -1 - ci;
                bit = single ? q == (int)best_c : (q < 64 && ((win >> (q & 63)) & 1ull) != 0);
            }
            a.decoding[b * n + j] = bit ? 1 : 0;
        }
        __syncthreads();
    }
}
