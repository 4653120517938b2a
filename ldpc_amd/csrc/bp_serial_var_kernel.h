// bp_serial_var_kernel.h -- the streamed serial schedule (bp.hpp:451-545) for ANY degree profile: rows of up to 16 entries, columns of up to 8
// Part of libldpc_hip.so (translation unit tu_serial.hip).
#pragma once

#include "bp_device_common.h"
#include "bp_serial_kernels.h"
#include "bp_serial_stream_kernel.h"

// bp_serial_stream_kernel (bp_serial_stream_kernel.h) is built around one record per POSITION of the schedule sized for a (6,3) code:
// 15 other entries, 3 own ones, one ring slot.  Here the unit is an ITEM = (position, one of the bit's checks): the other entries of ONE
// check row -- consecutive edges of the message array with the bit's own entry left out -- are what a check -> bit message needs
// (bp.hpp:492-498 / 505-519), whatever the row's weight, and a bit of whatever weight is its items one after the other plus a closing
// step (bp.hpp:500-501, 525-534).  The rest is that kernel's arrangement:
//
//   * levels of mutually check-disjoint positions, a workgroup barrier per level, a wavefront taking the level's positions w, w + W, ...;
//   * the host lays the items out as one linear STREAM per (level, wavefront): a wavefront reads its records in sequence through the
//     scalar cache, two items ahead of their use, the syndrome word of the item's check and the bit's prior with them;
//   * a wavefront's LDS is a circular queue of 1 KiB units (one `buffer_load_dwordx4 ... lds` = two arbitrary 512-byte segments: lanes
//     0-31 fetch one, lanes 32-63 the other); an item of a row of d entries takes d / 2 (rounded down) units = ceil((d - 1) / 2), up to
//     two items are queued behind the current one as far as they fit, and the wait before an item is read is the counted one with the
//     count kept at run time (bp_stream_kernel.h, LDPC_RING_VAR: `ops` counts the vector-memory instructions issued -- DMAs and message
//     / decision stores; the posterior stores are left out: a count that is too small only waits longer);
//   * per lane and iteration the traffic is what the schedule itself needs: every entry is read once by each OTHER bit of its row and
//     written once.
//
// Same operations on the same operands in the same order as bp_serial_kernel's walk: the same bits.  Orders that are no permutation work
// as there: levels and items belong to positions.
//   * The first iteration needs no initial messages in the tile's array: an entry no earlier position of the schedule has written still
//     holds its initial value tanh(llr0 / 2) | llr0, the same in all 64 lanes and in every tile -- the record carries a mask of the
//     entries already written ([6]), and the others are fetched from ONE array of initial segments shared by all tiles
//     (SerialArgs::var_init, [nnz][64]: 20 MB on the n = 10 000 codes, served by L2 / the Infinity Cache, not by HBM): in that
//     iteration the DMAs are `global_load_lds_dwordx4` with a 64-bit address per lane (either array), the array is neither written
//     beforehand (1 / sum-d^2-th of an iteration per tile) nor read from HBM for what nobody has written (about half of the first
//     iteration's reads).  Only when the order visits every bit (else the messages are written out as before).
//
// Item record, int32[8], 32-byte aligned:
//   [0] the bit's own edge in this row        [1] first edge of the row        [2] d | k_own << 8 | k << 16 | dj << 24
//         (d = entries of the row, k_own = which of them is the bit's, k = which of the bit's dj checks this is, rows ascending)
//   [3] the bit                               [4] the check                    [5] 0 (the lane kernel's list: 1 = a real item)
//   [6] mask: bit t set = the row's t-th OTHER entry has been written by an earlier position of the schedule (first iteration)       [7] 0
constexpr int SERIAL_VAR_REC = 8;
typedef int ldpc_v8i_rec __attribute__((ext_vector_type(8)));

// one DMA instruction with a 64-bit address per lane: 16 bytes per lane to LDS at lds_addr + 16 * lane (lds_dma16's twin for two arrays)
__device__ __forceinline__ void lds_dma16_flat(unsigned long long vaddr, unsigned lds_addr) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(vaddr), "s"(lds_addr) : "memory");
}
// [nnz][64] what an edge holds before the first iteration, in all 64 lanes: tanh(llr0[column] / 2) | llr0[column]
template <int METHOD, int MATH>
__global__ void __launch_bounds__(256) serial_var_init_kernel(const double *__restrict__ llr0, const int32_t *__restrict__ col_idx, int nnz, double *__restrict__ out) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e < nnz) out[(size_t)e * 64 + lane] = edge_form<METHOD, MATH>(llr0[col_idx[e]]);
}

template <int METHOD, int MATH, int DRMAX, int DCMAX>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) bp_serial_stream_var_kernel(const SerialArgs a) {
    static_assert(DRMAX <= 16 && DRMAX % 2 == 0 && DCMAX <= 8, "an item is at most 8 units; a bit has at most 8 checks");
    constexpr int NDMAX = DRMAX / 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    const uint64_t *par = a.par + tile * m;
    const MsgBufNT At = make_msgbuf<MsgBufNT>(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    uint64_t *dec = a.dec + tile * n;
    uint64_t *dcur = a.dcur + tile * n;
    const bool want_llr = a.llr_t != nullptr;
    const MsgBufNT Lt = make_msgbuf<MsgBufNT>(want_llr ? a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.A, want_llr ? (unsigned)n : 0u);
    const int l8 = lane * 8;
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    __shared__ uint64_t red[2][16];
    __shared__ unsigned long long clk_stamp[2];
    if (threadIdx.x == 0) clock_probe_begin(clk_stamp);
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = threadIdx.x; q < 256; q += blockDim.x) log_tab[q] = ldpc_math::k_log_tab[q];
    const int U = a.var_units;  // 1 KiB units of this wavefront's queue
    const unsigned ring_addr = (unsigned)(uintptr_t)ldpc_dyn_lds + (unsigned)wave * (unsigned)U * 1024u;
    const double *ringp = reinterpret_cast<const double *>(ldpc_dyn_lds + (size_t)wave * (size_t)U * 1024u);
    const unsigned l16 = (unsigned)(lane & 31) * 16u;
    const bool upper = lane >= 32;
    const unsigned beyond = (unsigned)nnz << 9;  // an offset the buffer's range check rejects: zeros, no memory access

    const int64_t valid = a.batch - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    const uint64_t never = a.invalid[tile];
    int my_iter = 0;
    if (a.resume) {  // the lanes an earlier pass over these tiles finished
        const int64_t b = tile * LDPC_WAVE + lane;
        const bool was = b < a.batch && a.conv[b] != 0;
        if (was) my_iter = a.iters[b];
        done |= __ballot(was);
    }
    const bool implicit_init = a.var_init != nullptr && a.it_start == 0;
    if (a.it_start == 0 && !implicit_init)  // initialise_log_domain_bp (bp.hpp:147-157)
        for (int e = wave; e < nnz; e += nwaves) At.st(l8, e, edge_form<METHOD, MATH>(sload(a.llr0 + sload(a.col_idx + e))));
    __syncthreads();
    const ldpc_v8i_rec *items = reinterpret_cast<const ldpc_v8i_rec *>(a.var_items);
    const unsigned long long tile_base = (unsigned long long)(uintptr_t)(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE);
    const unsigned long long init_base = (unsigned long long)(uintptr_t)a.var_init;

    for (int it = a.it_start + 1; it <= a.max_iter; ++it) {
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        const bool lane_live = !((done >> lane) & 1ull);
        // One level: this wavefront's stream of items.  FRESH = the first iteration of an implicitly initialised decode.
        auto run_level = [&](auto fresh_tag, const int l) {
            constexpr bool FRESH = decltype(fresh_tag)::value;
            const int q0 = sload(a.var_wq + l * nwaves + wave), q1 = sload(a.var_wq + l * nwaves + wave + 1);
            const int nitems = q1 - q0;
            int head = 0;
            unsigned ops = 0;
            // what travels with a queued item: its record, where its units start, the count of operations after its last DMA, the
            // syndrome word of its check, the prior of its bit
            struct Slot {
                ldpc_v8i_rec r;
                int pos;
                unsigned mark;
                uint64_t parw;
                double prior;
            };
            auto issue = [&](Slot &s) {
                const int d = s.r[2] & 255, kown = (s.r[2] >> 8) & 255, rs = s.r[1];
                const int units = d >> 1;  // ceil((d - 1) / 2)
                s.pos = head;
                const unsigned written = FRESH ? (unsigned)s.r[6] : ~0u;
#pragma unroll
                for (int c = 0; c < NDMAX; ++c)
                    if (c < units) {
                        const int t0 = 2 * c, t1 = 2 * c + 1;
                        int u = head + c;
                        if (u >= U) u -= U;
                        if (FRESH) {  // either array, a 64-bit address per lane (an odd last half fetches the first initial segment: valid memory, never read)
                            const unsigned long long a0 = (((written >> t0) & 1u) ? tile_base : init_base) + ((unsigned long long)(unsigned)(rs + t0 + (t0 >= kown ? 1 : 0)) << 9);
                            const unsigned long long a1 = t1 < d - 1 ? (((written >> t1) & 1u) ? tile_base : init_base) + ((unsigned long long)(unsigned)(rs + t1 + (t1 >= kown ? 1 : 0)) << 9)
                                                                     : init_base;
                            lds_dma16_flat((upper ? a1 : a0) + l16, ring_addr + (unsigned)u * 1024u);
                        } else {
                            const unsigned ea = (unsigned)(rs + t0 + (t0 >= kown ? 1 : 0)) << 9;
                            const unsigned eb = t1 < d - 1 ? (unsigned)(rs + t1 + (t1 >= kown ? 1 : 0)) << 9 : beyond;
                            lds_dma16(At.rsrc, (upper ? eb : ea) + l16, 0u, ring_addr + (unsigned)u * 1024u);
                        }
                    }
                head += units;
                if (head >= U) head -= U;
                ops += (unsigned)units;
                s.mark = ops;
                s.parw = sload(par + s.r[4]);
                s.prior = sload(a.llr0 + s.r[3]);
            };
            auto units_of = [](const ldpc_v8i_rec &r) { return (r[2] & 255) >> 1; };
            Slot c0 = {}, c1 = {}, c2 = {};
            ldpc_v8i_rec nx = {0, 0, 0, 0, 0, 0, 0, 0};
            int ahead = 0, queued = 0;
            if (nitems > 0) {
                c0.r = sload(items + q0);
                issue(c0);
                queued = 1;
                if (nitems > 1) nx = sload(items + q0 + 1);
            }
            double cs[DCMAX];   // the messages of the current position's checks so far (bp.hpp:492-521)
            int own[DCMAX];     // ... and the bit's own edges
#pragma unroll
            for (int k = 0; k < DCMAX; ++k) { cs[k] = 0.0; own[k] = 0; }
            for (int idx = 0; idx < nitems; ++idx) {
                wait_vmcnt_dyn((int)(ops - c0.mark));
                const int d = c0.r[2] & 255, k = (c0.r[2] >> 16) & 255, dj = (c0.r[2] >> 24) & 255;
                double v[DRMAX - 1];
#pragma unroll
                for (int t = 0; t < DRMAX - 1; ++t)
                    if (t < d - 1) {
                        int u = c0.pos + (t >> 1);
                        if (u >= U) u -= U;
                        v[t] = ringp[u * 128 + (t & 1) * LDPC_WAVE + lane];
                    }
                wait_lds_reads();  // the item's units are free (and the scalar loads asked for a step ago have landed)
                if (ahead == 0 && queued < nitems) {
                    c1.r = nx;
                    issue(c1);
                    ahead = 1;
                    if (++queued < nitems) nx = sload(items + q0 + queued);
                }
                if (ahead == 1 && queued < nitems && units_of(c1.r) + units_of(nx) <= U) {
                    c2.r = nx;
                    issue(c2);
                    ahead = 2;
                    if (++queued < nitems) nx = sload(items + q0 + queued);
                }
                const bool odd = (c0.parw >> lane) & 1ull;  // pow(-1, syndrome byte) / syndrome parity
                double c;
                if (METHOD == LDPC_HIP_PRODUCT_SUM) {
                    double x = 1.0;  // bp.hpp:492-498: the product over the row's other entries, in the row's order
#pragma unroll
                    for (int t = 0; t < DRMAX - 1; ++t)
                        if (t < d - 1) x *= v[t];
                    c = ps_message<MATH>(x, odd, log_tab);
                } else {
                    int sgn = odd ? 1 : 0;  // bp.hpp:505-519
                    double temp = DBL_MAX;
#pragma unroll
                    for (int t = 0; t < DRMAX - 1; ++t)
                        if (t < d - 1) {
                            const double ab = fabs(v[t]);
                            if (ab < temp) temp = ab;
                            if (v[t] <= 0) sgn ^= 1;
                        }
                    c = (alpha * (sgn ? -1.0 : 1.0)) * temp;
                }
#pragma unroll
                for (int kk = 0; kk < DCMAX; ++kk)
                    if (kk == k) { cs[kk] = c; own[kk] = c0.r[0]; }  // (k is wave-uniform)
                if (k == dj - 1) {
                    // the bit's closing step: posterior and hard decision (bp.hpp:488, 500-501 / 520-521, 525-529), then its new bit -> check
                    // messages, last check first (bp.hpp:530-534).  The running sums are formed again from the prior: the same additions
                    // in the same order as message by message.
                    const int bit = c0.r[3];
                    double pre[DCMAX];
                    double llr = c0.prior;
#pragma unroll
                    for (int kk = 0; kk < DCMAX; ++kk)
                        if (kk < dj) { pre[kk] = llr; llr += cs[kk]; }
                    double temp = 0.0;
#pragma unroll
                    for (int kk = DCMAX - 1; kk >= 0; --kk)
                        if (kk < dj) {
                            At.st(l8, own[kk], edge_form<METHOD, MATH>(pre[kk] + temp));
                            temp += cs[kk];
                            if (METHOD == LDPC_HIP_PRODUCT_SUM) LDPC_EDGE_FENCE();
                        }
                    const uint64_t hard = __ballot(llr <= 0);
                    if (lane == 0) dcur[bit] = hard;
                    if (want_llr && lane_live) Lt.st(l8, bit, llr);
                    ops += (unsigned)dj + 1u;  // the message stores and the decision word (the posterior store is not counted: see the top)
                }
                c0 = c1;
                c1 = c2;
                if (ahead > 0) --ahead;
            }
            wait_vmcnt<0>();
        };
        const bool fresh = implicit_init && it == 1;
        for (int l = 0; l < a.n_levels; ++l) {
            if (fresh) run_level(std::true_type{}, l);
            else run_level(std::false_type{}, l);
            __syncthreads();  // the next level reads what this one wrote
        }
        // candidate syndrome of this iteration's hard decision vs the syndrome bytes (bp.hpp:537-543)
        uint64_t unsat = 0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            uint64_t cand = 0;
            for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) cand ^= dcur[a.col_idx[e]];
            unsat |= cand ^ par[i];
        }
        unsat = wave_or(unsat);
        uint64_t *slot_red = red[it & 1];  // double-buffered: no barrier needed before the next reuse
        if (lane == 0) slot_red[wave] = unsat;
        __syncthreads();
        unsat = never;
        for (int w = 0; w < nwaves; ++w) unsat |= slot_red[w];
        const uint64_t newly = uniform64(~unsat & ~done);
        if (newly) {
            if ((newly >> lane) & 1ull) my_iter = it;
            for (int j = threadIdx.x; j < n; j += blockDim.x) dec[j] = (dec[j] & ~newly) | (dcur[j] & newly);
            done |= newly;
            __syncthreads();  // (newly is workgroup-uniform) the next iteration overwrites dcur
        }
        if (done == ~0ull) break;
    }
    __syncthreads();
    if (done != ~0ull)
        for (int j = threadIdx.x; j < n; j += blockDim.x) dec[j] = (dec[j] & done) | (dcur[j] & ~done);
    if (wave == 0) {
        const int64_t b = tile * LDPC_WAVE + lane;
        if (b < a.batch) {
            const bool cv = ((done >> lane) & 1ull) != 0;
            if (a.iters) a.iters[b] = cv ? my_iter : a.max_iter;
            if (a.conv) a.conv[b] = cv ? 1 : 0;
        }
    }
    if (threadIdx.x == 0) clock_probe_end(a.clk, clk_stamp);
}

// ---- the same schedule for a HANDFUL of syndromes: one workgroup per syndrome, lane = item ------------------------------------------------
// bp_serial_lane_kernel's arrangement (bp_serial_stream_kernel.h) for any degree profile: a syndrome's messages as one row-major array
// [nnz] in L2, a level's items one per LANE.  The host pads the level-major item list so that the items of a position sit in ONE
// wavefront (lane_items: records as above, [5] = 1 for a real item, 0 for padding): each lane forms its check's message, the position's
// messages go round its lanes by lane permutation, every lane adds them up in the reference's order and writes its own entry's new
// message -- the dependent chain of a level is one `log` and one `tanh` long whatever the bit's weight.
struct SerialLaneVarArgs {
    int32_t m, n, nnz, max_iter, it_start, n_levels;
    double ms_scaling_factor;
    int64_t rows;
    const int32_t *row_ptr, *col_idx;
    const int32_t *lane_lvl;    // [n_levels + 1] where a level's (padded, multiple of 64) items start
    const int32_t *lane_items;  // records
    const double *llr0;
    double *A;            // [rows][nnz] tanh(b2c / 2) | b2c, row-major per syndrome
    const uint8_t *synd;  // [rows][m]
    uint8_t *decoding;    // [rows][n]
    double *llr;          // [rows][n] or nullptr
    int32_t *iters;
    uint8_t *conv;
};

template <int METHOD, int MATH, int DRMAX, int DCMAX>
__global__ void __launch_bounds__(1024) bp_serial_lane_var_kernel(const SerialLaneVarArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lane_lds[];
    uint8_t *dbit = lane_lds;  // [n] this iteration's hard decisions
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const int64_t r = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    double *A = a.A + r * (int64_t)nnz;
    const uint8_t *synd = a.synd + r * (int64_t)m;
    double *llr_out = a.llr ? a.llr + r * (int64_t)n : nullptr;
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = tid; q < 256; q += T) log_tab[q] = ldpc_math::k_log_tab[q];
    for (int j = tid; j < n; j += T) dbit[j] = 0;
    if (a.it_start == 0)
        for (int e = tid; e < nnz; e += T) A[e] = edge_form<METHOD, MATH>(a.llr0[a.col_idx[e]]);
    __syncthreads();
    const ldpc_v8i_rec *items = reinterpret_cast<const ldpc_v8i_rec *>(a.lane_items);
    bool converged = false;
    int it_done = a.max_iter;
    for (int it = a.it_start + 1; it <= a.max_iter; ++it) {
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        for (int l = 0; l < a.n_levels; ++l) {
            const int p0 = a.lane_lvl[l], p1 = a.lane_lvl[l + 1];
            for (int pb = p0 + wave * LDPC_WAVE; pb < p1; pb += nwaves * LDPC_WAVE) {  // (whole wavefronts: the permutations below need them)
                const ldpc_v8i_rec rec = items[pb + lane];
                const bool on = rec[5] != 0;
                const int own = rec[0], rs = rec[1], d = rec[2] & 255, kown = (rec[2] >> 8) & 255, k = (rec[2] >> 16) & 255, dj = (rec[2] >> 24) & 255;
                const int bit = rec[3];
                double c = 0.0;
                if (on) {
                    const bool odd = synd[rec[4]] & 1;  // pow(-1, syndrome byte) / syndrome parity
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) {
                        double x = 1.0;  // bp.hpp:492-498
                        for (int t = 0; t < d - 1; ++t) x *= A[rs + t + (t >= kown ? 1 : 0)];
                        c = ps_message<MATH>(x, odd, log_tab);
                    } else {
                        int sgn = odd ? 1 : 0;  // bp.hpp:505-519
                        double temp = DBL_MAX;
                        for (int t = 0; t < d - 1; ++t) {
                            const double b = A[rs + t + (t >= kown ? 1 : 0)];
                            const double ab = fabs(b);
                            if (ab < temp) temp = ab;
                            if (b <= 0) sgn ^= 1;
                        }
                        c = (alpha * (sgn ? -1.0 : 1.0)) * temp;
                    }
                }
                const int base = lane - k;  // the position's first lane (padding: k = 0)
                double llr = on ? a.llr0[bit] : 0.0, pre = 0.0, temp = 0.0;  // bp.hpp:488, 500-501
                double csj[DCMAX];
#pragma unroll
                for (int j = 0; j < DCMAX; ++j) csj[j] = __shfl(c, (base + j) & 63, LDPC_WAVE);
#pragma unroll
                for (int j = 0; j < DCMAX; ++j)
                    if (j < dj) {
                        if (j == k) pre = llr;
                        llr += csj[j];
                    }
#pragma unroll
                for (int j = DCMAX - 1; j >= 0; --j)  // bp.hpp:530-534: what the entries after this one add
                    if (j < dj && j > k) temp += csj[j];
                if (on) {
                    A[own] = edge_form<METHOD, MATH>(pre + temp);
                    if (k == 0) {
                        dbit[bit] = llr <= 0 ? 1 : 0;  // bp.hpp:525-529
                        if (llr_out) llr_out[bit] = llr;
                    }
                }
            }
            __syncthreads();  // the next level reads what this one wrote
        }
        // candidate syndrome of this iteration's hard decision vs the syndrome BYTES (bp.hpp:537-543: a byte > 1 never matches)
        int bad = 0;
        for (int i = tid; i < m; i += T) {
            int cand = 0;
            for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) cand ^= dbit[a.col_idx[e]];
            bad |= cand != (int)synd[i];
        }
        if (!__syncthreads_or(bad)) { converged = true; it_done = it; break; }
    }
    uint8_t *dec = a.decoding + r * (int64_t)n;
    for (int j = tid; j < n; j += T) dec[j] = dbit[j];
    if (tid == 0) {
        if (a.iters) a.iters[r] = converged ? it_done : a.max_iter;
        if (a.conv) a.conv[r] = converged ? 1 : 0;
    }
}
