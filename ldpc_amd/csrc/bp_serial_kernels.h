// bp_serial_kernels.h -- serial schedule (bp.hpp:451-545) and soft-syndrome serial min-sum (bp.hpp:547-660)
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// ---- serial schedule (bp.hpp:451-545) with a fixed bit order -----------------------------------------
// The serial schedule is sequential in the bits of ONE syndrome (every bit update reads messages the
// previous bits just wrote) but the syndromes of a batch stay independent, so the lane = syndrome tile
// layout carries over: one wavefront walks the bits of its 64-syndrome tile in schedule order.  Per bit and
// per incident check the message is the plain sequential product (min) over the row's other entries
// (bp.hpp:493-498 / 507-517), signed by pow(-1, syndrome byte) (bp.hpp:499), in the reference's order.
// Only one message array is needed: check->bit messages never outlive the bit update that computes them.
// It holds tanh(b2c / 2) for product-sum (evaluated once per write instead of once per read: same value),
// b2c for min-sum.  The random and LLR-sorted ("serial_relative") orders differ per syndrome and are not
// provided on the device.
struct SerialArgs {
    int32_t m, n, nnz, max_iter, fast;
    double ms_scaling_factor;
    int64_t batch;
    const int32_t *row_ptr, *col_idx, *col_ptr, *csc_edge, *csc_row, *order;  // order may be nullptr (0..n-1)
    const double *llr0;
    double *A;                    // [tiles][nnz][64]  tanh(b2c/2) | b2c
    double *C;                    // [tiles][nnz][64]  scratch for nodes heavier than the register bounds
    const uint64_t *par, *invalid;
    uint64_t *dec, *dcur;
    double *llr_t;
    int32_t *iters;
    uint8_t *conv;
    // bp_serial_level_kernel: the schedule cut into levels of mutually check-disjoint bits (see below)
    const int32_t *lvl_ptr;   // [n_levels + 1]
    const int32_t *lvl_bits;  // [n] bits in level-major order (schedule order inside a level)
    int32_t n_levels;
    // random serial schedule (bp.hpp:467-468): iteration it walks orders[min(it, n_orders) - 1][0 .. n) instead of `order`
    const int32_t *orders;
    int32_t n_orders, orders_first;  // (the table is a ring: its first row sits at orders_first)
    // the same table for bp_serial_level_kernel: row r level-major in orders_lvl[r][0 .. n), orders_lvl_ptr[r][0] = its number of levels,
    // orders_lvl_ptr[r][1 + l] = where level l starts (stride n + 2)
    const int32_t *orders_lvl, *orders_lvl_ptr;
    // bp_serial_stream_kernel (bp_serial_stream_kernel.h): one record per position of the level-major order; the table of initial edge
    // values that stands in for the initial messages (or nullptr: they are written out); > 0: the message array holds the state after
    // it_start iterations (lanes compacted out of the tiles of a first pass), iterations count on from there
    const int32_t *pos_tab;
    const double *edge0;
    const double *pos_e0;  // [positions][16] initial values of the other entries of every position (with edge0)
    int32_t it_start;
    int32_t resume;  // 1: the same tiles carry on after a pass that stopped at it_start -- lanes whose `conv` says so are done (their `iters` kept), dec / llr_t are the pass's
    unsigned long long *clk;  // shader-clock probe (clock_probe_*), or nullptr
    // bp_serial_stream_var_kernel (bp_serial_var_kernel.h): item records, where the stream of (level l, wavefront w) starts
    // ([l * wavefronts + w], one more entry at the end), 1 KiB units of a wavefront's LDS queue
    const int32_t *var_items, *var_wq;
    int32_t var_units;
    const double *var_init;  // [nnz][64] initial segments shared by all tiles (the first iteration fetches what nobody has written yet from here), or nullptr
};

// One bit update of the serial schedule (bp.hpp:485-535) for the 64 syndromes of a tile: for every incident check the
// message from the row's other entries as they are NOW, the posterior, the hard decision, and the bit's new bit->check
// messages.  Shared by the single-wavefront kernel and the level-parallel one.
template <int METHOD, int MATH, int DCS, int DRS>
__device__ __forceinline__ void serial_update_bit(const SerialArgs &a, int bit, const MsgBuf &At, const MsgBuf &Ct, const MsgBuf &Lt,
                                                  const uint64_t *par, uint64_t *dcur, int lane, int l8, double alpha,
                                                  const double *log_tab, bool want_llr, bool lane_live) {
    const int cs = sload(a.col_ptr + bit);
    const int d = sload(a.col_ptr + bit + 1) - cs;
    double llr = sload(a.llr0 + bit);  // bp.hpp:488
    // the (other) entries of one incident check row -> its check->bit message for this bit
    auto row_message = [&](int chk, int e, const double *vals, int rs, int rd) {
        const bool odd = (sload(par + chk) >> lane) & 1ull;  // pow(-1, syndrome byte) / syndrome parity
        if (METHOD == LDPC_HIP_PRODUCT_SUM) {
            double c = 1.0;
            if (vals) {
#pragma unroll
                for (int q = 0; q < DRS; ++q)
                    if (q < rd && rs + q != e) c *= vals[q];
            } else {
                for (int g = rs; g < rs + rd; ++g)
                    if (g != e) c *= At.ld(l8, g);
            }
            c = ps_message<MATH>(c, odd, log_tab);
            return c;
        } else {
            int sgn = odd ? 1 : 0;
            double temp = DBL_MAX;
            if (vals) {
#pragma unroll
                for (int q = 0; q < DRS; ++q)
                    if (q < rd && rs + q != e) {
                        const double ab = fabs(vals[q]);
                        if (ab < temp) temp = ab;
                        if (vals[q] <= 0) sgn ^= 1;
                    }
            } else {
                for (int g = rs; g < rs + rd; ++g)
                    if (g != e) {
                        const double bg = At.ld(l8, g);
                        const double ab = fabs(bg);
                        if (ab < temp) temp = ab;
                        if (bg <= 0) sgn ^= 1;
                    }
            }
            return (alpha * (sgn ? -1.0 : 1.0)) * temp;  // alpha * message_sign * temp (bp.hpp:519)
        }
    };
    if (a.fast) {
        int e[DCS], chk[DCS], rs[DCS], rd[DCS];
        double vals[DCS][DRS], c[DCS], pre[DCS];
#pragma unroll
        for (int k = 0; k < DCS; ++k)
            if (k < d) {
                e[k] = sload(a.csc_edge + cs + k);
                chk[k] = sload(a.csc_row + cs + k);
                rs[k] = sload(a.row_ptr + chk[k]);
                rd[k] = sload(a.row_ptr + chk[k] + 1) - rs[k];
#pragma unroll
                for (int q = 0; q < DRS; ++q)
                    if (q < rd[k] && rs[k] + q != e[k]) vals[k][q] = At.ld(l8, rs[k] + q);
            }
#pragma unroll
        for (int k = 0; k < DCS; ++k)
            if (k < d) {
                c[k] = row_message(chk[k], e[k], vals[k], rs[k], rd[k]);
                pre[k] = llr;  // bp.hpp:501 / 520
                llr += c[k];
                if (METHOD == LDPC_HIP_PRODUCT_SUM) LDPC_EDGE_FENCE();
            }
        double temp = 0.0;  // bp.hpp:530-534
#pragma unroll
        for (int k = DCS - 1; k >= 0; --k)
            if (k < d) {
                At.st(l8, e[k], edge_form<METHOD, MATH>(pre[k] + temp));
                temp += c[k];
            }
    } else {
        for (int p = cs; p < cs + d; ++p) {
            const int e = sload(a.csc_edge + p), chk = sload(a.csc_row + p);
            const int rs = sload(a.row_ptr + chk), rd = sload(a.row_ptr + chk + 1) - rs;
            const double c = row_message(chk, e, nullptr, rs, rd);
            Ct.st(l8, e, c);
            At.st(l8, e, llr);  // partial sum; rewritten below before any other bit reads it
            llr += c;
        }
        double temp = 0.0;
        for (int p = cs + d - 1; p >= cs; --p) {
            const int e = sload(a.csc_edge + p);
            At.st(l8, e, edge_form<METHOD, MATH>(At.ld(l8, e) + temp));
            temp += Ct.ld(l8, e);
        }
    }
    const uint64_t hard = __ballot(llr <= 0);  // bp.hpp:525-529
    if (lane == 0) dcur[bit] = hard;
    if (want_llr && lane_live) Lt.st(l8, bit, llr);
}

// DCS / DRS: register bounds of the fast path (column / row weight); tighter bounds leave more wavefronts per SIMD, and
// this kernel is one dependent chain per bit, so the wavefronts in flight are what hides its latency
template <int METHOD, int MATH, int DCS, int DRS>
__global__ void __launch_bounds__(64) bp_serial_kernel(const SerialArgs a) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    const uint64_t *par = a.par + tile * m;
    const MsgBuf At = make_msgbuf(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf Ct = make_msgbuf(a.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    uint64_t *dec = a.dec + tile * n;
    uint64_t *dcur = a.dcur + tile * n;
    const bool want_llr = a.llr_t != nullptr;
    const MsgBuf Lt = make_msgbuf(want_llr ? a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.A, want_llr ? (unsigned)n : 0u);
    const int l8 = lane * 8;
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = lane; q < 256; q += 64) log_tab[q] = ldpc_math::k_log_tab[q];
    __builtin_amdgcn_wave_barrier();

    const int64_t valid = a.batch - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    const uint64_t never = a.invalid[tile];
    int my_iter = 0;

    for (int e = 0; e < nnz; ++e) At.st(l8, e, edge_form<METHOD, MATH>(sload(a.llr0 + sload(a.col_idx + e))));

    for (int it = 1; it <= a.max_iter; ++it) {
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        const bool lane_live = !((done >> lane) & 1ull);
        const int32_t *order = a.orders ? a.orders + (size_t)(((it < a.n_orders ? it : a.n_orders) - 1 + a.orders_first) % a.n_orders) * (size_t)n : a.order;
        for (int t = 0; t < n; ++t) {
            const int bit = order ? sload(order + t) : t;
            serial_update_bit<METHOD, MATH, DCS, DRS>(a, bit, At, Ct, Lt, par, dcur, lane, l8, alpha, log_tab, want_llr, lane_live);
        }
        // candidate syndrome of this iteration's hard decision vs the syndrome bytes (bp.hpp:537-543)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t unsat = 0;
        for (int i = lane; i < m; i += 64) {
            uint64_t cand = 0;
            for (int g = a.row_ptr[i]; g < a.row_ptr[i + 1]; ++g) cand ^= dcur[a.col_idx[g]];
            unsat |= cand ^ par[i];
        }
        unsat = wave_or(unsat) | never;
        const uint64_t newly = uniform64(~unsat & ~done);
        if (newly) {
            if ((newly >> lane) & 1ull) my_iter = it;
            for (int j = lane; j < n; j += 64) dec[j] = (dec[j] & ~newly) | (dcur[j] & newly);
            done |= newly;
        }
        if (done == ~0ull) break;
    }
    if (done != ~0ull)
        for (int j = lane; j < n; j += 64) dec[j] = (dec[j] & done) | (dcur[j] & ~done);
    const int64_t b = tile * LDPC_WAVE + lane;
    if (b < a.batch) {
        const bool cv = ((done >> lane) & 1ull) != 0;
        if (a.iters) a.iters[b] = cv ? my_iter : a.max_iter;
        if (a.conv) a.conv[b] = cv ? 1 : 0;
    }
}

// ---- the same schedule, level-parallel -----------------------------------------------------------------
// Two bits that share no check touch disjoint messages, so their serial updates commute.  Give every bit the level
// 1 + max(level of the EARLIER bits of the schedule it shares a check with): bits of one level are pairwise
// check-disjoint, and running level after level -- any order inside a level -- is indistinguishable from the serial
// order, because every bit still sees all and only the updates of the conflicting bits that precede it.  The (3,6)
// n = 10 000 code has 35 levels of ~286 bits, BB [[144,12,12]] 27 of ~5, the d = 21 surface code 61 of ~7.  So a
// WORKGROUP owns the tile and its wavefronts share each level's bits, with one workgroup barrier per level: the
// dependent chain per iteration shrinks from n bit updates to n_levels, and a tile keeps several wavefronts busy.
template <int METHOD, int MATH, int DCS, int DRS>
__global__ void __launch_bounds__(1024) bp_serial_level_kernel(const SerialArgs a) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    const uint64_t *par = a.par + tile * m;
    const MsgBuf At = make_msgbuf(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf Ct = make_msgbuf(a.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    uint64_t *dec = a.dec + tile * n;
    uint64_t *dcur = a.dcur + tile * n;
    const bool want_llr = a.llr_t != nullptr;
    const MsgBuf Lt = make_msgbuf(want_llr ? a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.A, want_llr ? (unsigned)n : 0u);
    const int l8 = lane * 8;
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    __shared__ uint64_t red[2][16];
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = threadIdx.x; q < 256; q += blockDim.x) log_tab[q] = ldpc_math::k_log_tab[q];

    const int64_t valid = a.batch - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    const uint64_t never = a.invalid[tile];
    int my_iter = 0;
    for (int e = wave; e < nnz; e += nwaves) At.st(l8, e, edge_form<METHOD, MATH>(sload(a.llr0 + sload(a.col_idx + e))));
    __syncthreads();

    for (int it = 1; it <= a.max_iter; ++it) {
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        const bool lane_live = !((done >> lane) & 1ull);
        // the levels of this iteration's order: the schedule's own, or (random schedule) row min(it, n_orders) - 1 of the ring
        const int32_t *lvl_ptr = a.lvl_ptr, *lvl_bits = a.lvl_bits;
        int n_levels = a.n_levels;
        if (a.orders_lvl) {
            const size_t row = (size_t)(((it < a.n_orders ? it : a.n_orders) - 1 + a.orders_first) % a.n_orders);
            lvl_ptr = a.orders_lvl_ptr + row * (size_t)(n + 2) + 1;
            lvl_bits = a.orders_lvl + row * (size_t)n;
            n_levels = sload(a.orders_lvl_ptr + row * (size_t)(n + 2));
        }
        for (int l = 0; l < n_levels; ++l) {
            const int p1 = sload(lvl_ptr + l + 1);
            for (int p = sload(lvl_ptr + l) + wave; p < p1; p += nwaves)
                serial_update_bit<METHOD, MATH, DCS, DRS>(a, sload(lvl_bits + p), At, Ct, Lt, par, dcur, lane, l8, alpha, log_tab,
                                                          want_llr, lane_live);
            __syncthreads();  // the next level reads what this one wrote
        }
        // candidate syndrome of this iteration's hard decision vs the syndrome bytes (bp.hpp:537-543)
        uint64_t unsat = 0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            uint64_t cand = 0;
            for (int g = a.row_ptr[i]; g < a.row_ptr[i + 1]; ++g) cand ^= dcur[a.col_idx[g]];
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
}

// ---- soft-syndrome serial min-sum: BpDecoder::soft_info_decode_serial (bp.hpp:547-660) ---------------------
// One wavefront per 64-shot tile, lane = shot.  The scaled analog syndrome S[tile][check][lane] and the hard
// syndrome (one ballot word per check, in LDS) are part of the decoder state: a check whose |S| is below the
// cutoff and below the smallest incoming magnitude behaves as a virtual variable node (bp.hpp:597-621).
struct SoftArgs {
    int32_t m, n, nnz, max_iter;
    double ms_scaling_factor, cutoff;
    int64_t batch;
    const int32_t *row_ptr, *col_idx, *col_ptr, *csc_edge, *csc_row, *order;  // order may be nullptr (0..n-1)
    const int32_t *orders;  // random serial schedule (bp.hpp:573-577): [n_orders][n], iteration it walks orders[min(it, n_orders) - 1]; else nullptr
    int32_t n_orders, orders_first;  // (a ring: its first row sits at orders_first)
    const int32_t *orders_lvl, *orders_lvl_ptr;  // the same rows level-major with their level bounds, as in SerialArgs (bp_softinfo_level_kernel)
    const double *llr0;
    double *A;        // [tiles][nnz][64] bit->check messages
    double *C;        // [tiles][nnz][64] check->bit messages of the bit being updated
    double *S;        // [tiles][m][64]   in: 2 s / sigma^2, out: the soft syndrome after decoding
    const uint64_t *syn;  // [tiles][m]   hard syndrome (S <= 0) at the start
    uint64_t *dec, *dcur;
    double *llr_t;
    int32_t *iters;
    uint8_t *conv;
    const int32_t *lvl_ptr, *lvl_bits;  // bp_softinfo_level_kernel: levels of check-disjoint bits (as for the serial schedule)
    int32_t n_levels;
};

// soft_info_decode_serial's preamble (bp.hpp:551-559): scale, take the sign, lay out lane-minor
__global__ void __launch_bounds__(256) softinfo_prepare_kernel(const double *__restrict__ soft, int64_t batch, int m, double sigma,
                                                               double *__restrict__ S, uint64_t *__restrict__ syn) {
    const int64_t tile = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= m) return;
    const int64_t b = tile * LDPC_WAVE + lane;
    double v = 1.0;
    if (b < batch) v = 2 * soft[b * m + i] / (sigma * sigma);
    S[((size_t)tile * m + i) * LDPC_WAVE + lane] = v;
    const uint64_t ones = __ballot(v <= 0);
    if (lane == 0) syn[tile * m + i] = ones;
}

// One bit update of soft_info_decode_serial (bp.hpp:580-639) for the 64 shots of a tile; `syn` = the tile's current hard
// syndrome words in LDS.  Shared by the single-wavefront kernel and the level-parallel one.
// The same bit update with everything the bit needs loaded up front (its <= DCS incident rows of <= DRS entries and their
// soft-syndrome values are distinct rows, so the loads are independent of what the update writes): one round of memory
// latency per bit instead of one per entry.  Used when the matrix respects the bounds.
template <int DCS, int DRS, class SynPtr>
__device__ __forceinline__ void soft_update_bit_fast(const SoftArgs &a, int bit, const MsgBuf &At, const MsgBuf &St, const MsgBuf &Lt,
                                                     SynPtr syn, uint64_t *dcur, int lane, int l8, bool want_llr, bool lane_live) {
    const int cs = sload(a.col_ptr + bit);
    const int d = sload(a.col_ptr + bit + 1) - cs;
    int e[DCS], chk[DCS], rs[DCS], rd[DCS];
    double vals[DCS][DRS], softv[DCS], c[DCS], pre[DCS];
#pragma unroll
    for (int k = 0; k < DCS; ++k)
        if (k < d) {
            e[k] = sload(a.csc_edge + cs + k);
            chk[k] = sload(a.csc_row + cs + k);
            rs[k] = sload(a.row_ptr + chk[k]);
            rd[k] = sload(a.row_ptr + chk[k] + 1) - rs[k];
#pragma unroll
            for (int q = 0; q < DRS; ++q)
                if (q < rd[k]) vals[k][q] = At.ld(l8, rs[k] + q);
            softv[k] = St.ld(l8, chk[k]);
        }
    double llr = sload(a.llr0 + bit);  // bp.hpp:583-584
#pragma unroll
    for (int k = 0; k < DCS; ++k)
        if (k < d) {
            int sgn = 0;
            double temp = DBL_MAX, own = 0.0;
#pragma unroll
            for (int q = 0; q < DRS; ++q)
                if (q < rd[k]) {
                    const double bg = vals[k][q];
                    if (rs[k] + q == e[k]) {
                        own = bg;
                    } else {  // bp.hpp:590-599
                        if (fabs(bg) < temp) temp = fabs(bg);
                        if (bg <= 0) sgn ^= 1;
                    }
                }
            const double min_msg = temp;
            double propagated = min_msg;
            double soft = softv[k];
            const double magnitude = fabs(soft);
            uint64_t word = syn[chk[k]];
            int hard = (int)((word >> lane) & 1ull);
            bool flip = false;
            if (magnitude < a.cutoff && magnitude < fabs(min_msg)) {  // bp.hpp:604-621
                propagated = magnitude;
                const int check_node_sgn = sgn ^ (own <= 0 ? 1 : 0);
                if (check_node_sgn == hard) {
                    const double mag = fabs(own) < min_msg ? fabs(own) : min_msg;
                    soft = hard ? -mag : mag;
                } else {
                    flip = true;
                    soft = -soft;
                }
                if (lane_live) St.st(l8, chk[k], soft);
            }
            const uint64_t flips = __ballot(flip);
            if (flips) {  // wave-uniform
                word ^= flips;
                if (lane == 0) syn[chk[k]] = word;
                hard = (int)((word >> lane) & 1ull);
                __builtin_amdgcn_wave_barrier();
            }
            sgn ^= hard;
            c[k] = (a.ms_scaling_factor * (sgn ? -1.0 : 1.0)) * propagated;  // bp.hpp:624
            pre[k] = llr;
            llr += c[k];
        }
    double back = 0.0;  // bp.hpp:634-638
#pragma unroll
    for (int k = DCS - 1; k >= 0; --k)
        if (k < d) {
            At.st(l8, e[k], pre[k] + back);
            back += c[k];
        }
    const uint64_t hard_bits = __ballot(llr <= 0);  // bp.hpp:628-633
    if (lane == 0) dcur[bit] = hard_bits;
    if (want_llr && lane_live) Lt.st(l8, bit, llr);
}

template <class SynPtr>
__device__ __forceinline__ void soft_update_bit(const SoftArgs &a, int bit, const MsgBuf &At, const MsgBuf &Ct, const MsgBuf &St,
                                                const MsgBuf &Lt, SynPtr syn, uint64_t *dcur, int lane, int l8, bool want_llr,
                                                bool lane_live) {
    const int cs = sload(a.col_ptr + bit);
    const int d = sload(a.col_ptr + bit + 1) - cs;
    double llr = sload(a.llr0 + bit);  // bp.hpp:583-584
    for (int p = cs; p < cs + d; ++p) {
        const int e = sload(a.csc_edge + p), chk = sload(a.csc_row + p);
        const int rs = sload(a.row_ptr + chk), re = sload(a.row_ptr + chk + 1);
        int sgn = 0;
        double temp = DBL_MAX;
        for (int g = rs; g < re; ++g)
            if (g != e) {  // bp.hpp:590-599
                const double bg = At.ld(l8, g);
                if (fabs(bg) < temp) temp = fabs(bg);
                if (bg <= 0) sgn ^= 1;
            }
        const double own = At.ld(l8, e);
        const double min_msg = temp;
        double propagated = min_msg;
        double soft = St.ld(l8, chk);
        const double magnitude = fabs(soft);
        uint64_t word = syn[chk];
        int hard = (int)((word >> lane) & 1ull);
        bool flip = false;
        if (magnitude < a.cutoff && magnitude < fabs(min_msg)) {  // bp.hpp:604-621
            propagated = magnitude;
            const int check_node_sgn = sgn ^ (own <= 0 ? 1 : 0);
            if (check_node_sgn == hard) {
                const double mag = fabs(own) < min_msg ? fabs(own) : min_msg;
                soft = hard ? -mag : mag;  // pow(-1, syndrome) * magnitude
            } else {
                flip = true;
                soft = -soft;
            }
            if (lane_live) St.st(l8, chk, soft);
        }
        const uint64_t flips = __ballot(flip);
        if (flips) {  // wave-uniform
            word ^= flips;
            if (lane == 0) syn[chk] = word;
            hard = (int)((word >> lane) & 1ull);
            __builtin_amdgcn_wave_barrier();
        }
        sgn ^= hard;
        const double c = (a.ms_scaling_factor * (sgn ? -1.0 : 1.0)) * propagated;  // bp.hpp:624
        Ct.st(l8, e, c);
        At.st(l8, e, llr);  // partial sum; completed by the reverse sweep below
        llr += c;
    }
    double back = 0.0;  // bp.hpp:634-638
    for (int p = cs + d - 1; p >= cs; --p) {
        const int e = sload(a.csc_edge + p);
        At.st(l8, e, At.ld(l8, e) + back);
        back += Ct.ld(l8, e);
    }
    const uint64_t hard_bits = __ballot(llr <= 0);  // bp.hpp:628-633
    if (lane == 0) dcur[bit] = hard_bits;
    if (want_llr && lane_live) Lt.st(l8, bit, llr);
}

// DCS / DRS > 0: the register path above (the matrix respects the bounds); DCS == 0: run-time loops for any degrees
template <int DCS, int DRS>
__global__ void __launch_bounds__(64) bp_softinfo_kernel(const SoftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char soft_lds[];
    volatile uint64_t *syn = reinterpret_cast<volatile uint64_t *>(soft_lds);  // [m] current hard syndrome
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    const MsgBuf At = make_msgbuf(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf Ct = make_msgbuf(a.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf St = make_msgbuf(a.S + (size_t)tile * (size_t)m * LDPC_WAVE, (unsigned)m);
    uint64_t *dec = a.dec + tile * n;
    uint64_t *dcur = a.dcur + tile * n;
    const bool want_llr = a.llr_t != nullptr;
    const MsgBuf Lt = make_msgbuf(want_llr ? a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.A, want_llr ? (unsigned)n : 0u);
    const int l8 = lane * 8;
    for (int i = lane; i < m; i += 64) syn[i] = a.syn[tile * m + i];
    __builtin_amdgcn_wave_barrier();

    const int64_t valid = a.batch - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    int my_iter = 0;
    for (int e = 0; e < nnz; ++e) At.st(l8, e, sload(a.llr0 + sload(a.col_idx + e)));  // bp.hpp:147-157

    for (int it = 1; it <= a.max_iter; ++it) {
        const bool lane_live = !((done >> lane) & 1ull);  // a converged shot keeps its outputs (bp.hpp:570-572)
        const int32_t *order = a.orders ? a.orders + (size_t)(((it < a.n_orders ? it : a.n_orders) - 1 + a.orders_first) % a.n_orders) * (size_t)n : a.order;
        for (int t = 0; t < n; ++t) {
            const int bit = order ? sload(order + t) : t;
            if constexpr (DCS > 0) soft_update_bit_fast<DCS, DRS>(a, bit, At, St, Lt, syn, dcur, lane, l8, want_llr, lane_live);
            else soft_update_bit(a, bit, At, Ct, St, Lt, syn, dcur, lane, l8, want_llr, lane_live);
        }
        // H x against the CURRENT hard syndrome (bp.hpp:640-655)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t unsat = 0;
        for (int i = lane; i < m; i += 64) {
            uint64_t cand = 0;
            for (int g = a.row_ptr[i]; g < a.row_ptr[i + 1]; ++g) cand ^= dcur[a.col_idx[g]];
            unsat |= cand ^ syn[i];
        }
        unsat = wave_or(unsat);
        const uint64_t newly = uniform64(~unsat & ~done);
        if (newly) {
            if ((newly >> lane) & 1ull) my_iter = it;
            for (int j = lane; j < n; j += 64) dec[j] = (dec[j] & ~newly) | (dcur[j] & newly);
            done |= newly;
        }
        if (done == ~0ull) break;
    }
    if (done != ~0ull)
        for (int j = lane; j < n; j += 64) dec[j] = (dec[j] & done) | (dcur[j] & ~done);
    const int64_t b = tile * LDPC_WAVE + lane;
    if (b < a.batch) {
        const bool cv = ((done >> lane) & 1ull) != 0;
        if (a.iters) a.iters[b] = cv ? my_iter : a.max_iter;
        if (a.conv) a.conv[b] = cv ? 1 : 0;
    }
}

// The same, level-parallel (see bp_serial_level_kernel): bits of a level share no check, hence touch disjoint messages,
// disjoint soft-syndrome entries and disjoint hard-syndrome words -- a workgroup runs the tile level by level.
template <int DCS, int DRS>
__global__ void __launch_bounds__(1024) bp_softinfo_level_kernel(const SoftArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char soft_lds[];
    volatile uint64_t *syn = reinterpret_cast<volatile uint64_t *>(soft_lds);  // [m] current hard syndrome, then [2][16] reduction slots
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    volatile uint64_t *red = syn + m;
    const MsgBuf At = make_msgbuf(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf Ct = make_msgbuf(a.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf St = make_msgbuf(a.S + (size_t)tile * (size_t)m * LDPC_WAVE, (unsigned)m);
    uint64_t *dec = a.dec + tile * n;
    uint64_t *dcur = a.dcur + tile * n;
    const bool want_llr = a.llr_t != nullptr;
    const MsgBuf Lt = make_msgbuf(want_llr ? a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.A, want_llr ? (unsigned)n : 0u);
    const int l8 = lane * 8;
    for (int i = threadIdx.x; i < m; i += blockDim.x) syn[i] = a.syn[tile * m + i];
    const int64_t valid = a.batch - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    int my_iter = 0;
    for (int e = wave; e < nnz; e += nwaves) At.st(l8, e, sload(a.llr0 + sload(a.col_idx + e)));
    __syncthreads();

    for (int it = 1; it <= a.max_iter; ++it) {
        const bool lane_live = !((done >> lane) & 1ull);
        const int32_t *lvl_ptr = a.lvl_ptr, *lvl_bits = a.lvl_bits;
        int n_levels = a.n_levels;
        if (a.orders_lvl) {  // random schedule: the levels of this iteration's order
            const size_t row = (size_t)(((it < a.n_orders ? it : a.n_orders) - 1 + a.orders_first) % a.n_orders);
            lvl_ptr = a.orders_lvl_ptr + row * (size_t)(n + 2) + 1;
            lvl_bits = a.orders_lvl + row * (size_t)n;
            n_levels = sload(a.orders_lvl_ptr + row * (size_t)(n + 2));
        }
        for (int l = 0; l < n_levels; ++l) {
            const int p1 = sload(lvl_ptr + l + 1);
            for (int p = sload(lvl_ptr + l) + wave; p < p1; p += nwaves) {
                if constexpr (DCS > 0) soft_update_bit_fast<DCS, DRS>(a, sload(lvl_bits + p), At, St, Lt, syn, dcur, lane, l8, want_llr, lane_live);
                else soft_update_bit(a, sload(lvl_bits + p), At, Ct, St, Lt, syn, dcur, lane, l8, want_llr, lane_live);
            }
            __syncthreads();
        }
        uint64_t unsat = 0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            uint64_t cand = 0;
            for (int g = a.row_ptr[i]; g < a.row_ptr[i + 1]; ++g) cand ^= dcur[a.col_idx[g]];
            unsat |= cand ^ syn[i];
        }
        unsat = wave_or(unsat);
        volatile uint64_t *slot_red = red + (it & 1) * 16;
        if (lane == 0) slot_red[wave] = unsat;
        __syncthreads();
        unsat = 0;
        for (int w = 0; w < nwaves; ++w) unsat |= slot_red[w];
        const uint64_t newly = uniform64(~unsat & ~done);
        if (newly) {
            if ((newly >> lane) & 1ull) my_iter = it;
            for (int j = threadIdx.x; j < n; j += blockDim.x) dec[j] = (dec[j] & ~newly) | (dcur[j] & newly);
            done |= newly;
            __syncthreads();
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
}
