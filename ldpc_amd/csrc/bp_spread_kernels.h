// bp_spread_kernels.h -- per-pass kernels: check pass, bit pass, syndrome test and bookkeeping spread over the whole chip
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// ---- per-pass kernels for handed-off tiles ---------------------------------------------------------------
// Same arithmetic, same arrays, but one launch per pass and one wavefront per NODE, so the rows / columns of a
// single tile are spread over all 256 compute units.  Used for the tiles the persistent kernel parks when the
// chip would otherwise idle; each launch handles every parked tile that is still running.
struct SpreadArgs {
    BpArgs bp;
    int32_t n_tiles;  // entries of bp.handoff_list; < 0: only the device knows (bp.counters[1], left by the persistent kernel)
    int32_t nodes;    // rows / columns per wavefront (1 for a handful of tiles: latency; 4, or 16 for >= 512 tiles from the start: amortises the table load)
    int32_t round;    // 0-based per-pass round; a tile's iteration number is it0 + round + 1
    unsigned *host_flag;  // host-mapped word: receives `seq` when the last parked tile becomes final (the host stops queueing rounds)
    unsigned seq;
    int32_t slot0;    // the first slot of the list this launch serves (workgroup row y: slot0 + y; the LOOP forms: and every gridDim.y-th after it)
};

// Late rounds.  After a few rounds all but a handful of the tiles are final (what is left is what never converges), yet a launch of a row of
// workgroups per parked tile -- 256 rows that leave at once -- costs ~50 us, four times a round, for the 40 rounds one hopeless syndrome keeps
// its tile going.  Where the host expects that (host_stream.h) the list is compacted on the device every 8 rounds (bp_spread_compact_kernel:
// the running tiles to the front) and a round is then TWO launches per kernel: 32 rows for slots 0 .. 31, exactly as before, and 8 rows of
// the LOOP form for whatever lies beyond (slot0 = 32: every 8th slot each; nothing, normally -- they leave at once).  The row-per-slot form
// stays as it was: with a loop over slots in it the check kernel needed 109 VGPRs instead of 73 and the headline lost 3 %.
__device__ __forceinline__ int spread_count(const SpreadArgs &a) { return a.n_tiles >= 0 ? a.n_tiles : (int)a.bp.counters[1]; }

// the list without the tiles that are final (in place, one wavefront; counters[1] = how many are left)
__global__ void __launch_bounds__(64) bp_spread_compact_kernel(const SpreadArgs a) {
    const int lane = threadIdx.x;
    const int n_tiles = spread_count(a);
    int kept = 0;
    for (int s0 = 0; s0 < n_tiles; s0 += 64) {  // (a chunk is read whole before any of it is overwritten: kept <= s0)
        const int slot = s0 + lane;
        int32_t tile = 0;
        bool live = false;
        if (slot < n_tiles) {
            tile = a.bp.handoff_list[slot];
            live = a.round <= a.bp.state[tile].end_round;
        }
        const uint64_t mask = __ballot(live);
        __builtin_amdgcn_wave_barrier();
        if (live) a.bp.handoff_list[kept + lane_rank(mask)] = tile;
        kept += __builtin_popcountll(mask);
    }
    if (lane == 0) a.bp.counters[1] = (unsigned)kept;
}

// tile in `slot`, its iteration number and converged mask in this round; false: no such slot, or the tile is final
__device__ __forceinline__ bool spread_tile(const SpreadArgs &a, int slot, int64_t &tile, const TileState *&st, int &it, uint64_t &done) {
    // The host sizes the grid for the most tiles that can have been parked and queues every round without waiting for
    // the device (the *_async entry points never synchronise): rows beyond the parked count and tiles that are final leave here.
    if (slot >= spread_count(a)) return false;
    tile = a.bp.handoff_list[slot];
    st = a.bp.state + tile;
    it = st->it0 + a.round + 1;
    done = st->done[a.round & 1];
    return a.round <= st->end_round;
}


template <int METHOD, int MATH, int DR, int NT, bool LOOP = false>
__global__ void __launch_bounds__(256) bp_spread_check_kernel(const SpreadArgs a) {
    typedef MsgBufT<NT ? 2 : 0> Buf;  // cache policy of the message traffic: non-temporal once the tiles outgrow the caches
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    __shared__ double near_bufs[4][LDPC_NEAR_SLOTS];
    int64_t tile;
    const TileState *st;
    int it;
    uint64_t done;
    int slot = a.slot0 + (int)blockIdx.y;
    if (LOOP) { if (slot >= spread_count(a)) return; }
    else if (!spread_tile(a, slot, tile, st, it, done)) return;  // (uniform over the workgroup)
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = threadIdx.x; q < 256; q += blockDim.x) log_tab[q] = ldpc_math::k_log_tab[q];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nnz = a.bp.nnz, l8 = lane * 8;
    do {
    if (LOOP && !spread_tile(a, slot, tile, st, it, done)) continue;
    const Buf At = make_msgbuf<Buf>(a.bp.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const Buf Ct = make_msgbuf<Buf>(a.bp.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const double alpha = (a.bp.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.bp.ms_scaling_factor;
    const int i0 = (blockIdx.x * 4 + wave) * a.nodes;
    for (int i = i0; i < i0 + a.nodes && i < a.bp.m; ++i) {
        const int rs = sload(a.bp.row_ptr + i), d = sload(a.bp.row_ptr + i + 1) - rs;
        const bool neg = (sload(a.bp.nzm + tile * a.bp.m + i) >> lane) & 1ull;
        const int parity = (int)((sload(a.bp.par + tile * a.bp.m + i) >> lane) & 1ull);
        if (d <= DR) {
            double cur[DR];
#pragma unroll
            for (int k = 0; k < DR; ++k)
                if (k < d) cur[k] = At.ld(l8, rs + k);
            check_row_live<METHOD, MATH, DR>(cur, d, rs, neg, parity, alpha, Ct, l8, log_tab, ~done, near_bufs[wave]);
        } else {
            check_row_streamed<METHOD, MATH>(d, rs, neg, parity, alpha, At, Ct, l8, log_tab);
        }
    }
    } while (LOOP && (slot += (int)gridDim.y) < spread_count(a));
}

template <int METHOD, int MATH, int DC, int NT, bool LOOP = false>
__global__ void __launch_bounds__(256) bp_spread_bit_kernel(const SpreadArgs a) {
    typedef MsgBufT<NT ? 2 : 0> Buf;
    int64_t tile;
    const TileState *st;
    int it;
    uint64_t done;
    int slot = a.slot0 + (int)blockIdx.y;
    if (LOOP) { if (slot >= spread_count(a)) return; }
    else if (!spread_tile(a, slot, tile, st, it, done)) return;
    do {
    if (LOOP && !spread_tile(a, slot, tile, st, it, done)) continue;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nnz = a.bp.nnz, n = a.bp.n, l8 = lane * 8;
    const Buf At = make_msgbuf<Buf>(a.bp.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const Buf Ct = make_msgbuf<Buf>(a.bp.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const bool want_llr = a.bp.llr_t != nullptr;
    const Buf Lt = make_msgbuf<Buf>(want_llr ? a.bp.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.bp.A, want_llr ? (unsigned)n : 0u);
    const bool last = it == a.bp.max_iter;
    const bool lane_live = !((done >> lane) & 1ull);
    const bool each = st->llr_each[a.round & 1] != 0;  // (as the persistent kernel's llr_each)
    const int j0 = (blockIdx.x * 4 + wave) * a.nodes;
    for (int j = j0; j < j0 + a.nodes && j < n; ++j) {
        const int cs = sload(a.bp.col_ptr + j), d = sload(a.bp.col_ptr + j + 1) - cs;
        const double prior = sload(a.bp.llr0 + j);
        double llr;
        if (d <= DC) {
            int e[DC];
            double c[DC];
#pragma unroll
            for (int k = 0; k < DC; ++k)
                if (k < d) { e[k] = sload(a.bp.csc_edge + cs + k); c[k] = Ct.ld(l8, e[k]); }
            llr = bit_column<METHOD, MATH, DC>(c, e, d, prior, At, l8, !last || a.bp.keep_state != 0);
        } else {  // the reference's two sweeps (bp.hpp:278-281, 313-316) through memory
            double temp = prior;
            for (int k = 0; k < d; ++k) {
                const int ee = sload(a.bp.csc_edge + cs + k);
                At.st(l8, ee, temp);
                temp += Ct.ld(l8, ee);
            }
            llr = temp;
            double sfx = 0.0;
            for (int k = d - 1; k >= 0; --k) {
                const int ee = sload(a.bp.csc_edge + cs + k);
                At.st(l8, ee, edge_form<METHOD, MATH>(At.ld(l8, ee) + sfx));
                sfx += Ct.ld(l8, ee);
            }
        }
        const uint64_t hard = __ballot(llr <= 0);
        if (lane == 0) a.bp.dcur[tile * n + j] = hard;
        if ((last || each) && want_llr && lane_live) Lt.st(l8, j, llr);
    }
    } while (LOOP && (slot += (int)gridDim.y) < spread_count(a));
}

// candidate syndrome vs syndrome (bp.hpp:292-294, 300-302) for the parked tiles, one thread per (tile, row); the
// per-tile verdict is OR-accumulated into TileState::unsat for bp_spread_finish_kernel
template <bool LOOP = false>
__global__ void __launch_bounds__(256) bp_spread_synd_kernel(const SpreadArgs a) {
    int64_t tile;
    const TileState *st;
    int it;
    uint64_t done;
    int slot = a.slot0 + (int)blockIdx.y;
    if (LOOP) { if (slot >= spread_count(a)) return; }
    else if (!spread_tile(a, slot, tile, st, it, done)) return;
    do {
    if (LOOP && !spread_tile(a, slot, tile, st, it, done)) continue;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t unsat = 0;
    if (i < a.bp.m) {
        const uint64_t *dcur = a.bp.dcur + tile * a.bp.n;
        uint64_t cand = 0;
        for (int e = a.bp.row_ptr[i]; e < a.bp.row_ptr[i + 1]; ++e) cand ^= dcur[a.bp.col_idx[e]];
        unsat = cand ^ a.bp.par[tile * a.bp.m + i];
    }
    unsat = wave_or(unsat);
    if ((threadIdx.x & 63) == 0 && unsat) atomicOr(&a.bp.state[tile].unsat[a.round & 1], (unsigned long long)unsat);
    } while (LOOP && (slot += (int)gridDim.y) < spread_count(a));
}

// batches of only a few tiles skip the persistent kernel altogether: state + message initialisation for the per-pass path
__global__ void __launch_bounds__(256) bp_spread_state_init_kernel(const SpreadArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_tiles) return;
    TileState *st = a.bp.state + t;
    const int64_t valid = a.bp.batch - (int64_t)t * LDPC_WAVE;
    st->done[0] = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    st->unsat[0] = st->unsat[1] = 0ull;
    st->it0 = a.bp.it_start;
    st->end_round = INT32_MAX;
    // lanes compacted out of a first pass (it_start > 0) are the ones about to converge: store their posteriors in every bit pass from
    // the start instead of paying a sweep over C per convergence event (1.3 ms per round on the headline code at p = 0.05)
    st->llr_each[0] = a.bp.it_start > 0 ? 1 : 0;
    for (int l = 0; l < 64; ++l) st->lane_iter[l] = 0;
    a.bp.handoff_list[t] = t;
    if (t == 0) a.bp.counters[1] = a.bp.counters[2] = (unsigned)a.n_tiles;  // parked, live
}

template <int METHOD, int MATH>
__global__ void __launch_bounds__(256) bp_spread_init_kernel(const SpreadArgs a) {  // bp.hpp:147-157
    const int64_t tile = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nnz = a.bp.nnz, l8 = lane * 8;
    const MsgBuf At = make_msgbuf(a.bp.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const int e0 = (blockIdx.x * 4 + wave) * 16;
    for (int e = e0; e < e0 + 16 && e < nnz; ++e)
        At.st(l8, e, edge_form<METHOD, MATH>(sload(a.bp.llr0 + sload(a.bp.col_idx + e))));
}

// [n] initial edge values for BpArgs::edge0
template <int METHOD, int MATH>
__global__ void __launch_bounds__(256) bp_edge0_kernel(const double *llr0, int n, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = edge_form<METHOD, MATH>(llr0[j]);
}

// convergence bookkeeping of a round (bp.hpp:296-311, 320-322): lanes whose candidate syndrome matched are frozen
// (decisions + posterior of THIS iteration), a tile whose lanes are all frozen or that reached max_iter gets its
// outputs.  64 bits per workgroup; workgroup 0 of a tile also advances its state.  Almost always there is nothing
// to freeze and every workgroup but the first leaves at once.
template <bool LOOP = false>
__global__ void __launch_bounds__(256) bp_spread_finish_kernel(const SpreadArgs a) {
    int64_t tile;
    const TileState *cst;
    int it;
    uint64_t done;
    int slot = a.slot0 + (int)blockIdx.y;
    if (LOOP) { if (slot >= spread_count(a)) return; }
    else if (!spread_tile(a, slot, tile, cst, it, done)) return;
    do {
    if (LOOP && !spread_tile(a, slot, tile, cst, it, done)) continue;
    TileState *st = a.bp.state + tile;
    const int par = a.round & 1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = a.bp.n, nnz = a.bp.nnz, l8 = lane * 8;
    const bool last = it == a.bp.max_iter;
    const uint64_t unsat = cst->unsat[par] | a.bp.invalid[tile];
    const uint64_t newly = ~unsat & ~done;
    const uint64_t ndone = done | newly;
    const bool over = ndone == ~0ull || last;
    const bool mine = (newly >> lane) & 1ull;
    const bool each = cst->llr_each[par] != 0;
    if (newly || (over && ndone != ~0ull)) {
        uint64_t *dec = a.bp.dec + tile * n;
        const uint64_t *dcur = a.bp.dcur + tile * n;
        const MsgBuf Ct = make_msgbuf(a.bp.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
        const bool want_llr = a.bp.llr_t != nullptr;
        const MsgBuf Lt = make_msgbuf(want_llr ? a.bp.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.bp.A, want_llr ? (unsigned)n : 0u);
        const int j0 = blockIdx.x * 64 + wave * 16;
        for (int j = j0; j < j0 + 16 && j < n; ++j) {
            if (lane == 0) {
                const uint64_t cur = dcur[j];
                uint64_t d = (dec[j] & ~newly) | (cur & newly);
                if (over) d = (d & ndone) | (cur & ~ndone);  // never converged: the last iteration's decisions
                dec[j] = d;
            }
            if (newly && !last && want_llr && !each) {  // (at the last iteration, or under llr_each, the bit pass has stored the posterior already)
                double temp = a.bp.llr0[j];
                for (int p = a.bp.col_ptr[j]; p < a.bp.col_ptr[j + 1]; ++p) temp += Ct.ld(l8, a.bp.csc_edge[p]);
                if (mine) Lt.st(l8, j, temp);
            }
        }
    }
    if (blockIdx.x != 0) continue;
    if (wave == 0) {
        if (mine) st->lane_iter[lane] = it;
        const int64_t b = tile * LDPC_WAVE + lane;
        if (over && b < (a.bp.rows_dev ? (int64_t)a.bp.rows_dev[0] : a.bp.batch)) {
            const int64_t row = a.bp.row_map ? (int64_t)a.bp.row_map[b] : b;
            const bool cv = ((ndone >> lane) & 1ull) != 0;
            if (a.bp.iters) a.bp.iters[row] = cv ? (mine ? it : st->lane_iter[lane]) : a.bp.max_iter;  // bp.hpp:304
            if (a.bp.conv) a.bp.conv[row] = cv ? 1 : 0;
        }
    }
    if (threadIdx.x == 0) {
        st->done[par ^ 1] = ndone;
        st->unsat[par ^ 1] = 0ull;
        st->llr_each[par ^ 1] = (each || newly) ? 1 : 0;  // after the first event: every bit pass stores the live lanes' posteriors
        if (over) {
            st->end_round = a.round;
            if (atomicSub(&a.bp.counters[2], 1u) == 1u && a.host_flag)  // that was the last live tile
                __hip_atomic_store(a.host_flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    } while (LOOP && (slot += (int)gridDim.y) < spread_count(a));
}
