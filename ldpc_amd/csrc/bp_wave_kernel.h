// bp_wave_kernel.h -- bp_wave_kernel: one wavefront decodes one syndrome of a small code out of LDS
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// ---- wavefront-per-syndrome variant for small codes with bounded degrees (BASELINE configs 3 and 5) --------
// Lane = NODE.  A wavefront owns one syndrome; its messages live in a wave-private LDS region, the check pass gives
// lane l the checks l, l+64, ..., the bit pass the bits l, l+64, ...  Nothing but wave-level ordering is needed between
// the passes (LDS operations of a wavefront complete in order), so there is NO workgroup barrier in the decode loop and
// a wavefront that finishes its syndrome pulls the next one from a device-wide counter at once: work is proportional
// to the iterations each syndrome really needs.
//
// ONE message array M, updated in place by both passes (bit->check messages before a check pass, check->bit messages
// after it): LDS capacity is what limits the wavefronts per compute unit here, and latency hiding needs them.
// M is "structure of arrays" by rows with the padded stride mp = roundup(m, 64): the k-th entry of row i sits at
// M[k * mp + i], k < DR (the template bound).  The check pass reads and writes unit-stride, conflict-free, with no
// index lookup; the bit pass reaches the k-th entry of column j through a position table apos[k * np + j] (u16).
// Rows / columns lighter than the bound and the padding nodes own PHANTOM entries: in a row they hold the neutral
// element of the check update for good (min-sum +DBL_MAX, product-sum 1.0; never written: the stores of phantom results
// are redirected to a dummy slot), in a column they all point at one slot that holds +0.0 for good.  Neutral
// elements do not change a single bit -- min(x, DBL_MAX) = x, x * 1.0 = x, and a partial sum is never -0.0 (priors are
// log((1-p)/p), never -0.0), so x + 0.0 = x -- and they let the min-sum arithmetic run without a per-entry branch;
// product-sum only guards its transcendentals.  Per node the entries are walked in the reference's order with the
// reference's two sweeps (bp.hpp:205-218, 278-281 + 313-316): results are bit-identical to every other kernel here
// and to the reference.
struct WaveArgs {
    int32_t m, n, mp, np, max_iter;
    int32_t llr_direct;          // 1: every bit pass stores its log-ratios straight to a.llr (no LDS copy; lets one more wavefront fit)
    double ms_scaling_factor;
    int64_t batch;
    const uint8_t *rdeg, *cdeg;  // [mp], [np] node degrees (0 for padding nodes)
    const uint16_t *col;         // [DR * mp] column of the k-th entry of row i at [k * mp + i]; phantom: np
    const uint16_t *apos;        // [DC * np] position in M of the k-th entry of column j at [k * np + j]; phantom: DR * mp + 1
    const double *llr0;          // [n]
    const double *prior_g;       // min-sum, optional: [np + 2] the priors padded as the LDS copy would be (1.0 beyond n, DBL_MAX at np) --
                                 // read from here instead of an LDS copy where those 8 (np + 2) bytes buy another resident workgroup
    const uint8_t *synd;         // [batch][m]
    uint8_t *decoding;           // [batch][n]
    double *llr;                 // [batch][n] or nullptr
    int32_t *iters;              // [batch] or nullptr
    uint8_t *conv;               // [batch] or nullptr
    // BP + OSD (host_osd.h): the kernel lists the rows it leaves unconverged itself (any order) and clears their status bytes -- the
    // launches that did so afterwards cost 4 - 5 us each whatever the batch.  All three nullptr in a plain BP decode.
    int32_t *osd_list; unsigned *osd_count; uint8_t *osd_status;
    unsigned long long *next;    // WORK_POOLS work counters (work_pool_next, bp_device_common.h; zeroed before launch)
    int32_t pool_per;            // syndromes per pool
    int32_t lds_shared, lds_per_wave;  // bytes
    unsigned long long *clk;     // shader-clock probe (clock_probe_*, bp_device_common.h) or nullptr
};

// LDS bytes: shared tables of a workgroup / private region of one wavefront (host and device agree through these)
__host__ __device__ inline size_t wave_lds_shared(int mp, int np, int DR, int DC, bool product_sum, bool prior_in_lds = true) {
    size_t b = (prior_in_lds ? (size_t)(np + 2) * 8 : 0) + (size_t)DR * mp * 2 + (size_t)DC * np * 2 + (size_t)mp + (size_t)np;
    b = (b + 15) & ~(size_t)15;
    if (product_sum) b += (size_t)(np + 2) * 8 + 256 * 8;  // edge form of the priors, log table
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t wave_lds_private(int mp, int np, int DR, bool want_llr) {
    size_t b = ((size_t)DR * mp + 2) * 8 + (want_llr ? (size_t)np * 8 : 0) + (size_t)(np / 64 + 1) * 8 + (size_t)mp;
    return (b + 15) & ~(size_t)15;
}

// TEAM: the whole workgroup decodes ONE syndrome at a time (its wavefronts share the rows of the check pass and the columns of
// the bit pass; workgroup barriers between the passes, the syndrome test reduced through an LDS flag).  For the codes whose
// message array leaves room for two or three wavefronts per compute unit (e.g. a 768 x 1600 hypergraph product: 49 KiB a
// syndrome) a lone wavefront per syndrome leaves the CU almost idle; eight or sixteen on one syndrome cut its latency instead.
template <int METHOD, int MATH, int DR, int DC, bool TEAM = false>
__global__ void __launch_bounds__(1024) bp_wave_kernel(const WaveArgs a) {
    // nodes per lane in flight: min-sum has few live values per node, the transcendental chains of product-sum many
    // (four nodes of four / two of eight entries each spilled 62 / 38 VGPRs in the bit pass: heavier columns take fewer nodes per lane)
    constexpr int U = METHOD == LDPC_HIP_MINIMUM_SUM ? (DR <= 4 ? (DC <= 2 ? 4 : 2) : DR <= 8 ? (DC <= 4 ? 2 : 1) : 1) : (DR <= 6 ? 2 : 1);
    // (a team has more wavefronts than the check pass has rounds of 64 U rows -- there are half as many rows as columns --: one row
    // per lane there, so that twice as many wavefronts take part)
    constexpr int UC = TEAM ? 1 : U;
    constexpr bool PS = METHOD == LDPC_HIP_PRODUCT_SUM;
    extern __shared__ __attribute__((aligned(16))) unsigned char wv_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = TEAM ? tid : lane, TS = TEAM ? T : 64;    // this thread within the team that shares a syndrome, the team's size
    const int wt = TEAM ? wave : 0, W = TEAM ? T >> 6 : 1;   // this wavefront within the team, wavefronts of a team
    __shared__ int team_unsat[2];
    __shared__ long long team_b;
    __shared__ unsigned long long clk_stamp[2];
    if (threadIdx.x == 0) clock_probe_begin(clk_stamp);
    auto team_sync = [&]() { if (TEAM) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    const int m = a.m, n = a.n, mp = a.mp, np = a.np, rm = DR * mp, cn = DC * np;
    const bool want_llr = a.llr != nullptr && !a.llr_direct, llr_direct = a.llr != nullptr && a.llr_direct;
    // Every LDS pointer is typed in the LDS address space from the start: generic ("flat") pointers into LDS make this
    // compiler emit null checks against the shared aperture that it then fails to select for some template variants.
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) uint64_t lds_u64;
    lds_u8 *base = (lds_u8 *)wv_lds;
    // shared, read-only after the barriers below: [llr0, np + 2][col][apos][rdeg][cdeg] and for product-sum
    // [edge form of llr0, np + 2][log table].  Entry np of llr0 / its edge form = what a phantom entry of a row holds.
    const bool PG = !PS && a.prior_g != nullptr;  // priors read from global memory (min-sum only)
    lds_f64 *prior = (lds_f64 *)base;
    lds_u16 *col = (lds_u16 *)(prior + (PG ? 0 : np + 2));
    lds_u16 *apos = col + rm;
    lds_u8 *rdeg = (lds_u8 *)(apos + cn);
    lds_u8 *cdeg = rdeg + mp;
    const int ps_off = (int)((((PG ? 0 : (size_t)(np + 2) * 8) + (size_t)rm * 2 + (size_t)cn * 2 + (size_t)mp + (size_t)np) + 15) & ~(size_t)15);
    lds_f64 *pform = PS ? (lds_f64 *)(base + ps_off) : prior;
    lds_f64 *log_tab_l = pform + np + 2;
    const double *log_tab = reinterpret_cast<const double *>(wv_lds + ps_off) + np + 2;  // same place, for the math routines' signature
    if (PS)
        for (int q = tid; q < 256; q += T) log_tab_l[q] = ldpc_math::k_log_tab[q];
    for (int q = tid; q < np; q += T) { if (!PG) prior[q] = q < n ? a.llr0[q] : 1.0; cdeg[q] = a.cdeg[q]; }
    for (int q = tid; q < mp; q += T) rdeg[q] = a.rdeg[q];
    for (int q = tid; q < rm; q += T) col[q] = a.col[q];
    for (int q = tid; q < cn; q += T) apos[q] = a.apos[q];
    if (tid == 0 && !PG) prior[np] = DBL_MAX;
    __syncthreads();
    if (PS) {
        for (int q = tid; q < n; q += T) pform[q] = edge_form<METHOD, MATH>(prior[q]);
        if (tid == 0) pform[np] = 1.0;
        __syncthreads();
    }

    // wave-private: [M DR*mp][dummy][+0.0][posteriors np, if asked for][hard decisions np/64 + 1 words, the last one zero][syndrome bytes mp]
    lds_u8 *mine = base + a.lds_shared + (TEAM ? 0 : wave) * a.lds_per_wave;
    lds_f64 *M = (lds_f64 *)mine;
    lds_f64 *L = M + rm + 2;
    volatile lds_u64 *hardw = (volatile lds_u64 *)(L + (want_llr ? np : 0));
    volatile lds_u8 *sy = (volatile lds_u8 *)(hardw + np / 64 + 1);
    const int DUMMY = rm, ZERO = rm + 1;
    if (tl == 0) { M[ZERO] = 0.0; hardw[np / 64] = 0; team_unsat[0] = 0; team_unsat[1] = 0; }
    for (int q = tl; q < mp; q += TS) sy[q] = 0;
    team_sync();

    int pool = (int)((blockIdx.x * (T / 64) + wave) & (WORK_POOLS - 1));  // work_pool_next (bp_device_common.h)
    bool first_turn = true;
    for (;;) {
        int64_t b;
        if (TEAM && a.next == nullptr) {  // no more syndromes than teams (a single decode()): team g takes syndrome g, no counter to reset or visit
            b = first_turn ? (int64_t)blockIdx.x : a.batch;
            first_turn = false;
        } else if (TEAM) {  // (a team pulls rarely: the first counter alone, which keeps the pool logic's registers out of this form)
            if (tid == 0) team_b = (long long)atomicAdd(a.next, 1ull);
            __syncthreads();
            b = team_b;
        } else {
            int b0 = 0, b1 = 0;
            b = work_pool_next(a.next, 0, a.pool_per, 1, (int)a.batch, lane, pool, b0, b1) ? (int64_t)b0 : a.batch;
        }
        if (b >= a.batch) break;
        // initialise_log_domain_bp (bp.hpp:147-157) + this syndrome's bytes; phantom entries get the neutral element
        for (int i = tl; i < m; i += TS) sy[i] = a.synd[b * m + i];
        if (PG) { for (int q = tl; q < rm; q += TS) M[q] = a.prior_g[col[q]]; }
        else { for (int q = tl; q < rm; q += TS) M[q] = pform[col[q]]; }
        if (TEAM && tid == 0) { team_unsat[0] = 0; team_unsat[1] = 0; }  // (a syndrome that ran out of iterations leaves its last flag raised)
        team_sync();

        int it = 0;
        bool unsat_any = true;
        do {
            ++it;
            const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
            // ---- check pass (bp.hpp:201-273), in place, U rows per lane in flight: loads, arithmetic, stores ----
            for (int i0 = wt * 64 * UC; i0 < mp; i0 += 64 * UC * W) {
                uint8_t sb[UC];
                int d[UC];
                double cur[UC][DR], out[UC][DR];
#pragma unroll
                for (int u = 0; u < UC; ++u)
                    if (i0 + u * 64 < mp) {  // wave-uniform
                        const int i = i0 + u * 64 + lane;
                        sb[u] = sy[i];
                        d[u] = rdeg[i];
#pragma unroll
                        for (int k = 0; k < DR; ++k) cur[u][k] = M[k * mp + i];
                    }
#pragma unroll
                for (int u = 0; u < UC; ++u)
                    if (i0 + u * 64 < mp) {
                        if (PS) {
                            const bool neg = sb[u] != 0;  // bp.hpp:213
                            double temp = 1.0;
#pragma unroll
                            for (int k = 0; k < DR; ++k) { out[u][k] = temp; temp *= cur[u][k]; }
                            temp = 1.0;
#pragma unroll
                            for (int k = DR - 1; k >= 0; --k) {
                                if (k < d[u]) out[u][k] = ps_message<MATH>(out[u][k] * temp, neg, log_tab);
                                temp *= cur[u][k];
                                LDPC_EDGE_FENCE();
                            }
                        } else {
                            int parity = sb[u] & 1;  // total_sgn = syndrome[i] + #{b2c <= 0}, parity only (bp.hpp:236-262)
                            double temp = DBL_MAX;
#pragma unroll
                            for (int k = 0; k < DR; ++k) {
                                if (cur[u][k] <= 0) parity ^= 1;
                                out[u][k] = temp;
                                const double ab = fabs(cur[u][k]);
                                if (ab < temp) temp = ab;
                            }
                            temp = DBL_MAX;
#pragma unroll
                            for (int k = DR - 1; k >= 0; --k) {
                                const int sgn = parity ^ (cur[u][k] <= 0 ? 1 : 0);
                                double mag = out[u][k];
                                if (temp < mag) mag = temp;
                                out[u][k] = mag * (sgn ? -alpha : alpha);
                                const double ab = fabs(cur[u][k]);
                                if (ab < temp) temp = ab;
                            }
                        }
                    }
#pragma unroll
                for (int u = 0; u < UC; ++u)
                    if (i0 + u * 64 < mp) {
                        const int i = i0 + u * 64 + lane;
#pragma unroll
                        for (int k = 0; k < DR; ++k) M[k < d[u] ? k * mp + i : DUMMY] = out[u][k];  // phantom entries keep their neutral value
                    }
            }
            team_sync();
            // ---- bit pass (bp.hpp:276-298, 311-318), in place through the position table, U bits per lane in flight ----
            for (int j0 = wt * 64 * U; j0 < np; j0 += 64 * U * W) {
                int d[U], pos[U][DC];
                double c[U][DC], pre[U][DC], pr[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j0 + u * 64 < np) {
                        const int j = j0 + u * 64 + lane;
                        pr[u] = PG ? a.prior_g[j] : prior[j];
                        if (PS) d[u] = cdeg[j];
#pragma unroll
                        for (int k = 0; k < DC; ++k) { pos[u][k] = apos[k * np + j]; c[u][k] = M[pos[u][k]]; }  // phantom: the +0.0 slot
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j0 + u * 64 < np) {
                        double temp = pr[u];
#pragma unroll
                        for (int k = 0; k < DC; ++k) { pre[u][k] = temp; temp += c[u][k]; }
                        if (want_llr) L[j0 + u * 64 + lane] = temp;
                        if (llr_direct && j0 + u * 64 + lane < n) a.llr[b * n + j0 + u * 64 + lane] = temp;  // the last pass's stay
                        const uint64_t word = __ballot(temp <= 0);  // padding bits: prior 1.0, no entries -> 0
                        if (lane == 0) hardw[(j0 >> 6) + u] = word;
                        double sfx = 0.0;
#pragma unroll
                        for (int k = DC - 1; k >= 0; --k) {
                            if (PS) {
                                if (k < d[u]) pre[u][k] = edge_form<METHOD, MATH>(pre[u][k] + sfx);
                                LDPC_EDGE_FENCE();
                            } else {
                                pre[u][k] = pre[u][k] + sfx;
                            }
                            sfx += c[u][k];
                        }
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j0 + u * 64 < np) {
#pragma unroll
                        for (int k = 0; k < DC; ++k) M[pos[u][k] == ZERO ? DUMMY : pos[u][k]] = pre[u][k];
                    }
            }
            team_sync();
            // ---- syndrome test (bp.hpp:292-294, 300-302): candidate parity of every check vs its syndrome BYTE ----
            bool unsat = false;
            for (int i0 = wt * 64; i0 < mp; i0 += 64 * W) {
                const int i = i0 + lane;
                unsigned par = 0;
#pragma unroll
                for (int k = 0; k < DR; ++k) {
                    const int cj = col[k * mp + i];  // phantom: bit np, always zero
                    par ^= (unsigned)(hardw[cj >> 6] >> (cj & 63)) & 1u;
                }
                unsat |= par != (unsigned)sy[i];
            }
            unsat_any = __ballot(unsat) != 0;
            if (TEAM) {  // (two flags: iteration it + 1 raises the other one, which nobody has read since iteration it - 1)
                if (unsat_any && lane == 0) team_unsat[it & 1] = 1;
                if (tid == 0) team_unsat[(it + 1) & 1] = 0;
                __syncthreads();
                unsat_any = team_unsat[it & 1] != 0;
            }
        } while (unsat_any && it < a.max_iter);

        // ---- outputs (bp.hpp:62,65,69,71) ----
        for (int j = tl; j < n; j += TS) {
            a.decoding[b * n + j] = (uint8_t)((hardw[j >> 6] >> (j & 63)) & 1ull);
            if (want_llr) a.llr[b * n + j] = L[j];
        }
        if (tl == 0) {
            if (a.iters) a.iters[b] = it;
            if (a.conv) a.conv[b] = unsat_any ? 0 : 1;
            if (a.osd_status) a.osd_status[b] = 0;
            if (a.osd_list && unsat_any) a.osd_list[atomicAdd(a.osd_count, 1u)] = (int32_t)b;
        }
        team_sync();
    }
    if (threadIdx.x == 0) clock_probe_end(a.clk, clk_stamp);
}

// ---- product-sum: node-per-lane bookkeeping, ENTRY-per-lane transcendentals ----------------------------------------
// For product-sum the cost of a pass is its transcendentals -- one log per entry in the check pass, one tanh per entry in
// the bit pass.  With lane = node a code like BB [[144,12,12]] (m = 72, n = 144) leaves a third of the lanes idle in the last
// round of each pass while the busy ones evaluate 6 resp. 3 of them in sequence; with lane = entry throughout (round 1 - 2)
// every lane re-did its row's prefix / suffix products and picked its own by selects: ~70 of a round's ~165 vector
// instructions were that bookkeeping (profiles/r3_*c5*).  Now each pass has two phases:
//   check A  lane = row     the reference's two sweeps of the row (bp.hpp:205-218), x_k = prefix_k * suffix_k stored at entry k
//   check B  lane = entry   C[slot] = sign * log((1 + x) / (1 - x)), in place: ONE LDS read, one transcendental, one LDS write
//   bit A    lane = column  the reference's two sweeps of the column (bp.hpp:276-281, 311-318): log-ratio, decision, the new
//                           bit_to_check values stored at their entries
//   bit B    lane = entry   A[slot] = tanh(A[slot] / 2), in place; the syndrome test rides in the same phase (lane = row)
// 432 entries fill 7 rounds of 64 lanes at 96 %; the node phases are 2 and 3 short rounds.  Entries live row-padded, entry k
// of row i at i * DR + k (phantoms hold 1.0 for good), in TWO arrays; the bit pass reaches the k-th entry of column j through
// epos[j * DC + k] (phantom: slot m * DR -- +0.0 for good in C, a dummy in A).  Per node the arithmetic is the node-owning
// lane's of every other kernel here: same operations, same order, same bits.
struct WavePsArgs {
    int32_t m, n, np, max_iter;
    int64_t batch;
    const uint8_t *rdeg;     // [m]
    const uint16_t *col;     // [m * DR] column of entry k of row i at [i * DR + k]; phantom: np
    const uint16_t *epos;    // [np * DC] position (i * DR + k) of the k-th entry of column j; phantom: m * DR
    const double *llr0;      // [n]
    const uint8_t *synd;
    uint8_t *decoding;
    double *llr;
    int32_t *iters;
    uint8_t *conv;
    int32_t *osd_list; unsigned *osd_count; uint8_t *osd_status;  // (as WaveArgs)
    unsigned long long *next;    // WORK_POOLS work counters (work_pool_next; zeroed before launch)
    int32_t pool_per;            // syndromes per pool
    int32_t lds_shared, lds_per_wave;
    int32_t min_rdeg;        // lightest row (a row of weight 1 has x = the empty product 1: q = 2 / 0, generic path only)
    unsigned long long *clk; // shader-clock probe (clock_probe_*, bp_device_common.h) or nullptr
    // RESIDENT form (TEAM, one workgroup, batch = 1; host_onchip.h: decode_onchip_resident): the workgroup does not leave after its
    // syndrome but waits for the next one -- the caller's `for shot: decode(shot)` loop finds the kernel there, tables in LDS, and a
    // decode costs neither a launch nor a completion.  mail: four 32-bit words in the host-mapped block the syndrome and the results
    // live in -- [0] request number (host writes), [1] the last request served (device writes, after the results), [2] alive (host
    // sets 1 before the launch, the device 0 when it leaves), [3] quit (host: leave now).  The workgroup leaves by itself after
    // linger_ticks (100 MHz) without a request: it can never spin for good.
    unsigned *mail;
    unsigned served0, linger_ticks;
};

#define LDPC_PS_NEAR_SLOTS 64  // entries whose log argument is near 1, listed per wavefront and iteration (see check B)
#ifdef LDPC_WPS_PROF  // measurement build (tools/wave_ps_phases.py): shader cycles of wavefront 0 of every workgroup per phase of an iteration, summed
// {check A, check B, bit A, bit B + syndrome test, the iteration's closing barrier / flags, set-up of a syndrome + results out, pulls}, iterations, syndromes
__device__ unsigned long long g_wps_prof[12];
#define WPS_MARK(k) do { if (wave == 0 && lane == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); wps_pf[k] += now_ - wps_t; wps_t = now_; } } while (0)
#else
#define WPS_MARK(k) do { } while (0)
#endif

__host__ __device__ inline size_t wave_ps_lds_shared(int m, int np, int DR, int DC) {
    size_t b = 256 * 8 + (size_t)(np + 2) * 16 + (size_t)m * DR * 2 + (size_t)np * DC * 2 + (size_t)m;
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t wave_ps_lds_private(int m, int np, int DR, bool want_llr) {
    size_t b = 2 * ((size_t)m * DR + 2) * 8 + (want_llr ? (size_t)np * 8 : 0) + LDPC_PS_NEAR_SLOTS * 4 + (size_t)(np + 16) + (size_t)m + (size_t)m * DR;
    return (b + 15) & ~(size_t)15;
}

// TEAM: as for bp_wave_kernel -- the workgroup's wavefronts share one syndrome, each taking rounds of 64 lanes of a phase.  For
// batches so small that a wavefront decodes only a few syndromes the time is the 50 iterations of the slowest one: a team cuts
// exactly that.
// DR > 16 (rows of up to 32 entries: hamming(6) and the like): a row's 2 x 32 prefix / suffix values sit in registers, which needs the
// register budget of a <= 4-wavefront workgroup (the host launches no more).
template <int MATH, int DR, int DC, bool TEAM = false>
__global__ void __launch_bounds__(DR > 16 ? 256 : 1024) bp_wave_ps_kernel(const WavePsArgs a) {
    constexpr int METHOD = LDPC_HIP_PRODUCT_SUM;
    extern __shared__ __attribute__((aligned(16))) unsigned char wv_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tl = TEAM ? tid : lane, TS = TEAM ? T : 64;
    const int wt = TEAM ? wave : 0, W = TEAM ? T >> 6 : 1;
    __shared__ int team_unsat[2];
    __shared__ long long team_b;
    __shared__ unsigned team_req;   // resident form: the request being served, and whether the workgroup leaves after it
    __shared__ int team_leaving;
    __shared__ unsigned long long clk_stamp[2];
    if (threadIdx.x == 0) clock_probe_begin(clk_stamp);
    auto team_sync = [&]() { if (TEAM) __syncthreads(); else __builtin_amdgcn_wave_barrier(); };
    const int m = a.m, n = a.n, np = a.np, rm = m * DR;
    const bool want_llr = a.llr != nullptr;
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    lds_u8 *base = (lds_u8 *)wv_lds;
    unsigned served = a.served0;
    // shared: [log table][llr0 np + 2][edge form of llr0 np + 2][col][epos][rdeg]
    lds_f64 *log_tab_l = (lds_f64 *)base;
    const double *log_tab = reinterpret_cast<const double *>(wv_lds);
    lds_f64 *prior = log_tab_l + 256;
    lds_f64 *pform = prior + np + 2;
    lds_u16 *col = (lds_u16 *)(pform + np + 2);
    lds_u16 *epos = col + rm;
    lds_u8 *rdeg = (lds_u8 *)(epos + np * DC);
    for (int q = tid; q < 256; q += T) log_tab_l[q] = ldpc_math::k_log_tab[q];
    for (int q = tid; q < np; q += T) prior[q] = q < n ? a.llr0[q] : 1.0;
    for (int q = tid; q < m; q += T) rdeg[q] = a.rdeg[q];
    for (int q = tid; q < rm; q += T) col[q] = a.col[q];
    for (int q = tid; q < np * DC; q += T) epos[q] = a.epos[q];
    __syncthreads();
    for (int q = tid; q < n; q += T) pform[q] = edge_form<METHOD, MATH>(prior[q]);
    if (tid == 0) pform[np] = 1.0;  // phantom entries of a row: neutral for the products
    __syncthreads();

    // wave-private: [A rm + 2 (entry rm = a dummy)][C rm + 2 (entry rm = the +0.0 slot)][posteriors np, if asked for][hard decisions np + 16 bytes]
    // [syndrome bytes m][per entry: 0 phantom, 1 real, 2 real in a row whose syndrome byte is not 0 (the message's sign, bp.hpp:213) rm]
    lds_u8 *mine = base + a.lds_shared + (TEAM ? 0 : wave) * a.lds_per_wave;
    lds_f64 *A = (lds_f64 *)mine;
    lds_f64 *C = A + rm + 2;
    lds_f64 *L = C + rm + 2;
    typedef __attribute__((address_space(3))) int lds_i32;
    lds_i32 *near_list = (lds_i32 *)(L + (want_llr ? np : 0));
    volatile lds_u8 *hard = (volatile lds_u8 *)(near_list + LDPC_PS_NEAR_SLOTS);
    volatile lds_u8 *sy = hard + np + 16;
    volatile lds_u8 *flag = sy + m;
    const int ZERO = rm;
    if (tl == 0) { C[ZERO] = 0.0; hard[np] = 0; team_unsat[0] = 0; team_unsat[1] = 0; }
    team_sync();
    // The exact log has two evaluation branches (argument within ~6 % of 1, or not) and a round of 64 entries nearly always
    // holds both kinds, so every round paid for both (37 + 42 instructions).  One wavefront per syndrome: every entry takes the
    // table branch; the few with an argument near 1 are LISTED (compacted over all rounds of the pass: ballot + mbcnt) and
    // the near-1 branch runs once over the list -- usually one round instead of seven.  Precondition, tested where the tanh
    // values are made: every |tanh| < 1 (then q = (1 + x) / (1 - x) is a normal number: no 0 / inf / NaN tails), and no row
    // of weight 1.  Same operations on the same operands as ps_log_ratio_libm: same bits (check_row_ps_exact_fast is the
    // streamed kernels' form of this).
    constexpr bool LISTED = !TEAM && MATH == 0;
    bool prior_tame = true;  // every |tanh(prior / 2)| < 1
    if (LISTED) {
        bool wild = false;
        for (int q = lane; q < n; q += 64) wild = wild || !(__builtin_fabs(pform[q]) < 1.0);
        prior_tame = __ballot(wild) == 0 && a.min_rdeg >= 2;
    }

    int pool = (int)((blockIdx.x * (T / 64) + wave) & (WORK_POOLS - 1));  // work_pool_next (bp_device_common.h)
    bool first_turn = true;
#ifdef LDPC_WPS_PROF
    unsigned long long wps_pf[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, wps_t = __builtin_readcyclecounter();
#endif
    for (;;) {
        int64_t b;
        WPS_MARK(5);
        if (TEAM && a.mail) {  // resident: wait for the next request (see WavePsArgs::mail)
            if (tid == 0) {
                const unsigned long long t_idle = (unsigned long long)__builtin_readsteadycounter();
                unsigned r;
                int leaving = 0;
                for (;;) {
                    r = __hip_atomic_load(a.mail + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (r != served) break;
                    if (__hip_atomic_load(a.mail + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u ||
                        (unsigned long long)__builtin_readsteadycounter() - t_idle > (unsigned long long)a.linger_ticks) {
                        // leaving: say so FIRST, then look once more -- a request posted while this was being decided is either seen here and
                        // served, or its poster sees alive == 0 (store / load on both sides: one of the two must see the other)
                        __hip_atomic_store(a.mail + 2, 0u, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                        r = __hip_atomic_load(a.mail + 0, __ATOMIC_SEQ_CST, __HIP_MEMORY_SCOPE_SYSTEM);
                        leaving = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                team_req = r;
                team_leaving = leaving;
                team_b = r != served ? 0ll : (long long)a.batch;
            }
            __syncthreads();
            b = team_b;
        } else if (TEAM && a.next == nullptr) {  // no more syndromes than teams (a single decode()): team g takes syndrome g, no counter to reset or visit
            b = first_turn ? (int64_t)blockIdx.x : a.batch;
            first_turn = false;
        } else if (TEAM) {  // (a team pulls rarely: the first counter alone, which keeps the pool logic's registers out of this form)
            if (tid == 0) team_b = (long long)atomicAdd(a.next, 1ull);
            __syncthreads();
            b = team_b;
        } else {
            int b0 = 0, b1 = 0;
            b = work_pool_next(a.next, 0, a.pool_per, 1, (int)a.batch, lane, pool, b0, b1) ? (int64_t)b0 : a.batch;
        }
        WPS_MARK(6);
        if (b >= a.batch) break;
#ifdef LDPC_WPS_PROF
        ++wps_pf[8];
#endif
        for (int i = tl; i < m; i += TS) sy[i] = a.synd[b * m + i];
        for (int q = tl; q < rm; q += TS) A[q] = pform[col[q]];  // initialise_log_domain_bp (bp.hpp:147-157)
        if (TEAM && tid == 0) { team_unsat[0] = 0; team_unsat[1] = 0; }  // (a syndrome that ran out of iterations leaves its last flag raised)
        team_sync();
        for (int q = tl; q < rm; q += TS) {
            const int i = q / DR, k = q - i * DR;
            flag[q] = k < (int)rdeg[i] ? (sy[i] != 0 ? 2 : 1) : 0;
        }
        team_sync();

        int it = 0;
        bool unsat_any = true;
        bool tame = prior_tame;  // wave-uniform: the tanh values of this iteration's check pass all lie inside (-1, 1)
        do {
            ++it;
            WPS_MARK(5);
#ifdef LDPC_WPS_PROF
            ++wps_pf[7];
#endif
            // ---- check A (bp.hpp:205-209, 211-212, 217): lane = row, both sweeps; x_k = prefix_k * suffix_k parked at entry k ----
            for (int i0 = wt * 64; i0 < m; i0 += 64 * W) {
                const int i = i0 + lane;
                if (i < m) {
                    double av[DR], pre[DR];
#pragma unroll
                    for (int kk = 0; kk < DR; ++kk) av[kk] = A[i * DR + kk];  // phantoms: 1.0, behind the real entries
                    double temp = 1.0;
#pragma unroll
                    for (int kk = 0; kk < DR; ++kk) { pre[kk] = temp; temp *= av[kk]; }
                    temp = 1.0;
#pragma unroll
                    for (int kk = DR - 1; kk >= 0; --kk) { C[i * DR + kk] = pre[kk] * temp; temp *= av[kk]; }
                }
            }
            team_sync();
            WPS_MARK(0);
            // ---- check B (bp.hpp:213-216): lane = entry, one log each, in place ----
            if (LISTED && tame) {
                int total = 0;
                for (int s0 = 0; s0 < rm; s0 += 64) {
                    const int slot = s0 + lane;
                    const int f = slot < rm ? (int)flag[slot] : 0;
                    const double x = f ? C[slot] : 0.0;
                    const double q = ldpc_math::div_cr(1.0 + x, 1.0 - x);
                    const bool near = f != 0 && ldpc_math::log_near_one(q);
                    const double y = ldpc_math::log_libm_general(q, log_tab);
                    const uint64_t mask = __ballot(near);
                    bool in_place = false;
                    if (mask) {
                        const int cnt = __builtin_popcountll(mask);
                        if (total + cnt <= LDPC_PS_NEAR_SLOTS) {
                            if (near) near_list[total + lane_rank(mask)] = slot;  // (its x stays in C[slot] for the second visit)
                            total += cnt;
                        } else in_place = near;  // list full: as the generic routine would
                    }
                    if (f && !near) C[slot] = f == 2 ? -y : y;
                    if (in_place) { const double yn = ldpc_math::log_libm_near_one(q); C[slot] = f == 2 ? -yn : yn; }
                }
                for (int c = lane; c < total; c += 64) {
                    const int slot = near_list[c];
                    const double x = C[slot];
                    const double yn = ldpc_math::log_libm_near_one(ldpc_math::div_cr(1.0 + x, 1.0 - x));
                    C[slot] = flag[slot] == 2 ? -yn : yn;
                }
            } else {
                for (int s0 = wt * 64; s0 < rm; s0 += 64 * W) {
                    const int slot = s0 + lane;
                    const int f = slot < rm ? (int)flag[slot] : 0;
                    if (f) C[slot] = ps_message<MATH>(C[slot], f == 2, log_tab);
                }
            }
            team_sync();
            WPS_MARK(1);
            // ---- bit A (bp.hpp:276-298, 311-318): lane = column, both sweeps; the new bit_to_check values parked at their entries ----
            for (int j0 = wt * 64; j0 < n; j0 += 64 * W) {
                const int j = j0 + lane;
                if (j < n) {
                    int pos[DC];
                    double cv[DC], pre[DC];
#pragma unroll
                    for (int kk = 0; kk < DC; ++kk) { pos[kk] = epos[j * DC + kk]; cv[kk] = C[pos[kk]]; }  // phantom: the +0.0 slot, behind the real entries
                    double temp = prior[j];
#pragma unroll
                    for (int kk = 0; kk < DC; ++kk) { pre[kk] = temp; temp += cv[kk]; }
                    hard[j] = temp <= 0 ? 1 : 0;
                    if (want_llr) L[j] = temp;
                    double sfx = 0.0;
#pragma unroll
                    for (int kk = DC - 1; kk >= 0; --kk) { A[pos[kk]] = pre[kk] + sfx; sfx += cv[kk]; }  // (phantom: A's dummy slot)
                }
            }
            team_sync();
            WPS_MARK(2);
            // ---- bit B: lane = entry, one tanh each, in place (A holds tanh(bit_to_check / 2), bp.hpp:208) ----
            bool wild = false;
            for (int s0 = wt * 64; s0 < rm; s0 += 64 * W) {
                const int slot = s0 + lane;
                if (slot < rm && flag[slot]) {
                    const double z = edge_form<METHOD, MATH>(A[slot]);
                    A[slot] = z;
                    if (LISTED) wild = wild || !(__builtin_fabs(z) < 1.0);
                }
            }
            if (LISTED) tame = __ballot(wild) == 0 && a.min_rdeg >= 2;
            // ---- syndrome test (bp.hpp:292-294, 300-302), same phase: it reads the decisions only ----
            bool unsat = false;
            for (int i = tl; i < m; i += TS) {
                unsigned par = 0;
#pragma unroll
                for (int kk = 0; kk < DR; ++kk) par ^= hard[col[i * DR + kk]];  // phantom: byte np, always zero
                unsat |= par != (unsigned)sy[i];
            }
            unsat_any = __ballot(unsat) != 0;
            WPS_MARK(3);
            if (TEAM) {  // (two flags: iteration it + 1 raises the other one, which nobody has read since iteration it - 1)
                if (unsat_any && lane == 0) team_unsat[it & 1] = 1;
                if (tid == 0) team_unsat[(it + 1) & 1] = 0;
                __syncthreads();
                unsat_any = team_unsat[it & 1] != 0;
            }
            WPS_MARK(4);
        } while (unsat_any && it < a.max_iter);

        for (int j = tl; j < n; j += TS) {
            a.decoding[b * n + j] = hard[j];
            if (want_llr) a.llr[b * n + j] = L[j];
        }
        if (tl == 0) {
            if (a.iters) a.iters[b] = it;
            if (a.conv) a.conv[b] = unsat_any ? 0 : 1;
            if (a.osd_status) a.osd_status[b] = 0;
            if (a.osd_list && unsat_any) a.osd_list[atomicAdd(a.osd_count, 1u)] = (int32_t)b;
        }
        if (TEAM && a.mail) {  // resident: the results are in the host's memory -- then, and only then, the request counts as served
            __threadfence_system();
            __syncthreads();
            served = team_req;
            const bool leaving = team_leaving != 0;
            if (tid == 0) __hip_atomic_store(a.mail + 1, served, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (leaving) break;
        }
        team_sync();
    }
    if (threadIdx.x == 0) clock_probe_end(a.clk, clk_stamp);
#ifdef LDPC_WPS_PROF
    if (tid == 0)
        for (int k = 0; k < 9; ++k) atomicAdd(&g_wps_prof[k], wps_pf[k]);
#endif
}
