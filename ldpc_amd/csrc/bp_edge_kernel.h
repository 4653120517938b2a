// bp_edge_kernel.h -- bp_edge_kernel: min-sum with LANE = EDGE and the messages in REGISTERS (BASELINE config 3)
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// For the codes of the surface-code family -- rows of weight <= 4, columns of weight <= 2 (rotated / toric surface codes,
// ring codes) -- the node-per-lane kernel (bp_wave_kernel.h) spends ~745 wave-instructions per syndrome-iteration on 840
// edges: address arithmetic, position-table reads, a separate syndrome sweep, every message through LDS twice per pass.
// Here a wavefront still owns one syndrome, but
//   * a LANE owns one EDGE per round (slot s = 64 r + lane, r < R rounds), and a row's (up to) four edges sit in four
//     neighbouring lanes (slot = 4 i + k): the check update (bp.hpp:220-273) needs the other entries of the row -- two
//     DPP quad permutations -- and the row's sign parity -- one ballot and scalar bit tricks on the 64-bit lane mask;
//   * the edge's message lives in a REGISTER for the whole decode (R of them per lane); the bit update (bp.hpp:276-318)
//     needs the one other entry of the column, which sits in some other lane and round: every lane stores its
//     check-to-bit message at its own slot of a wave-private LDS array and reads its partner's (2 LDS operations per edge
//     and iteration instead of 8, and no index tables: the partner's address is a register);
//   * hard decisions never leave the scalar unit: ballots of (log-ratio <= 0), nibble parities, XOR with the syndrome mask.
// Arithmetic, per edge and in the reference's association:
//   check pass   magnitude = min over the OTHER entries of |bit_to_check| (DBL_MAX if there is none) -- min is exact and
//                order-free, NaNs are skipped as `abs < temp` skips them (bp.hpp:241-247, :257-259, :268-270; v_min_f64
//                returns the other operand for a quiet NaN); sign = parity(syndrome byte + #{entries <= 0}) + own
//                (bp.hpp:236-262); message = magnitude * (+-alpha) (bp.hpp:264-266).
//   bit pass     column entries c0 (lower row), c1:  log-ratio = (prior + c0) + c1 (bp.hpp:278-281);
//                bit_to_check_0 = prior + (0.0 + c1), bit_to_check_1 = (prior + c0) + 0.0 (bp.hpp:279, :313-316).
//                Both equal prior + (the other entry) EXACTLY: x + 0.0 differs from x only for x = -0.0, and neither
//                0.0 + c1 feeding a sum with the prior nor prior + c0 can be a -0.0 that matters -- the prior is
//                log((1 - p) / p), never -0.0, and a sum is -0.0 only if both terms are.  A column of weight one reads
//                a slot that holds +0.0 for good: prior + 0.0.
// The reference's minimum starts from DBL_MAX and replaces it only by something SMALLER (bp.hpp:240-247): an infinite
// |bit_to_check| never enters it.  Hence the clamp min(., DBL_MAX) after the butterfly -- and hence phantom lanes (a row
// lighter than four, the padding behind the last row) may hold +inf for good (prior +inf, partner = a slot that holds +inf): clamped
// to DBL_MAX in the minimum, positive in the sign ballot, and their "log-ratio" is +inf or, at worst, inf - inf = NaN:
// never <= 0, so they drop out of the decision ballots by themselves.
// Results are bit-identical to bp_wave_kernel and to the reference (tests/test_gpu_parity.py, test_gpu_fuzz.py).
struct EdgeArgs {
    int32_t m, n, max_iter;
    double ms_scaling_factor;
    int64_t batch;
    const double *prior_s;     // [R * 64] prior of the slot's column; phantom: +inf          (general form)
    double prior_u;            // the one prior of every column                              (UNIFORM form: no prior registers)
    const uint16_t *partner;   // [R * 64] slot of the other entry of the slot's column; none: R * 64 (the +0.0 slot); a phantom lane:
                               // R * 64 + 1 (a slot that holds +inf for good)
    int32_t chunk;             // syndromes pulled per visit to a work counter
    int32_t static_per, dyn_base, pool_per;  // work_pool_next: static share per wavefront, where the pools start, syndromes per pool (batch < 2^30)
    const uint8_t *kind;       // [R * 64] 0 phantom, 1 first entry of its column (lower row), 2 second entry
    const int32_t *scol;       // [R * 64] column of the slot (outputs are written by the kind-1 lanes)
    const uint8_t *synd;       // [batch][m]
    uint8_t *decoding;         // [batch][n]
    double *llr;               // [batch][n] or nullptr
    int32_t *iters;            // [batch] or nullptr
    uint8_t *conv;             // [batch] or nullptr
    unsigned long long *next;  // WORK_POOLS work counters, WORK_POOL_STRIDE words apart (work_pool_next, bp_device_common.h) (zeroed before launch)
    unsigned long long *clk;   // shader-clock probe (clock_probe_*, bp_device_common.h) or nullptr
};

__host__ __device__ inline size_t edge_lds_bytes(int rounds) { return ((size_t)rounds * 64 + 3) * 8; }  // slots, +0.0, +inf, a dummy (bp_edge8_kernel's phantom lanes park their output there)

// How many of the R rounds let the vector unit do what the scalar unit would (both issue one instruction per SIMD turn, and the
// kernel's scalar work -- lane-mask parities -- outweighs its vector work): measured on BASELINE config 3, tools/bench_edge.py
// (C3, M syndromes/s: none 60.9; V2 all rounds 64.3; V1 all rounds 62.0; both 66.1; V1 = 2 rounds + V2 all 66.8)
#ifndef EDGE_V1
#define EDGE_V1 2
#endif
#ifndef EDGE_V1N  // ... in the form without the clamp (one vector instruction a round fewer: the balance moves; C3, M syndromes/s, x3 on one box:
#define EDGE_V1N 8  //     2: 65.7;  3: 66.2;  4: 66.2;  6: 66.7;  8: 66.7;  10: 66.3;  14: 66.3 -- flat from 6 up, profiles/r6_c3_noclamp_ab.txt)
#endif
#ifndef EDGE_V2
#define EDGE_V2 16
#endif
#ifndef EDGE_G
#define EDGE_G 4
#endif
namespace edge_detail {
// quad permutations of a double (two 32-bit DPP moves: 64-bit DPP allows row_newbcast only)
template <int CTRL>
__device__ __forceinline__ double quad_perm(double x) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// min(|a|, |b|) as ONE v_min_f64 with source modifiers (fmin() would first canonicalise both operands: two more instructions);
// a quiet NaN operand yields the other operand, which is how `if (abs < temp) temp = abs` treats it
__device__ __forceinline__ double min_abs(double a, double b) {
    double r;
    asm("v_min_f64 %0, |%1|, |%2|" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ double fmin_pos(double a, double uniform_b) {  // both >= +0 or NaN; the second from an SGPR pair
    double r;
    asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(uniform_b));
    return r;
}
// per lane: bit `lane` of the wave-uniform mask selects b, else a -- one v_cndmask_b32 reading the mask from an SGPR pair
__device__ __forceinline__ int select_by_mask(int a, int b, uint64_t mask) {
    int r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}
__device__ __forceinline__ double uniform_f64(double x) {  // wave-uniform value -> SGPR pair
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
}
// every bit of a nibble := XOR of the nibble's four bits (a row's four lanes)
__device__ __forceinline__ uint64_t nibble_parity_low(uint64_t x) {  // result in bit 0 of every nibble; the other bits are garbage
    x ^= x >> 1;
    x ^= x >> 2;
    return x;
}
__device__ __forceinline__ uint64_t spread_nibble(uint64_t low) {  // bit 0 of every nibble -> all four bits
    const uint32_t a = ((uint32_t)low & 0x11111111u) * 15u, b = ((uint32_t)(low >> 32) & 0x11111111u) * 15u;
    return ((uint64_t)b << 32) | a;
}
}  // namespace edge_detail

// UNIFORM: all columns have the same prior (a decoder built from `error_rate`): it is a scalar, which frees 2 R registers per
// lane -- a fifth resident wavefront per SIMD; the phantom lanes' +inf then comes from their partner slot instead of their prior.
// NOCLAMP (with UNIFORM): the clamp to DBL_MAX cannot bite and is left out (one vector instruction of ~19 per round, and the kernel is bound
// by vector issue).  The host grants it (plan_edge) when the prior is finite, |alpha| <= 1 and every row has at least two entries: then a
// real lane's minimum always covers a real entry, every message is bounded by (iterations + 1) x |prior| -- |check_to_bit| <= the largest
// |bit_to_check| of the round, |bit_to_check| <= |prior| + |check_to_bit| of its one partner -- and no infinity or NaN ever reaches a real
// lane; the phantom lanes' own infinities stay among themselves as before (inf - inf = NaN is not <= 0 either).
template <int R, bool UNIFORM, bool NOCLAMP = false>
__global__ void __launch_bounds__(64, UNIFORM ? 5 : 4) bp_edge_kernel(const EdgeArgs a) {
    using namespace edge_detail;
    typedef EdgeArgs ARGS_T;  // (cold fields: LDPC_KERNARG, bp_device_common.h)
    extern __shared__ __attribute__((aligned(16))) unsigned char edge_lds[];
    typedef __attribute__((address_space(3))) double lds_f64;
    lds_f64 *X = (lds_f64 *)edge_lds;  // [R * 64] check_to_bit of every slot, [R * 64] = +0.0 for good
    const int lane = threadIdx.x;
    __shared__ unsigned long long clk_stamp[2];
    if (lane == 0) clock_probe_begin(clk_stamp);
    const int m = a.m, n = a.n;
    constexpr int ZERO = R * 64;
    constexpr uint64_t LOW = 0x1111111111111111ull;

    // per lane and round, for the whole kernel: prior, partner address; per round: which lanes are first / second entries
    double prv[UNIFORM ? 1 : R], msg[R];
    int paddr[R];
    uint64_t k0[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = r * 64 + lane;
        if (!UNIFORM) prv[r] = a.prior_s[s];
        paddr[r] = (int)a.partner[s];
        k0[r] = __ballot(a.kind[s] == 1);
    }
    const double pu = uniform_f64(a.prior_u);
#define LDPC_EDGE_PRIOR(r) (UNIFORM ? pu : prv[UNIFORM ? 0 : (r)])
    const double dbl_max = uniform_f64(DBL_MAX);
    if (lane == 0) { X[ZERO] = 0.0; X[ZERO + 1] = __builtin_inf(); }

    // Work: the static share, then chunks from the pooled work counters (work_pool_next, bp_device_common.h)
    int b0 = (int)blockIdx.x * LDPC_KERNARG(ARGS_T, static_per), b1 = b0 + LDPC_KERNARG(ARGS_T, static_per);
    int pool = (int)(blockIdx.x & (WORK_POOLS - 1));
    for (;;) {
      for (int b = b0; b < b1; ++b) {
        // this syndrome's bytes as lane masks: bit 4 q of sy[r] = (byte & 1) of the row that lanes 4 q .. 4 q + 3 serve in round r;
        // a byte above 1 can never be matched (bp.hpp:300)
        uint64_t sy[R];
        bool never = false;
        const auto sb = global_ptr(LDPC_KERNARG(ARGS_T, synd)) + (int64_t)b * m;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = r * 16 + (lane >> 2);
            const int byte = row < m ? (int)sb[row] : 0;
            sy[r] = __ballot((byte & 1) != 0) & LOW;  // (kept in bit 0 of every nibble only: one set of masks for both passes)
            never = never || __ballot(byte > 1) != 0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) msg[r] = UNIFORM ? (paddr[r] == ZERO + 1 ? __builtin_inf() : pu) : prv[r];  // initialise_log_domain_bp (bp.hpp:147-157)

        int it = 0;
        bool unsat = true;
        do {
            ++it;
            const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
            const int alo = __double2loint(alpha), ahi = __double2hiint(alpha), nhi = ahi ^ (int)0x80000000;
            // ---- check pass: msg[r] (bit_to_check) -> msg[r] (check_to_bit), stored at the slot ----
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double cur = msg[r];
                const uint64_t neg = __ballot(cur <= 0.0);
                const double x1 = quad_perm<0xB1>(cur);               // lane ^ 1
                const double pairmin = min_abs(cur, x1);
                const double other = quad_perm<0x4E>(pairmin);        // the other pair's minimum (lane ^ 2)
                const double mag = NOCLAMP ? min_abs(x1, other) : fmin_pos(min_abs(x1, other), dbl_max);  // over the three other entries, from DBL_MAX down
                int shi;  // high word of +-alpha: sign = row parity (syndrome included) + own
                if (r < (NOCLAMP ? EDGE_V1N : EDGE_V1)) {
                    // the vector unit spreads the row's parity: lane 0 of the quad picks it up from the (unspread) scalar mask, a DPP
                    // broadcast inside the quad XORs it onto everyone's own sign -- 2 vector instructions more, 5 scalar ones fewer
                    const uint64_t par = nibble_parity_low(neg ^ sy[r]);
                    const int own = select_by_mask(ahi, nhi, neg);
                    const int rowbit = select_by_mask(0, (int)0x80000000, par);
                    shi = own ^ __builtin_amdgcn_mov_dpp(rowbit, 0x00, 0xf, 0xf, true);  // quad_perm [0, 0, 0, 0]
                } else {
                    const uint64_t flip = spread_nibble(nibble_parity_low(neg ^ sy[r])) ^ neg;  // row parity incl. the syndrome, own sign out
                    shi = select_by_mask(ahi, nhi, flip);
                }
                const double c = mag * __hiloint2double(shi, alo);
                msg[r] = c;
                X[r * 64 + lane] = c;
            }
            // ---- bit pass: the partner's message; log-ratio, decision, new bit_to_check ----
            uint64_t bad = 0;
            // (groups of EDGE_G rounds: their LDS reads are issued together, then consumed -- one round trip per group, not per round)
#pragma unroll
            for (int r0 = 0; r0 < R; r0 += EDGE_G) {
                double cpv[EDGE_G];
#pragma unroll
                for (int g = 0; g < EDGE_G; ++g)
                    if (r0 + g < R) cpv[g] = X[paddr[r0 + g]];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < EDGE_G; ++g) {
                    const int r = r0 + g;
                    if (r >= R) break;
                    const double cp = cpv[g];
                    const double c = msg[r];
                    const double b2c = LDPC_EDGE_PRIOR(r) + cp;
                    const double l1 = b2c + c;                       // second entry of its column: (prior + c0) + c1 with c0 = the partner's
                    const double l0 = (LDPC_EDGE_PRIOR(r) + c) + cp; // first entry: c0 = its own
                    uint64_t d;
                    if (r < EDGE_V2) {  // the vector unit picks the lane's own log-ratio: 1 vector instruction more, 3 scalar ones fewer
                        const double l = __hiloint2double(select_by_mask(__double2hiint(l1), __double2hiint(l0), k0[r]),
                                                          select_by_mask(__double2loint(l1), __double2loint(l0), k0[r]));
                        d = __ballot(l <= 0.0);
                    } else {
                        const uint64_t d1 = __ballot(l1 <= 0.0);
                        d = d1 ^ ((__ballot(l0 <= 0.0) ^ d1) & k0[r]);  // (phantom lanes: neither)
                    }
                    bad |= nibble_parity_low(d) ^ sy[r];  // candidate syndrome vs syndrome (bp.hpp:292-302), bit 0 of every nibble
                    msg[r] = b2c;
                }
            }
            unsat = never || (bad & LOW) != 0;
        } while (unsat && it < a.max_iter);

        // ---- outputs (bp.hpp:62,65,69,71): the log-ratios, formed from the last iteration's messages by the lanes that own the first entry
        //      of a column, change places with the messages (X[j] = log-ratio of column j; n <= R * 64: every column has an entry, and
        //      the next syndrome's first check pass rewrites every slot), then leave in whole 512- / 64-byte rows ----
#pragma unroll
        for (int r = 0; r < R; ++r) msg[r] = (LDPC_EDGE_PRIOR(r) + X[r * 64 + lane]) + X[paddr[r]];
        const auto scol_t = global_ptr(LDPC_KERNARG(ARGS_T, scol));
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int j = scol_t[r * 64 + lane];
            if ((k0[r] >> lane) & 1ull) X[j] = msg[r];
        }
        {
            const auto dp = global_ptr(LDPC_KERNARG(ARGS_T, decoding)) + (int64_t)b * n;
            auto lp = global_ptr(LDPC_KERNARG(ARGS_T, llr));
            if (lp) lp += (int64_t)b * n;
            for (int j = lane; j < n; j += 64) {
                const double l0 = X[j];
                dp[j] = l0 <= 0.0 ? 1 : 0;
                if (lp) lp[j] = l0;
            }
        }
        {
            const auto ip = global_ptr(LDPC_KERNARG(ARGS_T, iters));
            const auto cp = global_ptr(LDPC_KERNARG(ARGS_T, conv));
            if (lane == 0) {
                if (ip) ip[b] = it;
                if (cp) cp[b] = unsat ? 0 : 1;
            }
        }
        // The wavefront must be whole again before lane 0 pulls the next syndrome: without this (convergent) barrier the
        // compiler threads this `lane == 0` block into the one at the top of the loop and the readfirstlane there runs with
        // lane 0 masked off -- every other lane's `pulled` is 0, i.e. syndrome 0 for ever (seen with R = 1).
        __builtin_amdgcn_wave_barrier();
      }
        if (!work_pool_next(LDPC_KERNARG(ARGS_T, next), LDPC_KERNARG(ARGS_T, dyn_base), LDPC_KERNARG(ARGS_T, pool_per), LDPC_KERNARG(ARGS_T, chunk),
                            (int)LDPC_KERNARG(ARGS_T, batch), lane, pool, b0, b1)) break;
    }
    if (lane == 0) clock_probe_end(LDPC_KERNARG(ARGS_T, clk), clk_stamp);
#undef LDPC_EDGE_PRIOR
}

// ---- the same idea for heavier nodes: rows of weight <= 8 in EIGHT neighbouring lanes, columns of weight <= DC (<= 4) ----------
// Bivariate-bicycle codes (rows of 6, columns of 3: BB [[72,12,6]] ... [[144,12,12]]), small hypergraph products.  Differences to
// bp_edge_kernel:
//   * the butterfly over the other entries of a row has a third step across the two quads of the 8-lane group (row_shl:4 / row_shr:4
//     DPP moves, each writing the banks it is meant for), the sign parity is a BYTE parity of the lane mask;
//   * a column has up to DC entries, so a lane reads ALL of them from the wave-private LDS array in column order (its own
//     included; lighter columns: the +0.0 slot behind their entries) and forms the reference's sums with them:
//         log-ratio            (((prior + c0) + c1) + c2) + c3                                   (bp.hpp:278-281), the same in every lane of a column,
//         bit_to_check of k    ((prior + c0) + ... + c_{k-1})  +  (((0 + c_{DC-1}) + ...) + c_{k+1})  (bp.hpp:279, 311-318)
//     (0.0 + x and x + 0.0 are dropped where x + a prior follows or the sum contains the prior: they can only turn -0.0 into +0.0,
//     which a sum with a prior log((1-p)/p) -- never -0.0 -- does not see); the lane's own k picks its bit_to_check by two or three
//     v_cndmask pairs reading per-round lane masks from SGPRs.
struct Edge8Args {
    int32_t m, n, max_iter;
    double ms_scaling_factor;
    int64_t batch;
    const double *prior_s;     // [R * 64] prior of the slot's column; phantom: +inf          (general form)
    double prior_u;            // the one prior of every column                              (UNIFORM form)
    const uint16_t *cpos;      // [DC][R * 64] slot of the j-th entry of the slot's column, j < DC; beyond the column's weight: R * 64 (the +0.0
                               // slot); a phantom lane: R * 64 + 1 everywhere (a slot that holds +inf for good)
    const uint8_t *kind;       // [R * 64] 0 phantom, 1 + k for the k-th entry of its column
    const int32_t *scol;       // [R * 64] column of the slot; a phantom lane: R * 64 + 2 (the dummy slot)
    int32_t chunk;
    int32_t static_per, dyn_base, pool_per;  // (as EdgeArgs)
    const uint8_t *synd;
    uint8_t *decoding;
    double *llr;
    int32_t *iters;
    uint8_t *conv;
    unsigned long long *next;
    unsigned long long *clk;   // shader-clock probe (clock_probe_*, bp_device_common.h) or nullptr
};

namespace edge_detail {
// lane ^ 4 inside an 8-lane group: two DPP moves per dword, each writing only the banks (groups of 4 lanes) it is right for
__device__ __forceinline__ double xor4(double x) {
    int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x104, 0xf, 0x5, false);   // row_shl:4 -> banks 0, 2 (lanes 0-3, 8-11)
    lo = __builtin_amdgcn_update_dpp(lo, __double2loint(x), 0x114, 0xf, 0xa, false);      // row_shr:4 -> banks 1, 3
    int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x104, 0xf, 0x5, false);
    hi = __builtin_amdgcn_update_dpp(hi, __double2hiint(x), 0x114, 0xf, 0xa, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ uint64_t byte_parity_low(uint64_t x) {  // result in bit 0 of every byte; the other bits are garbage
    x ^= x >> 1;
    x ^= x >> 2;
    x ^= x >> 4;
    return x;
}
__device__ __forceinline__ uint64_t spread_byte(uint64_t low) {  // bit 0 of every byte -> all eight bits
    const uint32_t a = ((uint32_t)low & 0x01010101u) * 255u, b = ((uint32_t)(low >> 32) & 0x01010101u) * 255u;
    return ((uint64_t)b << 32) | a;
}
__device__ __forceinline__ double select_f64(double a, double b, uint64_t mask) {  // per lane: bit `lane` of mask ? b : a
    return __hiloint2double(select_by_mask(__double2hiint(a), __double2hiint(b), mask), select_by_mask(__double2loint(a), __double2loint(b), mask));
}
}  // namespace edge_detail

template <int R, int DC, bool UNIFORM>
__global__ void __launch_bounds__(64, UNIFORM ? 5 : 4) bp_edge8_kernel(const Edge8Args a) {
    using namespace edge_detail;
    typedef Edge8Args ARGS_T;  // (cold fields: LDPC_KERNARG, bp_device_common.h)
    static_assert(DC >= 2 && DC <= 4, "columns of 2 .. 4 entries");
    extern __shared__ __attribute__((aligned(16))) unsigned char edge_lds[];
    typedef __attribute__((address_space(3))) double lds_f64;
    lds_f64 *X = (lds_f64 *)edge_lds;
    const int lane = threadIdx.x;
    __shared__ unsigned long long clk_stamp[2];
    if (lane == 0) clock_probe_begin(clk_stamp);
    const int m = a.m, n = a.n;
    constexpr int ZERO = R * 64;
    constexpr uint64_t LOW = 0x0101010101010101ull;

    double prv[UNIFORM ? 1 : R], msg[R];
    int caddr[R][DC];
    uint64_t kmask[R][DC - 1];  // lanes whose entry is the (j + 1)-th of its column, j < DC - 1
    bool phantom_lane[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int s = r * 64 + lane;
        if (!UNIFORM) prv[r] = a.prior_s[s];
#pragma unroll
        for (int j = 0; j < DC; ++j) caddr[r][j] = (int)a.cpos[(size_t)j * (R * 64) + s];
        const int kd = a.kind[s];
        phantom_lane[r] = kd == 0;
#pragma unroll
        for (int j = 0; j < DC - 1; ++j) kmask[r][j] = __ballot(kd == j + 2);
    }
    const double pu = uniform_f64(a.prior_u);
#define LDPC_EDGE_PRIOR(r) (UNIFORM ? pu : prv[UNIFORM ? 0 : (r)])
    const double dbl_max = uniform_f64(DBL_MAX);
    if (lane == 0) { X[ZERO] = 0.0; X[ZERO + 1] = __builtin_inf(); }

    // Work: the static share, then chunks from the pooled work counters (work_pool_next, bp_device_common.h)
    int b0 = (int)blockIdx.x * LDPC_KERNARG(ARGS_T, static_per), b1 = b0 + LDPC_KERNARG(ARGS_T, static_per);
    int pool = (int)(blockIdx.x & (WORK_POOLS - 1));
    for (;;) {
      for (int b = b0; b < b1; ++b) {
        uint64_t sy[R];
        bool never = false;
        const auto sb = global_ptr(LDPC_KERNARG(ARGS_T, synd)) + (int64_t)b * m;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int row = r * 8 + (lane >> 3);
            const int byte = row < m ? (int)sb[row] : 0;
            sy[r] = __ballot((byte & 1) != 0) & LOW;
            never = never || __ballot(byte > 1) != 0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) msg[r] = phantom_lane[r] ? __builtin_inf() : LDPC_EDGE_PRIOR(r);  // initialise_log_domain_bp (bp.hpp:147-157)

        int it = 0;
        bool unsat = true;
        do {
            ++it;
            const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
            const int alo = __double2loint(alpha), ahi = __double2hiint(alpha), nhi = ahi ^ (int)0x80000000;
            // ---- check pass (bp.hpp:220-273): minimum over the seven other lanes of the group, sign by the group's parity ----
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const double cur = msg[r];
                const uint64_t neg = __ballot(cur <= 0.0);
                const double x1 = quad_perm<0xB1>(cur);                // lane ^ 1
                const double pairmin = min_abs(cur, x1);
                const double otherpair = quad_perm<0x4E>(pairmin);     // lane ^ 2: the other pair of the quad
                const double inquad = min_abs(x1, otherpair);          // the three others of the quad
                const double quadmin = min_abs(pairmin, otherpair);    // ... and the whole quad, for the other quad (values >= 0: |.| is idle)
                const double otherquad = xor4(quadmin);
                const double mag = fmin_pos(min_abs(inquad, otherquad), dbl_max);  // the seven other entries, from DBL_MAX down
                const uint64_t flip = spread_byte(byte_parity_low(neg ^ sy[r])) ^ neg;
                const double c = mag * __hiloint2double(select_by_mask(ahi, nhi, flip), alo);
                msg[r] = c;
                X[r * 64 + lane] = c;
            }
            // ---- bit pass (bp.hpp:276-318): the column's entries in order; log-ratio, decision, the lane's own bit_to_check ----
            uint64_t bad = 0;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                double c[DC];
#pragma unroll
                for (int j = 0; j < DC; ++j) c[j] = X[caddr[r][j]];
                const double pr = LDPC_EDGE_PRIOR(r);
                double pre[DC];  // pre[k] = prior + c0 + ... + c_{k-1}
                double t = pr;
#pragma unroll
                for (int j = 0; j < DC; ++j) { pre[j] = t; t += c[j]; }
                const uint64_t d = __ballot(t <= 0.0);  // (phantom lanes: +inf or NaN, never <= 0)
                bad |= byte_parity_low(d) ^ sy[r];
                // the lane's own bit_to_check: cand[k] = pre[k] + (((0 + c_{DC-1}) + ...) + c_{k+1}), accumulated downwards as the reference
                // does (bp.hpp:311-318); the k-th entry of its column takes cand[k]
                double cand[DC];
                double sfx = c[DC - 1];
                cand[DC - 1] = pre[DC - 1];
                cand[DC - 2] = pre[DC - 2] + sfx;
#pragma unroll
                for (int k = DC - 3; k >= 0; --k) { sfx += c[k + 1]; cand[k] = pre[k] + sfx; }
                double b2c = cand[0];
#pragma unroll
                for (int k = 1; k < DC; ++k) b2c = select_f64(b2c, cand[k], kmask[r][k - 1]);
                msg[r] = b2c;
            }
            unsat = never || (bad & LOW) != 0;
        } while (unsat && it < a.max_iter);

        // ---- outputs: as bp_edge_kernel -- every lane of a column holds the column's log-ratio (the same bits) and parks it at X[column]
        //      (phantom lanes: at the dummy slot behind +inf), then whole rows leave ----
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double t = LDPC_EDGE_PRIOR(r);
#pragma unroll
            for (int j = 0; j < DC; ++j) t += X[caddr[r][j]];
            msg[r] = t;
        }
        const auto scol_t = global_ptr(LDPC_KERNARG(ARGS_T, scol));
#pragma unroll
        for (int r = 0; r < R; ++r) X[scol_t[r * 64 + lane]] = msg[r];
        {
            const auto dp = global_ptr(LDPC_KERNARG(ARGS_T, decoding)) + (int64_t)b * n;
            auto lp = global_ptr(LDPC_KERNARG(ARGS_T, llr));
            if (lp) lp += (int64_t)b * n;
            for (int j = lane; j < n; j += 64) {
                const double t = X[j];
                dp[j] = t <= 0.0 ? 1 : 0;
                if (lp) lp[j] = t;
            }
        }
        {
            const auto ip = global_ptr(LDPC_KERNARG(ARGS_T, iters));
            const auto cp = global_ptr(LDPC_KERNARG(ARGS_T, conv));
            if (lane == 0) {
                if (ip) ip[b] = it;
                if (cp) cp[b] = unsat ? 0 : 1;
            }
        }
        __builtin_amdgcn_wave_barrier();  // (see bp_edge_kernel)
      }
        if (!work_pool_next(LDPC_KERNARG(ARGS_T, next), LDPC_KERNARG(ARGS_T, dyn_base), LDPC_KERNARG(ARGS_T, pool_per), LDPC_KERNARG(ARGS_T, chunk),
                            (int)LDPC_KERNARG(ARGS_T, batch), lane, pool, b0, b1)) break;
    }
    if (lane == 0) clock_probe_end(LDPC_KERNARG(ARGS_T, clk), clk_stamp);
#undef LDPC_EDGE_PRIOR
}
