// bp_device_common.h -- argument blocks, message addressing and the per-node arithmetic shared by every BP kernel
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <cstddef>

#include "../../include/ldpc_hip.h"
#include "bp_math.h"

// math modes of the product-sum kernel (ldpc_hip_bp_set_math): 0 = bit-identical twins of the host
// glibc the reference runs on (default), 1 = fast ~1-ulp routines (bp_math.h)

#define LDPC_WAVE 64  // gfx950 wavefront; also the tile width (syndromes per workgroup)


struct BpArgs {
    int32_t m, n, nnz, max_iter;
    double ms_scaling_factor;
    int64_t batch;       // syndromes in this launch (last tile may be partial)
    const int32_t *row_ptr, *col_idx;   // CSR
    const int32_t *col_ptr, *csc_edge;  // CSC: CSR edge id of each column entry, rows ascending
    const double *llr0;                 // initial_log_prob_ratios (bp.hpp:66), host-computed
    double *A;                          // bit_to_check_msg  [tiles][nnz][64]  (bp.hpp:44)
    double *C;                          // check_to_bit_msg  [tiles][nnz][64]  (bp.hpp:45)
    const uint64_t *par;                // [tiles][m]  bit l = syndrome byte & 1 of lane l
    const uint64_t *nzm;                // [tiles][m]  bit l = syndrome byte != 0 of lane l
    const uint64_t *invalid;            // [tiles]     bit l = some syndrome byte > 1 (never converges)
    uint64_t *dec;                      // [tiles][n]  frozen hard decisions, bit l = lane l (zero-initialised)
    uint64_t *dcur;                     // [tiles][n]  hard decisions of the running iteration
    double *llr_t;                      // [tiles][n][64] or nullptr
    int32_t *iters;                     // [batch] or nullptr
    uint8_t *conv;                      // [batch] or nullptr
    // hand-off of straggler tiles to the chip-wide per-pass kernels (see bp_spread_*): 0 = never
    struct TileState *state;            // [tiles]
    unsigned *counters;                 // [0] tiles finished by the persistent kernel, [1] tiles handed off, [2] handed-off tiles still running
    int32_t *handoff_list;              // [tiles] ids of handed-off tiles
    int32_t total_tiles, handoff_threshold;
    // [n] what an edge of column j holds before the first iteration (product-sum: tanh(llr0[j] / 2), min-sum: llr0[j]), or nullptr.
    // With it the persistent kernel (ring variant) neither writes the initial messages nor reads them back: the first check
    // pass takes its inputs from this table through the scalar cache -- one array write and one array read less per decode
    // (0.5 of 50 iterations on the headline, 0.5 of ~9 where everything converges early).
    const double *edge0;
    // > 0: the message array A already holds the bit_to_check state after `it_start` iterations (lanes compacted out of the tiles
    // of a first pass, decode_stream_repacked): no initialisation, iterations count on from it_start + 1
    int32_t it_start;
    // 1: the bit pass of iteration max_iter still writes the bit_to_check messages (a first pass whose state a second pass carries
    // on, decode_stream_repacked).  0: nobody reads them -- the last bit pass only forms the log-ratios (no tanh, no message stores)
    int32_t keep_state;
    // Rows known to the device only (second pass of a compacted decode, host_stream.h: decode_stream_repacked).  rows_dev (if not null):
    // [0] the number of rows of this launch, [1] its number of 64-row tiles -- they override `batch` and `total_tiles`, and workgroups
    // beyond the last tile leave at once (the grid is sized for the most rows there can be).  row_map (if not null): row r of the launch
    // is row row_map[r] of the caller's `iters` / `conv`.
    const unsigned *rows_dev;
    const int32_t *row_map;
    // shader-clock probe (clock_probe_*, below): {cycles, constant-rate ticks} summed over this kernel's workgroups, or nullptr
    unsigned long long *clk;
    // Variable-degree LDS ring (bp_decode_kernel<..., LDPC_RING_VAR>): the check rows and the bit-column pairs in the order the
    // wavefronts take them (entry w + step * wavefronts), four ints each -- row: {first edge, row, weight, 0}; pair of columns
    // (2g, 2g + 1): {first position in csc_edge, g, weight of column 2g, weight of column 2g + 1 or -1 if there is none}.
    // ring_units: 1 KiB units of LDS per wavefront.
    const int32_t *row_items, *pair_items;
    int32_t ring_units;
};

// What a 64-syndrome tile needs besides its message arrays to continue in the per-pass kernels.  Those run in
// ROUNDS (one BP iteration each) of four launches; within a launch many workgroups read a tile's state while one
// of them advances it, so everything that changes is double-buffered by round parity or written once.
struct TileState {
    uint64_t done[2];            // [round & 1]: lanes whose syndrome has converged (or that lie beyond the batch)
    unsigned long long unsat[2]; // [round & 1]: OR over rows of (candidate parity ^ syndrome), filled by the syndrome pass
    int32_t it0;                 // iterations completed before round 0
    int32_t end_round;           // round in which the tile's outputs became final (INT32_MAX while it runs)
    int32_t lane_iter[64];       // iteration at which each lane converged
    int32_t llr_each[2];         // [round & 1]: the bit pass stores the live lanes' posteriors every iteration (set at the tile's first convergence
                                 // event, or from the start where convergence is imminent: the second pass of a compacted decode)
};

__device__ __forceinline__ uint64_t sm64(uint64_t seed, uint64_t idx) {  // twin of ldpc_amd/prng.py
    uint64_t z = seed + (idx + 1ull) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Read-only tables (CSR/CSC indices, priors, packed syndrome masks: all written by EARLIER launches) are
// read through the constant address space: with a wave-uniform address that is an s_load on the scalar
// cache, tracked by lgkmcnt.  As plain global loads they would be vector-memory operations whose
// `s_waitcnt vmcnt(0)` also drains the asynchronous message prefetches queued behind them.
template <class T>
__device__ __forceinline__ T sload(const T *p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) T *>(reinterpret_cast<uintptr_t>(p));
}

__device__ __forceinline__ uint64_t uniform64(uint64_t v) {  // wave-uniform value -> SGPR pair
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

__device__ __forceinline__ uint64_t wave_or(uint64_t v) {
    uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        lo |= __shfl_xor(lo, off, LDPC_WAVE);
        hi |= __shfl_xor(hi, off, LDPC_WAVE);
    }
    return ((uint64_t)hi << 32) | lo;
}

// Message arrays are reached through buffer descriptors: "SGPR descriptor + SGPR edge offset + VGPR
// lane offset", so an access costs no per-lane 64-bit address arithmetic and no address VGPR pairs
// (flat global_load needs a VGPR pair per distinct address; with ~30 addresses live that alone cost
// an occupancy step).  One descriptor covers one tile's [nnz][64] doubles: nnz * 512 bytes < 4 GiB.
// Cache policy of the message traffic (the `aux` operand of the buffer instructions: 0 = default, 2 = nt, non-temporal).
// A large batch's messages are touched once per pass and come round again only after gigabytes of other tiles' traffic,
// so no cache level can hold them; nt tells L2 / MALL not to try (measured on the headline workload: +2.4 % at 1024 tiles,
// +3.8 % at 64).  A handful of tiles (<= ~250 MB of messages) DO live in the 256 MB MALL from one pass to the next and are
// better off with the default policy (8 tiles: nt -3 %), so the policy is a template parameter of the buffer type.
typedef unsigned int ldpc_v2u __attribute__((ext_vector_type(2)));
template <int AUX>
struct MsgBufT {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ double ld(int lane8, int edge) const {  // edge is wave-uniform
        ldpc_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane8, (int)((unsigned)edge << 9), AUX);
        return __builtin_bit_cast(double, v);
    }
    __device__ __forceinline__ void st(int lane8, int edge, double x) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ldpc_v2u, x), rsrc, lane8,
                                              (int)((unsigned)edge << 9), AUX);
    }
};
typedef MsgBufT<0> MsgBuf;    // default policy
typedef MsgBufT<2> MsgBufNT;  // non-temporal: streamed tiles
template <class BUF = MsgBuf>
__device__ __forceinline__ BUF make_msgbuf(double *base, unsigned rows) {
    BUF b;
    b.rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)(rows << 9), 0x00020000);
    return b;
}

// check -> bit, product-sum, one lane: message_sign * log((1 + x) / (1 - x))  (bp.hpp:211-216),
// x = (exclusive prefix product) * (exclusive suffix product) of the tanh values of the row
template <int MATH>
__device__ __forceinline__ double ps_message(double x, bool negate, const double *log_tab) {
    const double c = MATH == 0 ? ldpc_math::ps_log_ratio_libm(x, log_tab) : ldpc_math::ps_log_ratio(x);
    return negate ? -c : c;
}

// tanh(b / 2) of bp.hpp:208,217
template <int MATH>
__device__ __forceinline__ double ps_tanh_half(double b) {
    return MATH == 0 ? ldpc_math::tanh_half_libm(b) : ldpc_math::tanh_half(b);
}

// Keeps the scheduler from interleaving the (independent) per-edge transcendental chains: each chain
// needs ~20 VGPRs of temporaries and interleaving 6-8 of them costs occupancy for no gain -- latency is
// hidden by the other wavefronts of the SIMD, not by ILP inside one.
#define LDPC_EDGE_FENCE() __builtin_amdgcn_sched_barrier(0)

// What array A holds per edge: product-sum stores tanh(b2c / 2) (the only form the check update
// reads; evaluating it in the BIT pass puts half of the transcendental work next to each of the two
// memory passes), min-sum stores b2c itself.  Same value either way: one tanh per edge per iteration
// of the same argument the reference uses (it evaluates it twice, bp.hpp:208 and :217).
template <int METHOD, int MATH>
__device__ __forceinline__ double edge_form(double b2c) {
    return METHOD == LDPC_HIP_PRODUCT_SUM ? ps_tanh_half<MATH>(b2c) : b2c;
}

// ---- per-node arithmetic, shared by the register-prefetch and the LDS-ring variants ----------------

// One check row held in registers: cur[0..d) are the row's A values in ascending column order.
// Computes the d check->bit messages (bp.hpp:201-219 / 220-273) and stores them to C[rs + k].
// Returns the number of message stores it issued (the variable-degree ring's counted wait adds exactly that to its count of
// vector-memory operations: a store skipped here and not there would end a wait early, bp_stream_kernel.h).
template <int METHOD, int MATH, int DR, class BUF>
__device__ __forceinline__ int check_row(const double (&cur)[DR], int d, int rs, bool neg, int parity0,
                                          double alpha, const BUF &Ct, int l8, const double *log_tab) {
    double pre[DR];
    int stores = 0;
    if (METHOD == LDPC_HIP_PRODUCT_SUM) {
        double temp = 1.0;
#pragma unroll
        for (int k = 0; k < DR; ++k)
            if (k < d) { pre[k] = temp; temp *= cur[k]; }
        temp = 1.0;
#pragma unroll
        for (int k = DR - 1; k >= 0; --k)
            if (k < d) {
                Ct.st(l8, rs + k, ps_message<MATH>(pre[k] * temp, neg, log_tab));
                ++stores;
                temp *= cur[k];
                LDPC_EDGE_FENCE();
            }
    } else {
        // total_sgn = syndrome[i] + #{b2c <= 0}; only its parity is used (bp.hpp:236-262)
        int parity = parity0;
        double temp = DBL_MAX;
#pragma unroll
        for (int k = 0; k < DR; ++k)
            if (k < d) {
                if (cur[k] <= 0) parity ^= 1;
                pre[k] = temp;
                const double ab = fabs(cur[k]);
                if (ab < temp) temp = ab;
            }
        temp = DBL_MAX;
#pragma unroll
        for (int k = DR - 1; k >= 0; --k)
            if (k < d) {
                const int sgn = parity ^ (cur[k] <= 0 ? 1 : 0);
                double mag = pre[k];
                if (temp < mag) mag = temp;
                const double signed_alpha = sgn ? -alpha : alpha;  // message_sign * alpha
                Ct.st(l8, rs + k, mag * signed_alpha);
                ++stores;
                const double ab = fabs(cur[k]);
                if (ab < temp) temp = ab;
            }
    }
    return stores;
}

// ---- product-sum check row, libm-exact math: the fast path -------------------------------------------------------
// The exact `log` has two evaluation branches (argument within ~6 % of 1, or not), and in a wavefront of 64 syndromes
// both are almost always populated -- at the benchmark's operating point 5.7 % of the arguments are near 1, i.e. ~22 of
// a row's 384 (lane, entry) pairs, yet every entry paid for both branches (42 + 37 VALU instructions).  Here
//   * every lane evaluates the TABLE branch for each of its entries and keeps the result in a register (the one that
//     held the entry's prefix product, which is dead by then),
//   * lanes whose argument is near 1 park it in a wave-private LDS buffer, COMPACTED over the whole row (position =
//     entries parked before + rank of the lane among this entry's parkers: one ballot, one mbcnt),
//   * the near-1 branch then runs once per 64 parked arguments (usually once per row instead of six times),
//   * the owners fetch their results, and the row's messages are stored -- once each.
// The same operations reach the same operands, so every message keeps its bits.  The row-level preconditions (no NaN
// and no +-1 among the row's tanh values in any LIVE lane) remove the per-entry 0 / inf / NaN patches of the generic
// routine: then q = (1 + x) / (1 - x) is a normal number.  Lanes whose syndrome has converged keep computing (the
// wavefront executes them anyway) but their values are dead, so they neither veto the fast path nor park anything;
// a row that fails the precondition takes the generic path below.
#define LDPC_NEAR_SLOTS 48  // parked arguments per wavefront (LDS doubles); a row with more evaluates the excess in place
#define LDPC_NEAR_BYTES (LDPC_NEAR_SLOTS * 8)

__device__ __forceinline__ int lane_rank(uint64_t mask) {  // number of set bits of `mask` below this lane
    return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

// Returns the number of message stores issued, or -1 when the row does not qualify (nothing stored: the caller takes the generic row).
template <int DR, class BUF>
__device__ __forceinline__ int check_row_ps_exact_fast(const double (&cur)[DR], int d, int rs, bool neg, const BUF &Ct, int l8,
                                                        const double *log_tab, uint64_t live, double *near_buf) {
    if (d < 2) return -1;  // a weight-1 row: x is the empty product 1.0, q = 2 / 0 (generic path: +inf)
    bool bad = false;
#pragma unroll
    for (int k = 0; k < DR; ++k)
        if (k < d) bad = bad || !(__builtin_fabs(cur[k]) < 1.0);
    if (__builtin_amdgcn_ballot_w64(bad) & live) return -1;
    const int lane = (int)(threadIdx.x & (LDPC_WAVE - 1));
    const bool lane_live = (live >> lane) & 1ull;
    const uint64_t sign = neg ? 0x8000000000000000ull : 0ull;  // message_sign (bp.hpp:213) as a sign-bit flip
    double pre[DR];
    double temp = 1.0;
#pragma unroll
    for (int k = 0; k < DR; ++k)
        if (k < d) { pre[k] = temp; temp *= cur[k]; }
    temp = 1.0;
    uint64_t parked[DR];
    int first[DR];
    double out[DR];
    int total = 0;
#pragma unroll
    for (int k = DR - 1; k >= 0; --k) {
        parked[k] = 0;
        first[k] = 0;
        if (k < d) {
            const double x = pre[k] * temp;
            temp *= cur[k];
            const double q = ldpc_math::div_cr(1.0 + x, 1.0 - x);
            const bool near_any = ldpc_math::log_near_one(q);
            double y = ldpc_math::log_libm_general(q, log_tab);
            const uint64_t mask = __builtin_amdgcn_ballot_w64(near_any) & live;  // (scalar AND: dead lanes park nothing)
            const bool near = near_any && lane_live;
            if (mask) {
                const int cnt = __builtin_popcountll(mask);
                if (total + cnt <= LDPC_NEAR_SLOTS) {
                    if (near) near_buf[total + lane_rank(mask)] = q;
                    parked[k] = mask;
                    first[k] = total;
                    total += cnt;
                } else if (near) {
                    y = ldpc_math::log_libm_near_one(q);  // buffer full: in place, as the generic routine would
                }
            }
            out[k] = y;
            LDPC_EDGE_FENCE();
        }
    }
    if (total) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS operations of a wavefront execute in order; this only pins the compiler
        for (int c = lane; c < total; c += LDPC_WAVE) near_buf[c] = ldpc_math::log_libm_near_one(near_buf[c]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = DR - 1; k >= 0; --k)
            if (k < d && parked[k]) {
                if ((parked[k] >> lane) & 1ull) out[k] = near_buf[first[k] + lane_rank(parked[k])];
            }
    }
    int stores = 0;
#pragma unroll
    for (int k = 0; k < DR; ++k)
        if (k < d) {
            Ct.st(l8, rs + k, ldpc_math::as_f64(ldpc_math::as_u64(out[k]) ^ sign));
            ++stores;
        }
    return stores;
}

// product-sum rows of the streaming kernels: the fast path where it applies, else the generic row; returns the message stores issued
template <int METHOD, int MATH, int DR, class BUF>
__device__ __forceinline__ int check_row_live(const double (&cur)[DR], int d, int rs, bool neg, int parity0, double alpha,
                                               const BUF &Ct, int l8, const double *log_tab, uint64_t live, double *near_buf) {
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0) {
        const int fast = check_row_ps_exact_fast<DR, BUF>(cur, d, rs, neg, Ct, l8, log_tab, live, near_buf);
        if (fast >= 0) return fast;
    }
    return check_row<METHOD, MATH, DR, BUF>(cur, d, rs, neg, parity0, alpha, Ct, l8, log_tab);
}

// A row heavier than the register bound: two streaming sweeps, exactly the reference's loops.
template <int METHOD, int MATH, class BUF>
__device__ __forceinline__ void check_row_streamed(int d, int rs, bool neg, int parity, double alpha,
                                                   const BUF &At, const BUF &Ct, int l8,
                                                   const double *log_tab) {
    if (METHOD == LDPC_HIP_PRODUCT_SUM) {
        double temp = 1.0;
        for (int k = 0; k < d; ++k) {
            Ct.st(l8, rs + k, temp);
            temp *= At.ld(l8, rs + k);
        }
        temp = 1.0;
        for (int k = d - 1; k >= 0; --k) {
            Ct.st(l8, rs + k, ps_message<MATH>(Ct.ld(l8, rs + k) * temp, neg, log_tab));
            temp *= At.ld(l8, rs + k);
        }
    } else {
        double temp = DBL_MAX;
        for (int k = 0; k < d; ++k) {
            const double bk = At.ld(l8, rs + k);
            if (bk <= 0) parity ^= 1;
            Ct.st(l8, rs + k, temp);
            const double ab = fabs(bk);
            if (ab < temp) temp = ab;
        }
        temp = DBL_MAX;
        for (int k = d - 1; k >= 0; --k) {
            const double bk = At.ld(l8, rs + k);
            const int sgn = parity ^ (bk <= 0 ? 1 : 0);
            double mag = Ct.ld(l8, rs + k);
            if (temp < mag) mag = temp;
            const double signed_alpha = sgn ? -alpha : alpha;
            Ct.st(l8, rs + k, mag * signed_alpha);
            const double ab = fabs(bk);
            if (ab < temp) temp = ab;
        }
    }
}

// One bit column held in registers: c[0..d) are its check->bit messages in ascending row order, e[] the
// CSR edge ids.  Posterior (bp.hpp:276-287) returned; bit->check messages (bp.hpp:279 + 311-318) stored; *stores (if asked
// for) = the number of message stores issued (see check_row).
template <int METHOD, int MATH, int DC, class BUF>
__device__ __forceinline__ double bit_column(const double (&c)[DC], const int (&e)[DC], int d, double prior,
                                             const BUF &At, int l8, bool messages = true, int *stores = nullptr) {
    double pre[DC];
    if (stores) *stores = 0;
    double temp = prior;
#pragma unroll
    for (int k = 0; k < DC; ++k)
        if (k < d) { pre[k] = temp; temp += c[k]; }
    const double llr = temp;
    if (!messages) return llr;  // (wave-uniform) the decode's last bit pass: the messages it would send are never read
    double s = 0.0;
#pragma unroll
    for (int k = DC - 1; k >= 0; --k)
        if (k < d) {
            At.st(l8, e[k], edge_form<METHOD, MATH>(pre[k] + s));
            if (stores) ++*stores;
            s += c[k];
            if (METHOD == LDPC_HIP_PRODUCT_SUM) LDPC_EDGE_FENCE();
        }
    return llr;
}

// ---- LDS-DMA ring ------------------------------------------------------------------------------------
// A wavefront keeps RING_DEPTH rows (check pass) or bit pairs (bit pass) of message data in flight into
// its private LDS ring with `buffer_load_dwordx4 ... lds`: 64 lanes x 16 B = two whole 512-byte edge
// segments per instruction, no VGPRs held while the data is in flight.  hipcc does not count these loads,
// so the waits are explicit: vector-memory operations complete in issue order, hence "at most N
// operations outstanding", with N = the number of operations issued AFTER the wanted load, proves it has
// landed.  N must be a lower bound of that number (a smaller N only waits longer); the steady-state
// constants below assume exactly-regular node degrees, which is why the ring variant is only selected
// for such matrices (host side: rows all of weight DR, columns all of weight DC).
extern __shared__ __attribute__((aligned(16))) unsigned char ldpc_dyn_lds[];

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, unsigned lds_addr) {
    unsigned keep;  // M0 carries the LDS destination; it is compiler-reserved, so save/restore it in the same statement
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N < 63 ? N : 63) : "memory"); }
__device__ __forceinline__ void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// the same wait with a count known only at run time (wave-uniform): `s_waitcnt` takes an immediate, so one of 64 instructions is
// picked by a scalar branch.  As above a count that is too small only waits longer; 63 is the most the counter can express.
#define LDPC_WAIT_CASE(k) case (k): wait_vmcnt<(k)>(); break;
#define LDPC_WAIT_CASE8(b) LDPC_WAIT_CASE((b) + 0) LDPC_WAIT_CASE((b) + 1) LDPC_WAIT_CASE((b) + 2) LDPC_WAIT_CASE((b) + 3) \
                           LDPC_WAIT_CASE((b) + 4) LDPC_WAIT_CASE((b) + 5) LDPC_WAIT_CASE((b) + 6) LDPC_WAIT_CASE((b) + 7)
__device__ __forceinline__ void wait_vmcnt_dyn(int n) {
    switch (__builtin_amdgcn_readfirstlane(n < 0 ? 0 : n)) {  // (a count can not be negative; if it ever were, wait for everything)
        LDPC_WAIT_CASE8(0) LDPC_WAIT_CASE8(8) LDPC_WAIT_CASE8(16) LDPC_WAIT_CASE8(24) LDPC_WAIT_CASE8(32) LDPC_WAIT_CASE8(40) LDPC_WAIT_CASE8(48)
        LDPC_WAIT_CASE(56) LDPC_WAIT_CASE(57) LDPC_WAIT_CASE(58) LDPC_WAIT_CASE(59) LDPC_WAIT_CASE(60) LDPC_WAIT_CASE(61) LDPC_WAIT_CASE(62)
        default: wait_vmcnt<63>(); break;
    }
}

// ---- kernel arguments read on demand ---------------------------------------------------------------------------------------------
// A kernel whose inner loop fills the scalar register file (bp_edge_kernel: 56 lane masks) cannot also keep a dozen pointers it
// needs once per syndrome: the compiler loads every argument at entry, parks the block in VGPR lanes and fetches it back
// (v_readlane, a vector instruction each) wherever a field is used -- 16 per output round.  Such "cold" fields are instead read
// from the kernel-argument segment with a scalar load at the point of use: LDPC_KERNARG(ARGS, field) -- no vector instruction,
// nothing live across the loop.  (A field read this way must not ALSO be read as a.field, or the entry load comes back.)
template <typename T, int OFF>
__device__ __forceinline__ T kernarg_load() {
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "one or two dwords");
    const auto kp = __builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (sizeof(T) == 8) {
        unsigned long long v;
        asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(kp), "n"(OFF) : "memory");
        T out;
        __builtin_memcpy(&out, &v, 8);
        return out;
    } else {
        unsigned v;
        asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(v) : "s"(kp), "n"(OFF) : "memory");
        T out;
        __builtin_memcpy(&out, &v, 4);
        return out;
    }
}
#define LDPC_KERNARG(ARGS, field) kernarg_load<decltype(ARGS::field), (int)offsetof(ARGS, field)>()
// (a pointer fetched this way is a plain number to the compiler: say that it points to global memory, or every access becomes a flat_ one)
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T *global_ptr(T *p) { return (__attribute__((address_space(1))) T *)p; }

// ---- the shader clock a kernel actually ran at --------------------------------------------------------------------------------------
// The chip does not hold one clock under load (1.7 GHz on the FP64-heavy streamed kernel, 2.1 - 2.3 GHz on the on-chip ones, and it
// differs from box to box), so any "fraction of the issue slots" needs the clock of THE RUN.  Every workgroup of the long-running BP
// kernels reads the shader-cycle counter (s_memtime) and the constant-rate counter (s_memrealtime) when it starts and when it ends and
// adds both differences to two 64-bit words of the handle: sum(cycles) / sum(ticks) x the tick rate = the clock, averaged over
// the kernels' lifetime and weighted by it.  Two scalar reads and two atomics per workgroup LIFETIME (the kernels are persistent).
// The words only ever grow; ldpc_hip_bp_clock_probe reads them, and the caller takes differences around what it times.
__device__ __forceinline__ void clock_probe_begin(unsigned long long *stamp) {  // ONE thread of the workgroup; stamp: two words (LDS)
    stamp[0] = (unsigned long long)__builtin_readcyclecounter();
    stamp[1] = (unsigned long long)__builtin_readsteadycounter();
}
__device__ __forceinline__ void clock_probe_end(unsigned long long *clk, const unsigned long long *stamp) {  // the same thread, once
    if (!clk) return;
    const unsigned long long c = (unsigned long long)__builtin_readcyclecounter() - stamp[0];
    const unsigned long long t = (unsigned long long)__builtin_readsteadycounter() - stamp[1];
    __hip_atomic_fetch_add(clk, c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(clk + 1, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- work distribution of the one-syndrome-per-wavefront kernels ----------------------------------------------------------------
// A wavefront's syndromes take 3 .. 100 us and the wavefronts do not run at one speed (equal static shares finish 20 % apart on
// BASELINE config 3: tools/sweep_edge_split.py), so the batch is handed out syndrome by syndrome.  ONE counter word serves ~88
// atomic visits per microsecond device-wide -- 5 120 resident wavefronts outrun that on every code below d = 21 -- hence
// WORK_POOLS counters, each in its own 4 KiB of memory (another channel), each owning a contiguous slice of the batch.  A
// wavefront starts with its static share (if any: no counter), then draws from pool blockIdx mod WORK_POOLS (workgroups go round
// the 8 XCDs: a pool is served by one XCD) and, when that runs dry, reads all counters in one vector load and moves to the next
// pool that still has work: the faster XCDs finish the slower ones' slices.
constexpr int WORK_POOLS = 32;
constexpr int WORK_POOL_STRIDE = 512;  // 64-bit words between counters
__host__ __device__ inline size_t work_pool_bytes() { return (size_t)WORK_POOLS * WORK_POOL_STRIDE * 8; }
__host__ __device__ inline int32_t work_pool_share(int64_t batch, int64_t dyn_base) { return (int32_t)((batch - dyn_base + WORK_POOLS - 1) / WORK_POOLS); }

// The next syndromes [b0, b1) of this wavefront (q: its current pool); false: the batch is done.  Called by a whole wavefront;
// wave-uniform results.  batch < 2^30.
__device__ __forceinline__ bool work_pool_next(unsigned long long *next_generic, int dyn_base, int pool_per, int chunk, int batch, int lane, int &q, int &b0, int &b1) {
    // (the wavefront must be whole when lane 0 pulls: without a convergent operation between a caller's `lane == 0` block and the one
    // below, the compiler threads the two and the readfirstlane runs with lane 0 masked off -- seen in bp_edge_kernel<1>)
    __builtin_amdgcn_wave_barrier();
    const auto next = global_ptr(next_generic);
    for (;;) {
        unsigned pulled = 0;
        if (lane == 0) pulled = (unsigned)__hip_atomic_fetch_add(next + (size_t)q * WORK_POOL_STRIDE, (unsigned long long)chunk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int got = __builtin_amdgcn_readfirstlane((int)pulled);
        const int lo = dyn_base + q * pool_per;
        const int end = lo + pool_per < batch ? lo + pool_per : batch;
        if (got < end - lo) {
            b0 = lo + got;
            b1 = b0 + chunk < end ? b0 + chunk : end;
            return true;
        }
        // this pool is dry: which ones are not?  (one load per lane; a counter only grows, so a pool seen dry stays dry)
        bool has = false;
        if (lane < WORK_POOLS) {
            const unsigned long long c = __hip_atomic_load(next + (size_t)lane * WORK_POOL_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const long long plo = (long long)dyn_base + (long long)lane * pool_per;
            const long long pend = plo + pool_per < batch ? plo + pool_per : batch;
            has = plo + (long long)c < pend;
        }
        const uint64_t mask = __ballot(has);
        if (!mask) return false;
        const uint64_t at_or_after = mask & (~0ull << q);
        q = at_or_after ? __builtin_ctzll(at_or_after) : __builtin_ctzll(mask);
    }
}
