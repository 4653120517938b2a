// bp_hip.hip -- libldpc_hip.so: batched flooding belief propagation for gfx950 (MI355X, CDNA4).
//
// What it replaces (reference = quantumgizmos/ldpc @ /root/reference):
//   ldpc::bp::BpDecoder::bp_decode_parallel          src_cpp/bp.hpp:192-325
//   ldpc::bp::BpDecoder::initialise_log_domain_bp    src_cpp/bp.hpp:147-157
//   ldpc::gf2sparse::GF2Sparse::mulvec               src_cpp/gf2sparse.hpp:177-214
// for a BATCH of independent syndromes.  This is not a translation of that code: the reference walks
// a doubly linked list per syndrome on one CPU thread; here
//
//   * the unit of data parallelism is the syndrome.  64 syndromes form a TILE; lane l of every
//     wavefront owns syndrome l of its tile, so each per-edge message access of a wavefront is one
//     fully coalesced 512-byte row  msg[tile][edge][0..63]  (batch-minor layout),
//   * one workgroup owns one tile for the WHOLE decode (all iterations): its wavefronts stride over
//     the checks (check pass) and then over the bits (bit pass) of that tile, separated by
//     workgroup barriers only -- tiles never talk to each other, so there is no grid-wide sync,
//     no atomics and one kernel launch per decode,
//   * every lane walks its node's <= DR (row) / <= DC (column) edges SEQUENTIALLY in ascending
//     column / row order, i.e. in the reference's linked-list order (sparse_matrix_base.hpp:423-482
//     keeps both lists sorted), so every floating-point operation is performed in the reference's
//     association order -- min-sum is bit-identical, product-sum differs only by libm vs ocml ulps,
//   * hard decisions are kept bit-packed per (tile, bit) as the wavefront's ballot; the syndrome
//     test of bp.hpp:300-302 is an XOR-gather of those 64-bit words per check,
//   * a syndrome that converges freezes its outputs (bp.hpp:300-308 early return); a tile whose 64
//     syndromes are all done retires its workgroup.
//
// No MFMA: the path is a sparse gather/scatter bound by HBM bandwidth (4 * nnz * 8 bytes per
// syndrome-iteration) and by FP64 transcendentals.
//
// Built with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared

#include <hip/hip_runtime.h>

#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ldpc_hip.h"
#include "bp_math.h"

// math modes of the product-sum kernel (ldpc_hip_bp_set_math): 0 = bit-identical twins of the host
// glibc the reference runs on (default), 1 = fast ~1-ulp routines (bp_math.h)

#define LDPC_WAVE 64  // gfx950 wavefront; also the tile width (syndromes per workgroup)

// ------------------------------------------------------------------------------------------------
// device side
// ------------------------------------------------------------------------------------------------

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
    unsigned *counters;                 // [0] tiles finished by the persistent kernel, [1] tiles handed off
    int32_t *handoff_list;              // [tiles] ids of handed-off tiles
    int32_t total_tiles, handoff_threshold;
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
typedef unsigned int ldpc_v2u __attribute__((ext_vector_type(2)));
struct MsgBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ double ld(int lane8, int edge) const {  // edge is wave-uniform
        ldpc_v2u v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, lane8, (int)((unsigned)edge << 9), 0);
        return __builtin_bit_cast(double, v);
    }
    __device__ __forceinline__ void st(int lane8, int edge, double x) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ldpc_v2u, x), rsrc, lane8,
                                              (int)((unsigned)edge << 9), 0);
    }
};
__device__ __forceinline__ MsgBuf make_msgbuf(double *base, unsigned rows) {
    MsgBuf b;
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
template <int METHOD, int MATH, int DR>
__device__ __forceinline__ void check_row(const double (&cur)[DR], int d, int rs, bool neg, int parity0,
                                          double alpha, const MsgBuf &Ct, int l8, const double *log_tab) {
    double pre[DR];
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
                const double ab = fabs(cur[k]);
                if (ab < temp) temp = ab;
            }
    }
}

// A row heavier than the register bound: two streaming sweeps, exactly the reference's loops.
template <int METHOD, int MATH>
__device__ __forceinline__ void check_row_streamed(int d, int rs, bool neg, int parity, double alpha,
                                                   const MsgBuf &At, const MsgBuf &Ct, int l8,
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
// CSR edge ids.  Posterior (bp.hpp:276-287) returned; bit->check messages (bp.hpp:279 + 311-318) stored.
template <int METHOD, int MATH, int DC>
__device__ __forceinline__ double bit_column(const double (&c)[DC], const int (&e)[DC], int d, double prior,
                                             const MsgBuf &At, int l8) {
    double pre[DC];
    double temp = prior;
#pragma unroll
    for (int k = 0; k < DC; ++k)
        if (k < d) { pre[k] = temp; temp += c[k]; }
    const double llr = temp;
    double s = 0.0;
#pragma unroll
    for (int k = DC - 1; k >= 0; --k)
        if (k < d) {
            At.st(l8, e[k], edge_form<METHOD, MATH>(pre[k] + s));
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
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_addr), "s"(soff) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N < 63 ? N : 63) : "memory"); }
__device__ __forceinline__ void wait_lds_reads() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int METHOD, int MATH, int DR, int DC, int RING>
__global__ void __launch_bounds__(1024) bp_decode_kernel(const BpArgs a) {
    constexpr int UB = DC <= 4 ? 4 : (DC <= 8 ? 2 : 1);  // bits in flight per wavefront (register variant)
    const int lane = threadIdx.x & (LDPC_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;

    const int32_t *__restrict__ row_ptr = a.row_ptr;
    const int32_t *__restrict__ col_idx = a.col_idx;
    const int32_t *__restrict__ col_ptr = a.col_ptr;
    const int32_t *__restrict__ csc_edge = a.csc_edge;
    const double *__restrict__ llr0 = a.llr0;
    const uint64_t *__restrict__ par = a.par + tile * m;
    const uint64_t *__restrict__ nzm = a.nzm + tile * m;

    const MsgBuf At = make_msgbuf(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf Ct = make_msgbuf(a.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    uint64_t *dec = a.dec + tile * n;    // frozen decisions of converged syndromes (zero-initialised)
    uint64_t *dcur = a.dcur + tile * n;  // this iteration's hard decisions, all lanes
    const bool want_llr = a.llr_t != nullptr;
    const MsgBuf Lt = make_msgbuf(want_llr ? a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.A, want_llr ? (unsigned)n : 0u);
    const int l8 = lane * 8;

    __shared__ uint64_t red[2][16];
    __shared__ int red_i;
    __shared__ __attribute__((aligned(16))) double log_tab[256];  // glibc log's {1/c, log c} table, LDS-resident
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = threadIdx.x; q < 256; q += blockDim.x) log_tab[q] = ldpc_math::k_log_tab[q];

    // ring geometry (RING variant): one slot holds a check row (DR segments) or a pair of bit columns (2*DC)
    constexpr int ROW_DMAS = (DR + 1) / 2;                       // 1 KiB DMA instructions per row
    constexpr int SLOT_BYTES = (ROW_DMAS > DC ? ROW_DMAS : DC) * 1024;
    constexpr int N_CHECK = RING * DR + (RING - 1) * ROW_DMAS;
    constexpr int N_BIT = RING * (2 * DC + 2) + (RING - 1) * DC;
    const unsigned ring_addr = (unsigned)(uintptr_t)ldpc_dyn_lds + (unsigned)wave * (RING * SLOT_BYTES);
    const double *ringp = reinterpret_cast<const double *>(ldpc_dyn_lds + (size_t)wave * (RING * SLOT_BYTES));
    const unsigned l16 = (unsigned)lane * 16u;

    // lanes beyond the batch (partial last tile) are born "done"
    const int64_t valid = a.batch - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    const uint64_t never = a.invalid[tile];
    int my_iter = 0;  // meaningful in wave 0: iteration at which this lane's syndrome converged

    // initialise_log_domain_bp (bp.hpp:147-157): every edge of column j starts at llr0[j]
    for (int e = wave; e < nnz; e += nwaves) At.st(l8, e, edge_form<METHOD, MATH>(sload(llr0 + sload(col_idx + e))));
    __syncthreads();

    for (int it = 1; it <= a.max_iter; ++it) {
        // ---------------- check pass (bp.hpp:201-273) ----------------
        double alpha = 0.0;
        if (METHOD == LDPC_HIP_MINIMUM_SUM)
            alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;

        if (RING) {
            // every row has exactly DR entries: row i starts at edge i * DR
            const int nsteps = wave < m ? (m - wave + nwaves - 1) / nwaves : 0;
            auto issue_row = [&](int i, int slot) {
#pragma unroll
                for (int c = 0; c < ROW_DMAS; ++c)
                    lds_dma16(At.rsrc, l16, (unsigned)(i * DR + 2 * c) << 9, ring_addr + slot * SLOT_BYTES + c * 1024);
            };
            for (int p = 0; p < RING; ++p)
                if (p < nsteps) issue_row(wave + p * nwaves, p);
            int slot = 0;
            for (int idx = 0; idx < nsteps; ++idx) {
                const int i = wave + idx * nwaves;
                if (idx >= RING && idx + RING - 1 < nsteps) wait_vmcnt<N_CHECK>();
                else wait_vmcnt<0>();
                double cur[DR];
#pragma unroll
                for (int k = 0; k < DR; ++k) cur[k] = ringp[slot * (SLOT_BYTES / 8) + k * LDPC_WAVE + lane];
                wait_lds_reads();  // the slot is free once its values sit in registers
                if (idx + RING < nsteps) issue_row(i + RING * nwaves, slot);
                const bool neg = (sload(nzm + i) >> lane) & 1ull;         // syndrome[i] != 0 (bp.hpp:213)
                const int parity = (int)((sload(par + i) >> lane) & 1ull);
                check_row<METHOD, MATH, DR>(cur, DR, i * DR, neg, parity, alpha, Ct, l8, log_tab);
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
        } else {
            // The row's inputs are fetched one row ahead (register double buffer): while the wavefront
            // works on row i its loads for row i + nwaves are already in flight.
            double cur[DR];
            int rs = 0, d = 0;
            if (wave < m) {
                rs = sload(row_ptr + wave);
                d = sload(row_ptr + wave + 1) - rs;
                if (d <= DR) {
#pragma unroll
                    for (int k = 0; k < DR; ++k)
                        if (k < d) cur[k] = At.ld(l8, rs + k);
                }
            }
            for (int i = wave; i < m; i += nwaves) {
                const int inext = i + nwaves;
                double nxt[DR];
                int rs_n = 0, d_n = 0;
                if (inext < m) {
                    rs_n = sload(row_ptr + inext);
                    d_n = sload(row_ptr + inext + 1) - rs_n;
                    if (d_n <= DR) {
#pragma unroll
                        for (int k = 0; k < DR; ++k)
                            if (k < d_n) nxt[k] = At.ld(l8, rs_n + k);
                    }
                }
                const bool neg = (sload(nzm + i) >> lane) & 1ull;
                const int parity = (int)((sload(par + i) >> lane) & 1ull);
                if (d <= DR) check_row<METHOD, MATH, DR>(cur, d, rs, neg, parity, alpha, Ct, l8, log_tab);
                else check_row_streamed<METHOD, MATH>(d, rs, neg, parity, alpha, At, Ct, l8, log_tab);
                rs = rs_n;
                d = d_n;
#pragma unroll
                for (int k = 0; k < DR; ++k) cur[k] = nxt[k];
            }
        }
        __syncthreads();

        // ---------------- bit pass (bp.hpp:276-298 and 311-318, fused) ----------------
        const bool last = (it == a.max_iter);
        const bool lane_live = !((done >> lane) & 1ull);
        if (RING) {
            // every column has exactly DC entries; a step handles the column pair (2g, 2g + 1), whose
            // 2*DC gathered segments arrive as DC DMA instructions (lanes 0-31 one segment, 32-63 the next)
            const int ngroups = (n + 1) / 2;
            const int nsteps = wave < ngroups ? (ngroups - wave + nwaves - 1) / nwaves : 0;
            auto issue_pair = [&](int g, int slot) {
                const int base = 2 * g * DC;
#pragma unroll
                for (int c = 0; c < DC; ++c) {
                    const int q0 = base + 2 * c, q1 = base + 2 * c + 1;
                    const unsigned ea = (unsigned)sload(csc_edge + (q0 < nnz ? q0 : 0));
                    const unsigned eb = (unsigned)sload(csc_edge + (q1 < nnz ? q1 : 0));
                    const unsigned voff = ((lane < 32 ? ea : eb) << 9) + (unsigned)(lane & 31) * 16u;
                    lds_dma16(Ct.rsrc, voff, 0u, ring_addr + slot * SLOT_BYTES + c * 1024);
                }
            };
            for (int p = 0; p < RING; ++p)
                if (p < nsteps) issue_pair(wave + p * nwaves, p);
            int slot = 0;
            for (int idx = 0; idx < nsteps; ++idx) {
                const int g = wave + idx * nwaves;
                if (idx >= RING && idx + RING - 1 < nsteps) wait_vmcnt<N_BIT>();
                else wait_vmcnt<0>();
                double c[2][DC];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int k = 0; k < DC; ++k)
                        c[u][k] = ringp[slot * (SLOT_BYTES / 8) + (u * DC + k) * LDPC_WAVE + lane];
                wait_lds_reads();
                if (idx + RING < nsteps) issue_pair(g + RING * nwaves, slot);
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int j = 2 * g + u;
                    if (j < n) {
                        int e[DC];
#pragma unroll
                        for (int k = 0; k < DC; ++k) e[k] = sload(csc_edge + j * DC + k);
                        const double llr = bit_column<METHOD, MATH, DC>(c[u], e, DC, sload(llr0 + j), At, l8);
                        const uint64_t hard = __ballot(llr <= 0);  // bp.hpp:290
                        if (lane == 0) dcur[j] = hard;
                        if (last && want_llr && lane_live) Lt.st(l8, j, llr);
                    }
                }
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
        } else {
            // UB columns per wavefront step: all their message loads are issued before the first is used.
            for (int j0 = wave * UB; j0 < n; j0 += nwaves * UB) {
                int cs[UB], dg[UB], e[UB][DC];
                double c[UB][DC];
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = j0 + u;
                    cs[u] = 0;
                    dg[u] = -1;  // -1: no such column
                    if (j < n) {
                        cs[u] = sload(col_ptr + j);
                        dg[u] = sload(col_ptr + j + 1) - cs[u];
                        if (dg[u] <= DC) {
#pragma unroll
                            for (int k = 0; k < DC; ++k)
                                if (k < dg[u]) {
                                    e[u][k] = sload(csc_edge + cs[u] + k);
                                    c[u][k] = Ct.ld(l8, e[u][k]);
                                }
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < UB; ++u) {
                    const int j = j0 + u;
                    if (dg[u] < 0) continue;
                    const double prior = sload(llr0 + j);
                    double llr;
                    if (dg[u] <= DC) {
                        llr = bit_column<METHOD, MATH, DC>(c[u], e[u], dg[u], prior, At, l8);
                    } else {  // heavy column: two streaming sweeps like the reference's
                        double temp = prior;
                        for (int k = 0; k < dg[u]; ++k) {
                            const int ee = sload(csc_edge + cs[u] + k);
                            At.st(l8, ee, temp);
                            temp += Ct.ld(l8, ee);
                        }
                        llr = temp;
                        double s = 0.0;
                        for (int k = dg[u] - 1; k >= 0; --k) {
                            const int ee = sload(csc_edge + cs[u] + k);
                            At.st(l8, ee, edge_form<METHOD, MATH>(At.ld(l8, ee) + s));
                            s += Ct.ld(l8, ee);
                        }
                    }
                    const uint64_t hard = __ballot(llr <= 0);  // bp.hpp:290
                    if (lane == 0) dcur[j] = hard;
                    if (last && want_llr && lane_live) Lt.st(l8, j, llr);
                }
            }
        }
        __syncthreads();

        // ---------------- syndrome test (bp.hpp:292-294, 300-308) ----------------
        uint64_t unsat = 0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            uint64_t cand = 0;
            for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) cand ^= dcur[col_idx[e]];
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
            // these syndromes stop here (bp.hpp:300-308): freeze their decisions, and their posteriors are
            // those of THIS iteration (its check->bit messages are still intact in C)
            if ((newly >> lane) & 1ull) my_iter = it;
            const bool mine = (newly >> lane) & 1ull;
            for (int j = wave; j < n; j += nwaves) {
                if (lane == 0) dec[j] = (dec[j] & ~newly) | (dcur[j] & newly);
                if (!last && want_llr) {
                    double temp = llr0[j];
                    for (int p = col_ptr[j]; p < col_ptr[j + 1]; ++p) temp += Ct.ld(l8, csc_edge[p]);
                    if (mine) Lt.st(l8, j, temp);
                }
            }
            done |= newly;
            __syncthreads();  // C is overwritten by the next check pass
        }
        if (done == ~0ull) break;
        // Few tiles left running (stragglers, a tiny batch, or the tail of the launch): a lone tile is bound to
        // ONE compute unit (~3 ms per iteration on the n = 10 000 code), so park its state and let the per-pass
        // kernels below spread its remaining iterations over the whole chip.
        if (a.handoff_threshold > 0 && it < a.max_iter) {
            if (threadIdx.x == 0)
                red_i = a.total_tiles - (int)__hip_atomic_load(&a.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (red_i <= a.handoff_threshold) {
                TileState *stt = a.state + tile;
                if (wave == 0) stt->lane_iter[lane] = my_iter;
                if (threadIdx.x == 0) {
                    stt->done[0] = done;
                    stt->it0 = it;
                    stt->end_round = INT32_MAX;
                    stt->unsat[0] = stt->unsat[1] = 0ull;
                    a.handoff_list[atomicAdd(&a.counters[1], 1u)] = (int32_t)tile;
                }
                return;
            }
            __syncthreads();  // red_i is rewritten next iteration
        }
    }

    // syndromes that never converged report the last iteration's decisions (bp.hpp:320-322)
    if (done != ~0ull)
        for (int j = threadIdx.x; j < n; j += blockDim.x) dec[j] = (dec[j] & done) | (dcur[j] & ~done);

    if (wave == 0) {
        const int64_t b = tile * LDPC_WAVE + lane;
        if (b < a.batch) {
            const bool cv = ((done >> lane) & 1ull) != 0;
            if (a.iters) a.iters[b] = cv ? my_iter : a.max_iter;  // bp.hpp:304
            if (a.conv) a.conv[b] = cv ? 1 : 0;
        }
    }
    if (threadIdx.x == 0 && a.counters) atomicAdd(&a.counters[0], 1u);
}

// ---- per-pass kernels for handed-off tiles ---------------------------------------------------------------
// Same arithmetic, same arrays, but one launch per pass and one wavefront per NODE, so the rows / columns of a
// single tile are spread over all 256 compute units.  Used for the tiles the persistent kernel parks when the
// chip would otherwise idle; each launch handles every parked tile that is still running.
struct SpreadArgs {
    BpArgs bp;
    int32_t n_tiles;  // entries of bp.handoff_list
    int32_t nodes;    // rows / columns per wavefront (1 for a handful of tiles: latency; 4 otherwise: amortises the table load)
    int32_t round;    // 0-based per-pass round; a tile's iteration number is it0 + round + 1
};

// tile handled by workgroup row `slot`, its iteration number and converged mask in this round; false: already final
__device__ __forceinline__ bool spread_tile(const SpreadArgs &a, int slot, int64_t &tile, const TileState *&st, int &it, uint64_t &done) {
    tile = a.bp.handoff_list[slot];
    st = a.bp.state + tile;
    it = st->it0 + a.round + 1;
    done = st->done[a.round & 1];
    return a.round <= st->end_round;
}


template <int METHOD, int MATH, int DR>
__global__ void __launch_bounds__(256) bp_spread_check_kernel(const SpreadArgs a) {
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = threadIdx.x; q < 256; q += blockDim.x) log_tab[q] = ldpc_math::k_log_tab[q];
    __syncthreads();
    int64_t tile;
    const TileState *st;
    int it;
    uint64_t done;
    if (!spread_tile(a, blockIdx.y, tile, st, it, done)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nnz = a.bp.nnz, l8 = lane * 8;
    const MsgBuf At = make_msgbuf(a.bp.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf Ct = make_msgbuf(a.bp.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
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
            check_row<METHOD, MATH, DR>(cur, d, rs, neg, parity, alpha, Ct, l8, log_tab);
        } else {
            check_row_streamed<METHOD, MATH>(d, rs, neg, parity, alpha, At, Ct, l8, log_tab);
        }
    }
}

template <int METHOD, int MATH, int DC>
__global__ void __launch_bounds__(256) bp_spread_bit_kernel(const SpreadArgs a) {
    int64_t tile;
    const TileState *st;
    int it;
    uint64_t done;
    if (!spread_tile(a, blockIdx.y, tile, st, it, done)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nnz = a.bp.nnz, n = a.bp.n, l8 = lane * 8;
    const MsgBuf At = make_msgbuf(a.bp.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBuf Ct = make_msgbuf(a.bp.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const bool want_llr = a.bp.llr_t != nullptr;
    const MsgBuf Lt = make_msgbuf(want_llr ? a.bp.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.bp.A, want_llr ? (unsigned)n : 0u);
    const bool last = it == a.bp.max_iter;
    const bool lane_live = !((done >> lane) & 1ull);
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
            llr = bit_column<METHOD, MATH, DC>(c, e, d, prior, At, l8);
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
        if (last && want_llr && lane_live) Lt.st(l8, j, llr);
    }
}

// candidate syndrome vs syndrome (bp.hpp:292-294, 300-302) for the parked tiles, one thread per (tile, row); the
// per-tile verdict is OR-accumulated into TileState::unsat for bp_spread_finish_kernel
__global__ void __launch_bounds__(256) bp_spread_synd_kernel(const SpreadArgs a) {
    int64_t tile;
    const TileState *st;
    int it;
    uint64_t done;
    if (!spread_tile(a, blockIdx.y, tile, st, it, done)) return;
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
}

// batches of only a few tiles skip the persistent kernel altogether: state + message initialisation for the per-pass path
__global__ void __launch_bounds__(256) bp_spread_state_init_kernel(const SpreadArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.n_tiles) return;
    TileState *st = a.bp.state + t;
    const int64_t valid = a.bp.batch - (int64_t)t * LDPC_WAVE;
    st->done[0] = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    st->unsat[0] = st->unsat[1] = 0ull;
    st->it0 = 0;
    st->end_round = INT32_MAX;
    for (int l = 0; l < 64; ++l) st->lane_iter[l] = 0;
    a.bp.handoff_list[t] = t;
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

// convergence bookkeeping of a round (bp.hpp:296-311, 320-322): lanes whose candidate syndrome matched are frozen
// (decisions + posterior of THIS iteration), a tile whose lanes are all frozen or that reached max_iter gets its
// outputs.  64 bits per workgroup; workgroup 0 of a tile also advances its state.  Almost always there is nothing
// to freeze and every workgroup but the first leaves at once.
__global__ void __launch_bounds__(256) bp_spread_finish_kernel(const SpreadArgs a, unsigned *live_tiles) {
    int64_t tile;
    const TileState *cst;
    int it;
    uint64_t done;
    if (!spread_tile(a, blockIdx.y, tile, cst, it, done)) return;
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
            if (newly && !last && want_llr) {  // at the last iteration the bit pass has stored the posterior already
                double temp = a.bp.llr0[j];
                for (int p = a.bp.col_ptr[j]; p < a.bp.col_ptr[j + 1]; ++p) temp += Ct.ld(l8, a.bp.csc_edge[p]);
                if (mine) Lt.st(l8, j, temp);
            }
        }
    }
    if (blockIdx.x != 0) return;
    if (wave == 0) {
        if (mine) st->lane_iter[lane] = it;
        const int64_t b = tile * LDPC_WAVE + lane;
        if (over && b < a.bp.batch) {
            const bool cv = ((ndone >> lane) & 1ull) != 0;
            if (a.bp.iters) a.bp.iters[b] = cv ? (mine ? it : st->lane_iter[lane]) : a.bp.max_iter;  // bp.hpp:304
            if (a.bp.conv) a.bp.conv[b] = cv ? 1 : 0;
        }
    }
    if (threadIdx.x == 0) {
        st->done[par ^ 1] = ndone;
        st->unsat[par ^ 1] = 0ull;
        if (over) {
            st->end_round = a.round;
            atomicSub(live_tiles, 1u);
        }
    }
}

// syndromes [batch][m] u8  ->  par / nzm [tiles][m] u64, invalid [tiles] u64 (pre-zeroed)
__global__ void pack_syndromes_kernel(const uint8_t *__restrict__ synd, int64_t batch, int m,
                                      uint64_t *par, uint64_t *nzm, uint64_t *invalid) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tile = blockIdx.y;
    if (i >= m) return;
    uint64_t p = 0, z = 0, inv = 0;
    const int64_t b0 = tile * LDPC_WAVE;
    for (int l = 0; l < LDPC_WAVE; ++l) {
        const int64_t b = b0 + l;
        if (b < batch) {
            const uint8_t v = synd[b * m + i];
            p |= (uint64_t)(v & 1u) << l;
            z |= (uint64_t)(v != 0u) << l;
            inv |= (uint64_t)(v > 1u) << l;
        }
    }
    par[tile * m + i] = p;
    nzm[tile * m + i] = z;
    if (inv) atomicOr((unsigned long long *)&invalid[tile], (unsigned long long)inv);
}

// dec [tiles][n] u64 -> decoding [batch][n] u8
__global__ void unpack_decoding_kernel(const uint64_t *__restrict__ dec, int64_t batch, int n,
                                       uint8_t *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t tile = blockIdx.y;
    if (j >= n) return;
    const uint64_t v = dec[tile * n + j];
    const int64_t b0 = tile * LDPC_WAVE;
    for (int l = 0; l < LDPC_WAVE; ++l) {
        const int64_t b = b0 + l;
        if (b < batch) out[b * n + j] = (uint8_t)((v >> l) & 1ull);
    }
}

// llr_t [tiles][n][64] f64 -> llr [batch][n] f64, 64x64 tiles through LDS
__global__ void __launch_bounds__(256) transpose_llr_kernel(const double *__restrict__ llr_t,
                                                            int64_t batch, int n, double *out) {
    __shared__ double tilebuf[LDPC_WAVE][LDPC_WAVE + 1];
    const int j0 = blockIdx.x * LDPC_WAVE;
    const int64_t tile = blockIdx.y;
    const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
    for (int r = 0; r < 16; ++r) {
        const int jj = r * 4 + hi;
        if (j0 + jj < n) tilebuf[jj][lo] = llr_t[((size_t)tile * n + j0 + jj) * LDPC_WAVE + lo];
    }
    __syncthreads();
    for (int r = 0; r < 16; ++r) {
        const int l = r * 4 + hi;
        const int64_t b = tile * LDPC_WAVE + l;
        if (b < batch && j0 + lo < n) out[(size_t)b * n + j0 + lo] = tilebuf[lo][l];
    }
}

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
};

template <int METHOD, int MATH>
__global__ void __launch_bounds__(64) bp_serial_kernel(const SerialArgs a) {
    constexpr int DCS = 4, DRS = 8;  // register bounds of the fast path (column / row weight)
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
        for (int t = 0; t < n; ++t) {
            const int bit = a.order ? sload(a.order + t) : t;
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

// ---- on-chip variant for small codes (BASELINE configs 3 and 5) ------------------------------------
// When both message arrays of a syndrome fit in a few KiB (rotated surface d=21: 13 KiB, BB [[144,12,12]]:
// 7 KiB) nothing but the syndrome and the results needs to touch HBM.  A workgroup keeps SLOTS syndromes
// resident in LDS and iterates them together; work items are (slot, node) pairs so lanes stay busy when
// m or n is not a multiple of 64.  A slot whose syndrome converged (or hit max_iter) writes its outputs and
// immediately pulls the next syndrome from a device-wide counter, so the work done is proportional to the
// iterations each syndrome really needs (the streaming kernel's 64-lane tiles run until their slowest
// lane finishes).  Per node the edges are walked sequentially in the reference's order with the
// reference's two sweeps (bp.hpp:205-218, 278-281 + 313-316), so results are bit-identical to the
// streaming kernel's and to the reference's.
struct SmallArgs {
    int32_t m, n, nnz, max_iter, slots;
    double ms_scaling_factor;
    int64_t batch;
    const int32_t *row_ptr, *col_idx, *col_ptr, *csc_edge;
    const double *llr0;
    const uint8_t *synd;        // [batch][m]
    uint8_t *decoding;          // [batch][n]
    double *llr;                // [batch][n] or nullptr
    int32_t *iters;             // [batch] or nullptr
    uint8_t *conv;              // [batch] or nullptr
    unsigned long long *next;   // device-wide work counter (zeroed before launch)
};

template <int METHOD, int MATH>
__global__ void __launch_bounds__(256) bp_small_kernel(const SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int m = a.m, n = a.n, nnz = a.nnz, S = a.slots;
    // LDS carve-up: [log table 2 KiB][llr0 n][row_ptr m+1][col_idx nnz][col_ptr n+1][csc_edge nnz] then per slot
    // [A nnz f64][C nnz f64][L n f64][hard n u8][sy m u8]
    double *log_tab = reinterpret_cast<double *>(sm_lds);
    double *prior = log_tab + 256;
    int32_t *rp = reinterpret_cast<int32_t *>(prior + n);
    int32_t *ci = rp + (m + 1);
    int32_t *cp = ci + nnz;
    int32_t *ce = cp + (n + 1);
    size_t off = (size_t)(reinterpret_cast<unsigned char *>(ce + nnz) - sm_lds);
    off = (off + 15) & ~(size_t)15;
    const size_t slot_bytes = ((size_t)nnz * 16 + (size_t)n * 8 + (size_t)n + (size_t)m + 15) & ~(size_t)15;
    __shared__ long long slot_synd[16];  // syndrome index held by the slot, -1 = idle
    __shared__ int slot_iter[16];
    __shared__ int slot_unsat[16];
    __shared__ int slot_state[16];       // 0 running, 1 finished this iteration (write out + refill), 2 fresh (needs init)
    __shared__ int n_active;

    for (int q = tid; q < 256; q += T) log_tab[q] = ldpc_math::k_log_tab[q];
    for (int q = tid; q < n; q += T) prior[q] = a.llr0[q];
    for (int q = tid; q <= m; q += T) rp[q] = a.row_ptr[q];
    for (int q = tid; q < nnz; q += T) { ci[q] = a.col_idx[q]; ce[q] = a.csc_edge[q]; }
    for (int q = tid; q <= n; q += T) cp[q] = a.col_ptr[q];
    if (tid < S) {
        const unsigned long long idx = atomicAdd(a.next, 1ull);
        slot_synd[tid] = idx < (unsigned long long)a.batch ? (long long)idx : -1;
        slot_state[tid] = 2;
        slot_iter[tid] = 0;
        slot_unsat[tid] = 0;
    }
    __syncthreads();

    const float inv_m = m > 0 ? 1.0f / (float)m : 0.f, inv_n = n > 0 ? 1.0f / (float)n : 0.f, inv_e = nnz > 0 ? 1.0f / (float)nnz : 0.f;
    auto slot_base = [&](int s) { return sm_lds + off + (size_t)s * slot_bytes; };
    auto split = [](int w, int len, float inv, int &s, int &r) {  // w = s * len + r, exact for w < 2^22
        s = (int)(((float)w + 0.5f) * inv);
        r = w - s * len;
        if (r < 0) { --s; r += len; }
        if (r >= len) { ++s; r -= len; }
    };

    for (;;) {
        // ---- (re)initialise fresh slots: initialise_log_domain_bp (bp.hpp:147-157) + syndrome bytes ----
        for (int w = tid; w < S * nnz; w += T) {
            int s, e;
            split(w, nnz, inv_e, s, e);
            if (slot_state[s] == 2 && slot_synd[s] >= 0)
                reinterpret_cast<double *>(slot_base(s))[e] = edge_form<METHOD, MATH>(prior[ci[e]]);
        }
        for (int w = tid; w < S * m; w += T) {
            int s, i;
            split(w, m, inv_m, s, i);
            if (slot_state[s] == 2 && slot_synd[s] >= 0)
                (slot_base(s) + (size_t)nnz * 16 + (size_t)n * 9)[i] = a.synd[slot_synd[s] * m + i];
        }
        __syncthreads();
        if (tid < S && slot_state[tid] == 2) slot_state[tid] = 0;
        if (tid == 0) {
            int act = 0;
            for (int s = 0; s < S; ++s) act += slot_synd[s] >= 0;
            n_active = act;
        }
        __syncthreads();
        if (n_active == 0) break;

        // ---- check pass (bp.hpp:201-273): item = (slot, check) ----
        for (int w = tid; w < S * m; w += T) {
            int s, i;
            split(w, m, inv_m, s, i);
            if (slot_synd[s] < 0) continue;
            double *A = reinterpret_cast<double *>(slot_base(s));
            double *Cm = A + nnz;
            const uint8_t sb = (slot_base(s) + (size_t)nnz * 16 + (size_t)n * 9)[i];
            const int lo = rp[i], hi = rp[i + 1];
            if (METHOD == LDPC_HIP_PRODUCT_SUM) {
                const bool neg = sb != 0;
                double temp = 1.0;
                for (int e = lo; e < hi; ++e) { Cm[e] = temp; temp *= A[e]; }
                temp = 1.0;
                for (int e = hi - 1; e >= lo; --e) {
                    Cm[e] = ps_message<MATH>(Cm[e] * temp, neg, log_tab);
                    temp *= A[e];
                }
            } else {
                const int it = slot_iter[s] + 1;
                const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
                int parity = sb & 1;
                double temp = DBL_MAX;
                for (int e = lo; e < hi; ++e) {
                    const double bk = A[e];
                    if (bk <= 0) parity ^= 1;
                    Cm[e] = temp;
                    const double ab = fabs(bk);
                    if (ab < temp) temp = ab;
                }
                temp = DBL_MAX;
                for (int e = hi - 1; e >= lo; --e) {
                    const double bk = A[e];
                    const int sgn = parity ^ (bk <= 0 ? 1 : 0);
                    double mag = Cm[e];
                    if (temp < mag) mag = temp;
                    Cm[e] = mag * (sgn ? -alpha : alpha);
                    const double ab = fabs(bk);
                    if (ab < temp) temp = ab;
                }
            }
        }
        __syncthreads();

        // ---- bit pass (bp.hpp:276-298, 311-318): item = (slot, bit) ----
        for (int w = tid; w < S * n; w += T) {
            int s, j;
            split(w, n, inv_n, s, j);
            if (slot_synd[s] < 0) continue;
            double *A = reinterpret_cast<double *>(slot_base(s));
            double *Cm = A + nnz;
            double *L = Cm + nnz;
            uint8_t *hard = reinterpret_cast<uint8_t *>(L + n);
            const int lo = cp[j], hi = cp[j + 1];
            double temp = prior[j];
            for (int p = lo; p < hi; ++p) { const int e = ce[p]; A[e] = temp; temp += Cm[e]; }
            L[j] = temp;
            hard[j] = temp <= 0 ? 1 : 0;
            double sfx = 0.0;
            for (int p = hi - 1; p >= lo; --p) {
                const int e = ce[p];
                A[e] = edge_form<METHOD, MATH>(A[e] + sfx);
                sfx += Cm[e];
            }
        }
        __syncthreads();

        // ---- syndrome test (bp.hpp:292-294, 300-302): candidate parity of every check vs its syndrome BYTE ----
        for (int w = tid; w < S * m; w += T) {
            int s, i;
            split(w, m, inv_m, s, i);
            if (slot_synd[s] < 0) continue;
            const uint8_t *hard = slot_base(s) + (size_t)nnz * 16 + (size_t)n * 8;
            const uint8_t sb = (slot_base(s) + (size_t)nnz * 16 + (size_t)n * 9)[i];
            uint8_t par = 0;
            for (int e = rp[i]; e < rp[i + 1]; ++e) par ^= hard[ci[e]];
            if (par != sb) atomicOr(&slot_unsat[s], 1);
        }
        __syncthreads();
        if (tid < S && slot_synd[tid] >= 0) {
            const int it = ++slot_iter[tid];
            if (!slot_unsat[tid] || it >= a.max_iter) slot_state[tid] = 1;
        }
        __syncthreads();

        // ---- finished slots: outputs (bp.hpp:62,65,69,71), then pull the next syndrome ----
        for (int w = tid; w < S * n; w += T) {
            int s, j;
            split(w, n, inv_n, s, j);
            if (slot_state[s] != 1) continue;
            const double *L = reinterpret_cast<const double *>(slot_base(s)) + 2 * (size_t)nnz;
            const uint8_t *hard = reinterpret_cast<const uint8_t *>(L + n);
            const long long b = slot_synd[s];
            a.decoding[b * n + j] = hard[j];
            if (a.llr) a.llr[b * n + j] = L[j];
        }
        __syncthreads();
        if (tid < S) {
            if (slot_state[tid] == 1) {
                const long long b = slot_synd[tid];
                if (a.iters) a.iters[b] = slot_iter[tid];
                if (a.conv) a.conv[b] = slot_unsat[tid] ? 0 : 1;
                const unsigned long long idx = atomicAdd(a.next, 1ull);
                slot_synd[tid] = idx < (unsigned long long)a.batch ? (long long)idx : -1;
                slot_state[tid] = 2;
                slot_iter[tid] = 0;
            }
            slot_unsat[tid] = 0;
        }
        __syncthreads();
    }
}

// ---- OSD-0 (osd.hpp:110-117 = sort.hpp:48-62 + gf2sparse_linalg.hpp:298-401, 237-288) -------------
// One wavefront per syndrome that BP left unconverged.  The reference sorts the columns by ascending
// log-ratio (glibc qsort: stable, so ties keep ascending index), runs a greedy column-ordered Gaussian
// elimination on a linked-list matrix until the syndrome is in the span of the pivots, and solves on
// the pivot columns.  That solution is unique given the column order (the reference's min-row-weight
// pivoting only picks which ROW carries a pivot), so here the augmented matrix [H | s] lives bit-packed
// in LDS (lane l owns rows l, l+64, ...), columns are visited in rank order and eliminated
// Gauss-Jordan style with wave ballots.  All LDS traffic is wave-private: no workgroup barriers.
// ---- soft-syndrome serial min-sum: BpDecoder::soft_info_decode_serial (bp.hpp:547-660) ---------------------
// One wavefront per 64-shot tile, lane = shot.  The scaled analog syndrome S[tile][check][lane] and the hard
// syndrome (one ballot word per check, in LDS) are part of the decoder state: a check whose |S| is below the
// cutoff and below the smallest incoming magnitude behaves as a virtual variable node (bp.hpp:597-621).
struct SoftArgs {
    int32_t m, n, nnz, max_iter;
    double ms_scaling_factor, cutoff;
    int64_t batch;
    const int32_t *row_ptr, *col_idx, *col_ptr, *csc_edge, *csc_row, *order;  // order may be nullptr (0..n-1)
    const double *llr0;
    double *A;        // [tiles][nnz][64] bit->check messages
    double *C;        // [tiles][nnz][64] check->bit messages of the bit being updated
    double *S;        // [tiles][m][64]   in: 2 s / sigma^2, out: the soft syndrome after decoding
    const uint64_t *syn;  // [tiles][m]   hard syndrome (S <= 0) at the start
    uint64_t *dec, *dcur;
    double *llr_t;
    int32_t *iters;
    uint8_t *conv;
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
        for (int t = 0; t < n; ++t) {
            const int bit = a.order ? sload(a.order + t) : t;
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

struct OsdArgs {
    int32_t m, n, words;  // words = ceil((n + 1) / 64): n matrix bits + the syndrome bit per row
    int64_t batch;
    const int32_t *row_ptr, *col_idx;
    const uint8_t *synd;   // [batch][m]
    const double *llr;     // [batch][n]  BP posteriors
    const uint8_t *conv;   // [batch]     1 = BP converged: row left untouched
    uint8_t *decoding;     // [batch][n]  in: BP decisions, out: OSD solution for unconverged rows
    int32_t lds_per_wave;  // bytes
    int32_t method, order; // osdw_kernel: 2 = exhaustive (OSD_E), 3 = combination sweep (OSD_CS); order > 0
    const double *wt;      // [n] log(1 / p_j): the weight of bit j in a candidate (osd.hpp:134, 173)
};

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
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= a.batch || a.conv[b]) return;  // wave-uniform
    const int m = a.m, n = a.n, W = a.words;
    unsigned char *base = osd_lds + (size_t)wave * a.lds_per_wave;
    volatile uint64_t *mat = reinterpret_cast<volatile uint64_t *>(base);                  // [m][W]
    volatile double *keys = reinterpret_cast<volatile double *>(base + (size_t)m * W * 8);  // [n]
    volatile int32_t *order = reinterpret_cast<volatile int32_t *>(base + (size_t)m * W * 8 + (size_t)n * 8);  // [n]
    volatile int32_t *pivot_col = order + n;                                                // [m]
    volatile uint8_t *x = reinterpret_cast<volatile uint8_t *>(const_cast<int32_t *>(pivot_col + m));  // [n]

    const int sw = n >> 6;
    const uint64_t sbit = 1ull << (n & 63);
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
    int32_t single; // a chosen non-pivot column beyond the first 64 (OSD_CS weight-one strings), else -1
    bool valid;
};

__device__ __forceinline__ OsdCandidate osd_candidate(int method, int order, int k, long c) {
    OsdCandidate r;
    r.mask = 0;
    r.single = -1;
    r.valid = true;
    const uint64_t kmask = k >= 64 ? ~0ull : ((1ull << k) - 1ull);
    if (method == 2) {  // numbers 1 .. 2^order - 1, bit j -> j-th non-pivot column, bits >= k dropped (util.hpp:12-38)
        r.mask = (uint64_t)(c + 1) & kmask;
    } else if (c < k) {  // weight one, every non-pivot column (osd.hpp:84-89)
        if (c < 64) r.mask = 1ull << c; else r.single = (int32_t)c;
    } else {  // pairs (i, j), i < j < order, i-major (osd.hpp:91-99)
        long p = c - k;
        int i = 0;
        while (p >= order - 1 - i) { p -= order - 1 - i; ++i; }
        const int j = i + 1 + (int)p;
        if (j >= k) r.valid = false;  // past the candidate string in the reference
        else r.mask = (1ull << i) | (1ull << j);
    }
    return r;
}

__global__ void __launch_bounds__(256) osdw_kernel(const OsdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
    if (b >= a.batch || a.conv[b]) return;  // wave-uniform
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
            return (v & 1ull) != 0;
        }
        const int q = -1 - cdi;
        return (q < 64 && ((cd.mask >> q) & 1ull)) || q == cd.single;
    };
    auto weight_of = [&](const OsdCandidate &cd) -> double {
        double acc = 0;
        for (int i = 0; i < n; ++i)
            if (bit_of(cd, i)) acc += keys[i];
        return acc;
    };
    OsdCandidate none;
    none.mask = 0; none.single = -1; none.valid = true;
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
}

// GF2Sparse::mulvec over a batch (gf2sparse.hpp:177-214): one thread per (vector, check)
__global__ void gf2_mulvec_kernel(const int32_t *__restrict__ row_ptr,
                                  const int32_t *__restrict__ col_idx, int m, int n,
                                  const uint8_t *__restrict__ in, int64_t batch, uint8_t *out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * m) return;
    const int64_t b = t / m;
    const int i = (int)(t - b * m);
    uint8_t s = 0;
    for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e) s ^= in[b * n + col_idx[e]];
    out[t] = s;
}

// ---- bit-packed shot data ("b8": bit i of a shot is bit i % 8 of its byte i / 8; every shot starts on a byte) --
// the wire format of the reference's sinter decoders (sinter_decoders/sinter_bposd_decoder.py:57-130)
__global__ void unpack_b8_kernel(const uint8_t *__restrict__ in, int64_t batch, int bits, uint8_t *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * bits) return;
    const int64_t b = t / bits;
    const int i = (int)(t - b * bits);
    out[t] = (in[b * ((bits + 7) >> 3) + (i >> 3)] >> (i & 7)) & 1;
}

__global__ void pack_b8_kernel(const uint8_t *__restrict__ in, int64_t batch, int bits, uint8_t *__restrict__ out) {
    const int nb = (bits + 7) >> 3;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * nb) return;
    const int64_t b = t / nb;
    const int byte = (int)(t - b * nb);
    uint8_t v = 0;
    for (int q = 0; q < 8 && byte * 8 + q < bits; ++q) v |= (uint8_t)((in[b * bits + byte * 8 + q] & 1) << q);
    out[t] = v;
}

// BpDecoder.decode / BpOsdDecoder.decode return the zero vector for an all-zero input without running BP
// (_bp_decoder.pyx:679-681, _bposd_decoder.pyx:118-123): converge = True, iterations reported as 0 by the batch API
__global__ void zero_shot_shortcut_kernel(const uint8_t *__restrict__ dets_b8, int64_t batch, int m, int n, uint8_t *dec,
                                          int32_t *iters, uint8_t *conv) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int mb = (m + 7) >> 3;
    uint8_t any = 0;
    for (int q = 0; q < mb; ++q) {
        uint8_t v = dets_b8[b * mb + q];
        if (q == mb - 1 && (m & 7)) v &= (uint8_t)((1u << (m & 7)) - 1u);  // padding bits carry no data
        any |= v;
    }
    if (any) return;
    for (int j = 0; j < n; ++j) dec[b * n + j] = 0;
    if (iters) iters[b] = 0;
    if (conv) conv[b] = 1;
}

// predicted observables L x (mod 2) of every decoding, bit-packed: one thread per (shot, output byte)
// (SinterBpOsdDecoder.decode: `(observables_matrix @ corr) % 2`, sinter_bposd_decoder.py:128-130)
__global__ void observables_b8_kernel(const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col_idx, int k, int n,
                                      const uint8_t *__restrict__ dec, int64_t batch, uint8_t *__restrict__ out) {
    const int nb = (k + 7) >> 3;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * nb) return;
    const int64_t b = t / nb;
    const int byte = (int)(t - b * nb);
    uint8_t v = 0;
    for (int q = 0; q < 8 && byte * 8 + q < k; ++q) {
        const int o = byte * 8 + q;
        uint8_t s = 0;
        for (int e = row_ptr[o]; e < row_ptr[o + 1]; ++e) s ^= dec[b * n + col_idx[e]];
        v |= (uint8_t)((s & 1) << q);
    }
    out[t] = v;
}

// synthetic BSC shots: syndrome[b][i] = XOR_{j in row i} bernoulli(seed, (shot0+b)*n + j)
__global__ void gen_bsc_syndromes_kernel(const int32_t *__restrict__ row_ptr,
                                         const int32_t *__restrict__ col_idx, int m, int n,
                                         uint64_t seed, uint64_t threshold, int64_t shot0,
                                         int64_t batch, uint8_t *synd) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * m) return;
    const int64_t b = t / m;
    const int i = (int)(t - b * m);
    const uint64_t base = (uint64_t)(shot0 + b) * (uint64_t)n;
    uint8_t s = 0;
    for (int e = row_ptr[i]; e < row_ptr[i + 1]; ++e)
        s ^= (uint8_t)((sm64(seed, base + (uint64_t)col_idx[e]) >> 11) < threshold);
    synd[t] = s;
}

__global__ void gen_bsc_errors_kernel(int n, uint64_t seed, uint64_t threshold, int64_t shot0,
                                      int64_t batch, uint8_t *err) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= batch * n) return;
    const uint64_t idx = (uint64_t)shot0 * (uint64_t)n + (uint64_t)t;
    err[t] = (uint8_t)((sm64(seed, idx) >> 11) < threshold);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(_e == hipErrorOutOfMemory ? LDPC_HIP_ERR_NOMEM : LDPC_HIP_ERR_DEVICE, \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,      \
                        __LINE__);                                                            \
    } while (0)

struct DeviceBuf {  // grow-only device allocation
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(LDPC_HIP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes,
                        hipGetErrorString(e));
        }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct ldpc_hip_bp {
    int device = 0;
    int32_t m = 0, n = 0, nnz = 0;
    int32_t max_iter = 1, bp_method = 0;
    double ms_scaling_factor = 1.0;
    int32_t max_row_deg = 0, max_col_deg = 0;
    int32_t waves_per_wg = 0;  // 0 = auto
    int32_t math_mode = LDPC_HIP_MATH_LIBM_EXACT;
    bool regular = false;   // every row has the same weight and every column has the same weight
    int32_t ring_depth = 2; // LDS-DMA ring slots per wavefront for regular matrices (0 = register variant)
    int32_t small_mode = -1; // on-chip kernel for small codes: -1 auto, 0 never, 1 whenever it fits
    int32_t handoff = -1;    // straggler hand-off threshold in tiles: -1 auto (256), 0 off
    DeviceBuf tile_state, handoff_list;
    unsigned *h_counters = nullptr;  // pinned host copy of the device counters
    int32_t schedule = 1;    // ldpc::bp::BpSchedule (bp.hpp:28-32): 0 serial (fixed order), 1 parallel
    int32_t *d_csc_row = nullptr, *d_order = nullptr;
    bool custom_order = false;
    DeviceBuf counter;
    std::vector<double> channel_probs;

    int32_t *d_row_ptr = nullptr, *d_col_idx = nullptr, *d_col_ptr = nullptr, *d_csc_edge = nullptr;
    double *d_llr0 = nullptr;
    double *d_osd_wt = nullptr;  // [n] log(1 / p_j), the candidate weights of higher-order OSD
    int32_t osd_method = 1, osd_order = 0;  // ldpc::osd::OsdMethod (osd.hpp:18-23) used by ldpc_hip_bposd_decode_batch

    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    float accumulated_ms = 0.f;

    DeviceBuf msgA, msgC, par, nzm, invalid, dec, dcur, llr_t;       // workspace
    DeviceBuf st_synd, st_dec, st_llr, st_iters, st_conv, st_misc;  // staging for host pointers
    DeviceBuf osd_llr, osd_conv;                                    // BP outputs OSD-0 needs when the caller does not ask for them
    DeviceBuf soft_S, soft_in, soft_out;                             // soft-syndrome decoding: scaled analog syndromes, staging
    DeviceBuf b8_in, b8_out, b8_synd, b8_dec, obs_row_ptr, obs_col_idx;  // bit-packed shot I/O and the observables matrix
    int32_t obs_k = -1;                                              // rows of the observables matrix (-1: not set)
    int64_t max_chunk_tiles = 0;                                     // 0 = decide from free memory
};

static int upload_priors(ldpc_hip_bp *h) {
    // bp.hpp:150-151, evaluated by the host libm so that priors are bit-identical to the reference's
    std::vector<double> llr0((size_t)h->n);
    for (int j = 0; j < h->n; ++j)
        llr0[(size_t)j] = std::log((1 - h->channel_probs[(size_t)j]) / h->channel_probs[(size_t)j]);
    HIPCHK(hipMemcpy(h->d_llr0, llr0.data(), sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice));
    for (int j = 0; j < h->n; ++j) llr0[(size_t)j] = std::log(1 / h->channel_probs[(size_t)j]);  // osd.hpp:134
    HIPCHK(hipMemcpy(h->d_osd_wt, llr0.data(), sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice));
    return 0;
}

static bool is_device_ptr(const void *p) {
    if (!p) return true;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory reports an error: clear it
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

extern "C" {

const char *ldpc_hip_last_error(void) { return g_last_error.c_str(); }
const char *ldpc_hip_version(void) { return "ldpc_hip 0.1 (gfx950)"; }

int ldpc_hip_bp_create(const ldpc_hip_bp_desc *d, ldpc_hip_bp **out) {
    if (!d || !out) return fail(LDPC_HIP_ERR_INVALID, "null descriptor or output");
    *out = nullptr;
    if (d->m < 0 || d->n < 0 || !d->csr_row_ptr || (d->nnz > 0 && !d->csr_col_idx) || !d->channel_probs)
        return fail(LDPC_HIP_ERR_INVALID, "bad matrix description");
    if (d->csr_row_ptr[0] != 0 || d->csr_row_ptr[d->m] != d->nnz)
        return fail(LDPC_HIP_ERR_INVALID, "csr_row_ptr[0] must be 0 and csr_row_ptr[m] == nnz");
    if (d->max_iter < 1) return fail(LDPC_HIP_ERR_INVALID, "max_iter must be >= 1");
    if (d->bp_method != LDPC_HIP_PRODUCT_SUM && d->bp_method != LDPC_HIP_MINIMUM_SUM)
        return fail(LDPC_HIP_ERR_INVALID, "bp_method must be 0 (product_sum) or 1 (minimum_sum)");
    if (d->nnz >= (1 << 23) || d->n >= (1 << 23))  // one buffer descriptor spans a tile: rows * 512 B < 4 GiB
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "matrices with nnz or n >= 2^23 are not supported");
    int32_t max_row = 0;
    for (int i = 0; i < d->m; ++i) {
        const int lo = d->csr_row_ptr[i], hi = d->csr_row_ptr[i + 1];
        if (hi < lo) return fail(LDPC_HIP_ERR_INVALID, "csr_row_ptr not monotone at row %d", i);
        if (hi - lo > max_row) max_row = hi - lo;
        for (int e = lo; e < hi; ++e) {
            if (d->csr_col_idx[e] < 0 || d->csr_col_idx[e] >= d->n)
                return fail(LDPC_HIP_ERR_INVALID, "column index out of range in row %d", i);
            if (e > lo && d->csr_col_idx[e] <= d->csr_col_idx[e - 1])
                return fail(LDPC_HIP_ERR_INVALID, "row %d: column indices must be strictly ascending", i);
        }
    }
    int device = d->device;
    if (device < 0) HIPCHK(hipGetDevice(&device));
    HIPCHK(hipSetDevice(device));

    auto *h = new ldpc_hip_bp;
    h->device = device;
    h->m = d->m; h->n = d->n; h->nnz = d->nnz;
    h->max_iter = d->max_iter; h->bp_method = d->bp_method;
    h->ms_scaling_factor = d->ms_scaling_factor;
    h->max_row_deg = max_row;
    int32_t min_row = d->m ? max_row : 0;
    for (int i = 0; i < d->m; ++i)
        if (d->csr_row_ptr[i + 1] - d->csr_row_ptr[i] < min_row) min_row = d->csr_row_ptr[i + 1] - d->csr_row_ptr[i];
    h->channel_probs.assign(d->channel_probs, d->channel_probs + d->n);

    // CSC view: csc_edge[p] = CSR edge id; filling by ascending row keeps rows ascending per column
    std::vector<int32_t> col_ptr((size_t)d->n + 1, 0), csc_edge((size_t)(d->nnz ? d->nnz : 1)), csc_row((size_t)(d->nnz ? d->nnz : 1));
    for (int e = 0; e < d->nnz; ++e) col_ptr[(size_t)d->csr_col_idx[e] + 1]++;
    int32_t min_col = d->n ? INT32_MAX : 0;
    for (int j = 0; j < d->n; ++j) {
        if (col_ptr[(size_t)j + 1] > h->max_col_deg) h->max_col_deg = col_ptr[(size_t)j + 1];
        if (col_ptr[(size_t)j + 1] < min_col) min_col = col_ptr[(size_t)j + 1];
        col_ptr[(size_t)j + 1] += col_ptr[(size_t)j];
    }
    h->regular = d->m > 0 && d->n > 0 && min_row == max_row && min_col == h->max_col_deg;
    {
        std::vector<int32_t> fill(col_ptr.begin(), col_ptr.end() - 1);
        for (int i = 0; i < d->m; ++i)
            for (int e = d->csr_row_ptr[i]; e < d->csr_row_ptr[i + 1]; ++e)
            {
                const size_t pos = (size_t)fill[(size_t)d->csr_col_idx[e]]++;
                csc_edge[pos] = e;
                csc_row[pos] = i;
            }
    }
#define ALLOC_COPY(dst, src, count, T)                                                          \
    do {                                                                                        \
        hipError_t _e = hipMalloc((void **)&(dst), sizeof(T) * (size_t)((count) ? (count) : 1)); \
        if (_e == hipSuccess && (count))                                                        \
            _e = hipMemcpy((dst), (src), sizeof(T) * (size_t)(count), hipMemcpyHostToDevice);   \
        if (_e != hipSuccess) {                                                                 \
            ldpc_hip_bp_destroy(h);                                                             \
            return fail(LDPC_HIP_ERR_DEVICE, "device upload failed: %s", hipGetErrorString(_e)); \
        }                                                                                       \
    } while (0)
    ALLOC_COPY(h->d_row_ptr, d->csr_row_ptr, d->m + 1, int32_t);
    ALLOC_COPY(h->d_col_idx, d->csr_col_idx, d->nnz, int32_t);
    ALLOC_COPY(h->d_col_ptr, col_ptr.data(), d->n + 1, int32_t);
    ALLOC_COPY(h->d_csc_edge, csc_edge.data(), d->nnz, int32_t);
    ALLOC_COPY(h->d_csc_row, csc_row.data(), d->nnz, int32_t);
    ALLOC_COPY(h->d_llr0, d->channel_probs, d->n, double);  // overwritten by upload_priors
    ALLOC_COPY(h->d_osd_wt, d->channel_probs, d->n, double);  // likewise
#undef ALLOC_COPY
    int rc = upload_priors(h);
    if (rc) { ldpc_hip_bp_destroy(h); return rc; }
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e != hipSuccess) {
        ldpc_hip_bp_destroy(h);
        return fail(LDPC_HIP_ERR_DEVICE, "stream/event creation failed: %s", hipGetErrorString(e));
    }
    h->stream = h->own_stream;
    *out = h;
    return LDPC_HIP_OK;
}

void ldpc_hip_bp_destroy(ldpc_hip_bp *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    for (DeviceBuf *b : {&h->msgA, &h->msgC, &h->par, &h->nzm, &h->invalid, &h->dec, &h->dcur, &h->llr_t,
                         &h->st_synd, &h->st_dec, &h->st_llr, &h->st_iters, &h->st_conv, &h->st_misc, &h->osd_llr, &h->osd_conv, &h->counter,
                         &h->soft_S, &h->soft_in, &h->soft_out, &h->b8_in, &h->b8_out, &h->b8_synd, &h->b8_dec, &h->obs_row_ptr, &h->obs_col_idx,
                         &h->tile_state, &h->handoff_list})
        b->release();
    if (h->d_row_ptr) (void)hipFree(h->d_row_ptr);
    if (h->d_col_idx) (void)hipFree(h->d_col_idx);
    if (h->d_col_ptr) (void)hipFree(h->d_col_ptr);
    if (h->d_csc_edge) (void)hipFree(h->d_csc_edge);
    if (h->d_csc_row) (void)hipFree(h->d_csc_row);
    if (h->d_order) (void)hipFree(h->d_order);
    if (h->h_counters) (void)hipHostFree(h->h_counters);
    if (h->d_llr0) (void)hipFree(h->d_llr0);
    if (h->d_osd_wt) (void)hipFree(h->d_osd_wt);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
    if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
    delete h;
}

int ldpc_hip_bp_set_channel(ldpc_hip_bp *h, const double *p, int32_t n) {
    if (!h || !p) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    if (n != h->n)  // bp.hpp:103-106
        return fail(LDPC_HIP_ERR_INVALID,
                    "Channel probabilities vector must have length equal to the number of bits");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    h->channel_probs.assign(p, p + n);
    return upload_priors(h);
}

int ldpc_hip_bp_set_params(ldpc_hip_bp *h, int32_t max_iter, int32_t bp_method, double alpha) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (max_iter < 1) return fail(LDPC_HIP_ERR_INVALID, "max_iter must be >= 1");
    if (bp_method != LDPC_HIP_PRODUCT_SUM && bp_method != LDPC_HIP_MINIMUM_SUM)
        return fail(LDPC_HIP_ERR_INVALID, "bp_method must be 0 (product_sum) or 1 (minimum_sum)");
    h->max_iter = max_iter;
    h->bp_method = bp_method;
    h->ms_scaling_factor = alpha;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_stream(ldpc_hip_bp *h, void *s) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (s == LDPC_HIP_STREAM_LEGACY_DEFAULT) h->stream = nullptr;  // hipStream_t 0: the device's legacy default stream
    else h->stream = s ? (hipStream_t)s : h->own_stream;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_tuning(ldpc_hip_bp *h, int32_t waves_per_wg, int32_t max_chunk_tiles) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (waves_per_wg < 0 || waves_per_wg > 16)
        return fail(LDPC_HIP_ERR_INVALID, "waves_per_workgroup must be in [0, 16]");
    h->waves_per_wg = waves_per_wg;
    h->max_chunk_tiles = max_chunk_tiles > 0 ? max_chunk_tiles : 0;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_ring(ldpc_hip_bp *h, int32_t enable) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (enable < 0 || enable > 3) return fail(LDPC_HIP_ERR_INVALID, "ring depth must be 0 (off), 1 (default depth), 2 or 3");
    h->ring_depth = enable == 1 ? 2 : enable;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_schedule(ldpc_hip_bp *h, int32_t schedule, const int32_t *serial_schedule_order) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (schedule == 2)
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "serial_relative (per-syndrome LLR-sorted order, bp.hpp:470-483) is not available on the device");
    if (schedule != 0 && schedule != 1) return fail(LDPC_HIP_ERR_INVALID, "Invalid BP schedule");  // bp.hpp:188
    HIPCHK(hipSetDevice(h->device));
    if (schedule == 0 && serial_schedule_order) {
        for (int j = 0; j < h->n; ++j)
            if (serial_schedule_order[j] < 0 || serial_schedule_order[j] >= h->n)
                return fail(LDPC_HIP_ERR_INVALID, "serial_schedule_order[%d] is out of range", j);
        if (!h->d_order) HIPCHK(hipMalloc((void **)&h->d_order, sizeof(int32_t) * (size_t)(h->n ? h->n : 1)));
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->d_order, serial_schedule_order, sizeof(int32_t) * (size_t)h->n, hipMemcpyHostToDevice));
        h->custom_order = true;
    } else {
        h->custom_order = false;
    }
    h->schedule = schedule;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_handoff(ldpc_hip_bp *h, int32_t threshold_tiles) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (threshold_tiles < -1) return fail(LDPC_HIP_ERR_INVALID, "threshold must be -1 (auto), 0 (off) or a tile count");
    h->handoff = threshold_tiles > 32768 ? 32768 : threshold_tiles;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_small_code_kernel(ldpc_hip_bp *h, int32_t mode) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (mode < -1 || mode > 1) return fail(LDPC_HIP_ERR_INVALID, "mode must be -1 (auto), 0 (off) or 1 (whenever it fits)");
    h->small_mode = mode;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_math(ldpc_hip_bp *h, int32_t math_mode) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (math_mode != LDPC_HIP_MATH_LIBM_EXACT && math_mode != LDPC_HIP_MATH_FAST)
        return fail(LDPC_HIP_ERR_INVALID, "math_mode must be 0 (libm-exact) or 1 (fast)");
    h->math_mode = math_mode;
    return LDPC_HIP_OK;
}

int64_t ldpc_hip_bp_workspace_bytes(const ldpc_hip_bp *h, int64_t batch) {
    if (!h || batch < 0) return -1;
    const int64_t tiles = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    return tiles * (2ll * 8 * h->nnz * LDPC_WAVE + 2ll * 8 * h->m + 8 + 8ll * h->n +
                    8ll * h->n * LDPC_WAVE);
}

int ldpc_hip_bp_last_kernel_ms(ldpc_hip_bp *h, float *ms) {
    if (!h || !ms) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    *ms = 0.f;
    if (!h->timed) return LDPC_HIP_OK;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipEventSynchronize(h->ev1));
    float last = 0.f;
    HIPCHK(hipEventElapsedTime(&last, h->ev0, h->ev1));
    *ms = h->accumulated_ms + last;
    return LDPC_HIP_OK;
}

}  // extern "C"

typedef void (*bp_kernel_t)(const BpArgs);
typedef void (*spread_kernel_t)(const SpreadArgs);

template <int METHOD, int MATH>
static void pick_spread_m(int max_row, int max_col, spread_kernel_t &kc, spread_kernel_t &kb) {
    kc = max_row <= 8 ? bp_spread_check_kernel<METHOD, MATH, 8> : bp_spread_check_kernel<METHOD, MATH, 16>;
    kb = max_col <= 4 ? bp_spread_bit_kernel<METHOD, MATH, 4> : bp_spread_bit_kernel<METHOD, MATH, 8>;
}

struct KernelChoice {
    bp_kernel_t fn;
    int ring_slot_bytes;  // 0: register-prefetch variant, no dynamic LDS
    int ring_depth;
};

template <int METHOD, int MATH>
static KernelChoice pick_kernel(int max_row, int max_col, int ring_depth) {
    // Register arrays are sized by the template bounds, so the common regular codes get exact fits:
    // (3,6)-LDPC / bivariate-bicycle rows of 6 and columns of 3 use the LDS-DMA ring variant.
    if (ring_depth == 2 && max_row == 6 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 2>, 3 * 1024, 2};
    if (ring_depth >= 3 && max_row == 6 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 3>, 3 * 1024, 3};
    if (ring_depth >= 2 && max_row == 8 && max_col == 4) return {bp_decode_kernel<METHOD, MATH, 8, 4, 3>, 4 * 1024, 3};
    if (max_row <= 4 && max_col <= 3) return {bp_decode_kernel<METHOD, MATH, 4, 3, 0>, 0, 0};
    if (max_row <= 6 && max_col <= 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 0>, 0, 0};
    if (max_row <= 8 && max_col <= 4) return {bp_decode_kernel<METHOD, MATH, 8, 4, 0>, 0, 0};
    if (max_row <= 8 && max_col <= 8) return {bp_decode_kernel<METHOD, MATH, 8, 8, 0>, 0, 0};
    if (max_col <= 8) return {bp_decode_kernel<METHOD, MATH, 16, 8, 0>, 0, 0};
    return {bp_decode_kernel<METHOD, MATH, 16, 16, 0>, 0, 0};  // heavier nodes take the streaming path inside
}


static int decode_device(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                         double *llr, int32_t *iters, uint8_t *conv);

// Serial schedule: one wavefront per 64-syndrome tile (bp_serial_kernel).  Device pointers, on h->stream.
static int decode_serial(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                         int32_t *iters, uint8_t *conv) {
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    const size_t per_tile_msg = sizeof(double) * (size_t)(h->nnz ? h->nnz : 1) * LDPC_WAVE;
    const size_t per_tile_llr = llr ? sizeof(double) * (size_t)(h->n ? h->n : 1) * LDPC_WAVE : 0;
    const bool fast = h->max_col_deg <= 4 && h->max_row_deg <= 8;
    int64_t chunk = tiles_total;
    if (h->max_chunk_tiles > 0 && chunk > h->max_chunk_tiles) chunk = h->max_chunk_tiles;
    if (chunk > 32768) chunk = 32768;
    {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t have = h->msgA.cap + h->msgC.cap + h->llr_t.cap;
        const size_t budget = (size_t)((double)(free_b + have) * 0.85);
        const size_t per_tile = (fast ? 1 : 2) * per_tile_msg + per_tile_llr + 24 * (size_t)(h->m + h->n + 1);
        int64_t fit = (int64_t)(budget / (per_tile ? per_tile : 1));
        if (fit < 1) return fail(LDPC_HIP_ERR_NOMEM, "not enough device memory for one 64-syndrome tile");
        if (chunk > fit) chunk = fit;
    }
    int rc;
    if ((rc = h->msgA.ensure(per_tile_msg * (size_t)chunk))) return rc;
    if ((rc = h->msgC.ensure(fast ? 16 : per_tile_msg * (size_t)chunk))) return rc;
    if ((rc = h->par.ensure(sizeof(uint64_t) * (size_t)(h->m ? h->m : 1) * (size_t)chunk))) return rc;
    if ((rc = h->nzm.ensure(sizeof(uint64_t) * (size_t)(h->m ? h->m : 1) * (size_t)chunk))) return rc;
    if ((rc = h->invalid.ensure(sizeof(uint64_t) * (size_t)chunk))) return rc;
    if ((rc = h->dec.ensure(sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)chunk))) return rc;
    if ((rc = h->dcur.ensure(sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)chunk))) return rc;
    if (llr && (rc = h->llr_t.ensure(per_tile_llr * (size_t)chunk))) return rc;
    void (*kern)(const SerialArgs);
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = bp_serial_kernel<LDPC_HIP_MINIMUM_SUM, 0>;
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = bp_serial_kernel<LDPC_HIP_PRODUCT_SUM, 1>;
    else kern = bp_serial_kernel<LDPC_HIP_PRODUCT_SUM, 0>;
    h->accumulated_ms = 0.f;
    h->timed = false;
    hipStream_t st = h->stream;
    for (int64_t t0 = 0; t0 < tiles_total; t0 += chunk) {
        const int64_t tiles = (tiles_total - t0 < chunk) ? tiles_total - t0 : chunk;
        const int64_t b0 = t0 * LDPC_WAVE;
        const int64_t nb = (batch - b0 < tiles * LDPC_WAVE) ? batch - b0 : tiles * LDPC_WAVE;
        HIPCHK(hipMemsetAsync(h->invalid.p, 0, sizeof(uint64_t) * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dec.p, 0, sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dcur.p, 0, sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)tiles, st));
        if (h->m > 0) {
            dim3 g((unsigned)((h->m + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(pack_syndromes_kernel, g, dim3(256), 0, st, synd + b0 * h->m, nb, h->m,
                               (uint64_t *)h->par.p, (uint64_t *)h->nzm.p, (uint64_t *)h->invalid.p);
        }
        SerialArgs a = {};
        a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter; a.fast = fast ? 1 : 0;
        a.ms_scaling_factor = h->ms_scaling_factor;
        a.batch = nb;
        a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx; a.col_ptr = h->d_col_ptr;
        a.csc_edge = h->d_csc_edge; a.csc_row = h->d_csc_row; a.order = h->custom_order ? h->d_order : nullptr;
        a.llr0 = h->d_llr0;
        a.A = (double *)h->msgA.p; a.C = (double *)h->msgC.p;
        a.par = (const uint64_t *)h->par.p; a.invalid = (const uint64_t *)h->invalid.p;
        a.dec = (uint64_t *)h->dec.p; a.dcur = (uint64_t *)h->dcur.p;
        a.llr_t = llr ? (double *)h->llr_t.p : nullptr;
        a.iters = iters ? iters + b0 : nullptr;
        a.conv = conv ? conv + b0 : nullptr;
        if (h->timed) {
            float prev = 0.f;
            HIPCHK(hipEventSynchronize(h->ev1));
            HIPCHK(hipEventElapsedTime(&prev, h->ev0, h->ev1));
            h->accumulated_ms += prev;
        }
        HIPCHK(hipEventRecord(h->ev0, st));
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(64), 0, st, a);
        HIPCHK(hipEventRecord(h->ev1, st));
        h->timed = true;
        HIPCHK(hipGetLastError());
        if (h->n > 0) {
            dim3 g((unsigned)((h->n + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(unpack_decoding_kernel, g, dim3(256), 0, st, (const uint64_t *)h->dec.p, nb, h->n,
                               decoding + b0 * h->n);
            if (llr) {
                dim3 gt((unsigned)((h->n + LDPC_WAVE - 1) / LDPC_WAVE), (unsigned)tiles);
                hipLaunchKernelGGL(transpose_llr_kernel, gt, dim3(256), 0, st, (const double *)h->llr_t.p, nb, h->n,
                                   llr + (size_t)b0 * h->n);
            }
        }
        HIPCHK(hipGetLastError());
    }
    return LDPC_HIP_OK;
}

// soft_info_decode_serial over a batch (bp_softinfo_kernel).  Device pointers, on h->stream.
static int soft_info_device(ldpc_hip_bp *h, const double *soft, int64_t batch, double cutoff, double sigma, uint8_t *decoding,
                            double *llr, int32_t *iters, uint8_t *conv, double *soft_out) {
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    if (tiles_total == 0) return LDPC_HIP_OK;
    const size_t m1 = (size_t)(h->m ? h->m : 1), n1 = (size_t)(h->n ? h->n : 1);
    const size_t per_tile_msg = sizeof(double) * (size_t)(h->nnz ? h->nnz : 1) * LDPC_WAVE;
    const size_t per_tile_llr = llr ? sizeof(double) * n1 * LDPC_WAVE : 0;
    const size_t per_tile_soft = sizeof(double) * m1 * LDPC_WAVE;
    const size_t lds = sizeof(uint64_t) * m1;
    if (lds > 150u * 1024u)
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "soft-syndrome decoding keeps one hard-syndrome word per check in LDS: m <= 19200");
    int64_t chunk = tiles_total;
    if (h->max_chunk_tiles > 0 && chunk > h->max_chunk_tiles) chunk = h->max_chunk_tiles;
    if (chunk > 32768) chunk = 32768;
    {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t have = h->msgA.cap + h->msgC.cap + h->llr_t.cap + h->soft_S.cap;
        const size_t budget = (size_t)((double)(free_b + have) * 0.85);
        const size_t per_tile = 2 * per_tile_msg + per_tile_llr + per_tile_soft + 24 * (m1 + n1);
        int64_t fit = (int64_t)(budget / per_tile);
        if (fit < 1) return fail(LDPC_HIP_ERR_NOMEM, "not enough device memory for one 64-shot tile");
        if (chunk > fit) chunk = fit;
    }
    int rc;
    if ((rc = h->msgA.ensure(per_tile_msg * (size_t)chunk))) return rc;
    if ((rc = h->msgC.ensure(per_tile_msg * (size_t)chunk))) return rc;
    if ((rc = h->soft_S.ensure(per_tile_soft * (size_t)chunk))) return rc;
    if ((rc = h->par.ensure(sizeof(uint64_t) * m1 * (size_t)chunk))) return rc;
    if ((rc = h->dec.ensure(sizeof(uint64_t) * n1 * (size_t)chunk))) return rc;
    if ((rc = h->dcur.ensure(sizeof(uint64_t) * n1 * (size_t)chunk))) return rc;
    if (llr && (rc = h->llr_t.ensure(per_tile_llr * (size_t)chunk))) return rc;
    if (lds > 48u * 1024u)
        HIPCHK(hipFuncSetAttribute((const void *)bp_softinfo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->accumulated_ms = 0.f;
    h->timed = false;
    hipStream_t st = h->stream;
    for (int64_t t0 = 0; t0 < tiles_total; t0 += chunk) {
        const int64_t tiles = (tiles_total - t0 < chunk) ? tiles_total - t0 : chunk;
        const int64_t b0 = t0 * LDPC_WAVE;
        const int64_t nb = (batch - b0 < tiles * LDPC_WAVE) ? batch - b0 : tiles * LDPC_WAVE;
        HIPCHK(hipMemsetAsync(h->dec.p, 0, sizeof(uint64_t) * n1 * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dcur.p, 0, sizeof(uint64_t) * n1 * (size_t)tiles, st));
        if (h->m > 0) {
            dim3 g((unsigned)((h->m + 3) / 4), (unsigned)tiles);
            hipLaunchKernelGGL(softinfo_prepare_kernel, g, dim3(256), 0, st, soft + (size_t)b0 * h->m, nb, h->m, sigma,
                               (double *)h->soft_S.p, (uint64_t *)h->par.p);
        }
        SoftArgs a = {};
        a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter;
        a.ms_scaling_factor = h->ms_scaling_factor; a.cutoff = cutoff;
        a.batch = nb;
        a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx; a.col_ptr = h->d_col_ptr;
        a.csc_edge = h->d_csc_edge; a.csc_row = h->d_csc_row; a.order = h->custom_order ? h->d_order : nullptr;
        a.llr0 = h->d_llr0;
        a.A = (double *)h->msgA.p; a.C = (double *)h->msgC.p; a.S = (double *)h->soft_S.p;
        a.syn = (const uint64_t *)h->par.p;
        a.dec = (uint64_t *)h->dec.p; a.dcur = (uint64_t *)h->dcur.p;
        a.llr_t = llr ? (double *)h->llr_t.p : nullptr;
        a.iters = iters ? iters + b0 : nullptr;
        a.conv = conv ? conv + b0 : nullptr;
        if (h->timed) {
            float prev = 0.f;
            HIPCHK(hipEventSynchronize(h->ev1));
            HIPCHK(hipEventElapsedTime(&prev, h->ev0, h->ev1));
            h->accumulated_ms += prev;
        }
        HIPCHK(hipEventRecord(h->ev0, st));
        hipLaunchKernelGGL(bp_softinfo_kernel, dim3((unsigned)tiles), dim3(64), (unsigned)lds, st, a);
        HIPCHK(hipEventRecord(h->ev1, st));
        h->timed = true;
        HIPCHK(hipGetLastError());
        if (h->n > 0) {
            dim3 g((unsigned)((h->n + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(unpack_decoding_kernel, g, dim3(256), 0, st, (const uint64_t *)h->dec.p, nb, h->n,
                               decoding + b0 * h->n);
            if (llr) {
                dim3 gt((unsigned)((h->n + LDPC_WAVE - 1) / LDPC_WAVE), (unsigned)tiles);
                hipLaunchKernelGGL(transpose_llr_kernel, gt, dim3(256), 0, st, (const double *)h->llr_t.p, nb, h->n,
                                   llr + (size_t)b0 * h->n);
            }
        }
        if (soft_out && h->m > 0) {
            dim3 gt((unsigned)((h->m + LDPC_WAVE - 1) / LDPC_WAVE), (unsigned)tiles);
            hipLaunchKernelGGL(transpose_llr_kernel, gt, dim3(256), 0, st, (const double *)h->soft_S.p, nb, h->m,
                               soft_out + (size_t)b0 * h->m);
        }
        HIPCHK(hipGetLastError());
    }
    return LDPC_HIP_OK;
}

// LDS bytes of the on-chip kernel for `slots` resident syndromes; 0 if the code is too large for it
static size_t small_lds_bytes(const ldpc_hip_bp *h, int slots) {
    size_t fixed = 256 * 8 + (size_t)h->n * 8 + ((size_t)h->m + 1 + h->nnz + h->n + 1 + h->nnz) * 4;
    fixed = (fixed + 15) & ~(size_t)15;
    const size_t per_slot = ((size_t)h->nnz * 16 + (size_t)h->n * 9 + (size_t)h->m + 15) & ~(size_t)15;
    return fixed + per_slot * (size_t)slots;
}

// On-chip variant (bp_small_kernel): chosen automatically when four resident syndromes per workgroup still
// leave room for four workgroups per CU.  Device pointers, on h->stream.
static int decode_small(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                        int32_t *iters, uint8_t *conv, int slots) {
    int rc;
    if ((rc = h->counter.ensure(8))) return rc;
    HIPCHK(hipMemsetAsync(h->counter.p, 0, 8, h->stream));
    SmallArgs a = {};
    a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter; a.slots = slots;
    a.ms_scaling_factor = h->ms_scaling_factor;
    a.batch = batch;
    a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx; a.col_ptr = h->d_col_ptr; a.csc_edge = h->d_csc_edge;
    a.llr0 = h->d_llr0;
    a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    a.next = (unsigned long long *)h->counter.p;
    void (*kern)(const SmallArgs);
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = bp_small_kernel<LDPC_HIP_MINIMUM_SUM, 0>;
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = bp_small_kernel<LDPC_HIP_PRODUCT_SUM, 1>;
    else kern = bp_small_kernel<LDPC_HIP_PRODUCT_SUM, 0>;
    const size_t dyn = small_lds_bytes(h, slots);
    if (dyn > 48u * 1024u)
        HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    // persistent workgroups: enough to fill the chip, never more than there are syndromes to hand out
    int64_t groups = (batch + slots - 1) / slots;
    const int64_t resident = 256 * (int64_t)((150u * 1024u) / dyn > 8 ? 8 : (150u * 1024u) / dyn);
    if (groups > resident) groups = resident;
    h->accumulated_ms = 0.f;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(256), (unsigned)dyn, h->stream, a);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

static void pick_spread(const ldpc_hip_bp *h, spread_kernel_t &kc, spread_kernel_t &kb) {
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) pick_spread_m<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg, kc, kb);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) pick_spread_m<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg, kc, kb);
    else pick_spread_m<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg, kc, kb);
}

// Everything below runs on h->stream with device pointers only.
static int decode_device(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                         double *llr, int32_t *iters, uint8_t *conv) {
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    if (tiles_total == 0) return LDPC_HIP_OK;
    if (h->schedule == 0) return decode_serial(h, synd, batch, decoding, llr, iters, conv);
    if (h->small_mode != 0 && h->m > 0 && h->n > 0 && h->nnz > 0 && (int64_t)h->nnz * 16 < (1 << 22)) {
        // small code: keep the messages on chip.  auto: the most resident syndromes (<= 4) per workgroup that
        // still leave four workgroups per CU (<= 39.5 KiB each); forced: whatever fits in 150 KiB
        int slots = 0;
        const size_t budget = h->small_mode == 1 ? 150u * 1024u : 39u * 1024u + 512u;
        for (int sl = 4; sl >= 1 && !slots; --sl)
            if (small_lds_bytes(h, sl) <= budget) slots = sl;
        if (slots) return decode_small(h, synd, batch, decoding, llr, iters, conv, slots);
    }
    const size_t per_tile_msg = sizeof(double) * (size_t)(h->nnz ? h->nnz : 1) * LDPC_WAVE;
    const size_t per_tile_llr = llr ? sizeof(double) * (size_t)(h->n ? h->n : 1) * LDPC_WAVE : 0;

    int64_t chunk = tiles_total;
    if (h->max_chunk_tiles > 0 && chunk > h->max_chunk_tiles) chunk = h->max_chunk_tiles;
    if (chunk > 32768) chunk = 32768;  // grid.y of the pack/unpack launches stays below 65536
    {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t have = h->msgA.cap + h->msgC.cap + h->llr_t.cap;
        const size_t budget = (size_t)((double)(free_b + have) * 0.85);
        const size_t per_tile = 2 * per_tile_msg + per_tile_llr + 16 * (size_t)(h->m + h->n + 1);
        int64_t fit = (int64_t)(budget / (per_tile ? per_tile : 1));
        if (fit < 1) return fail(LDPC_HIP_ERR_NOMEM, "not enough device memory for one 64-syndrome tile");
        if (chunk > fit) chunk = fit;
    }
    int rc;
    if ((rc = h->msgA.ensure(per_tile_msg * (size_t)chunk))) return rc;
    if ((rc = h->msgC.ensure(per_tile_msg * (size_t)chunk))) return rc;
    if ((rc = h->par.ensure(sizeof(uint64_t) * (size_t)(h->m ? h->m : 1) * (size_t)chunk))) return rc;
    if ((rc = h->nzm.ensure(sizeof(uint64_t) * (size_t)(h->m ? h->m : 1) * (size_t)chunk))) return rc;
    if ((rc = h->invalid.ensure(sizeof(uint64_t) * (size_t)chunk))) return rc;
    if ((rc = h->dec.ensure(sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)chunk))) return rc;
    if ((rc = h->dcur.ensure(sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)chunk))) return rc;
    if (llr && (rc = h->llr_t.ensure(per_tile_llr * (size_t)chunk))) return rc;
    const int handoff = h->handoff < 0 ? 256 : h->handoff;
    if ((rc = h->tile_state.ensure(sizeof(TileState) * (size_t)chunk))) return rc;
    if ((rc = h->handoff_list.ensure(sizeof(int32_t) * (size_t)chunk))) return rc;
    if ((rc = h->counter.ensure(16))) return rc;
    if (!h->h_counters) HIPCHK(hipHostMalloc((void **)&h->h_counters, 16, hipHostMallocDefault));

    const int ring = h->regular ? h->ring_depth : 0;
    KernelChoice kern;
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_kernel<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg, ring);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_kernel<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg, ring);
    else kern = pick_kernel<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg, ring);
    h->accumulated_ms = 0.f;
    h->timed = false;
    hipStream_t st = h->stream;

    for (int64_t t0 = 0; t0 < tiles_total; t0 += chunk) {
        const int64_t tiles = (tiles_total - t0 < chunk) ? tiles_total - t0 : chunk;
        const int64_t b0 = t0 * LDPC_WAVE;
        const int64_t nb = (batch - b0 < tiles * LDPC_WAVE) ? batch - b0 : tiles * LDPC_WAVE;

        HIPCHK(hipMemsetAsync(h->invalid.p, 0, sizeof(uint64_t) * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dec.p, 0, sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)tiles, st));
        if (h->m > 0) {
            dim3 g((unsigned)((h->m + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(pack_syndromes_kernel, g, dim3(256), 0, st, synd + b0 * h->m, nb, h->m,
                               (uint64_t *)h->par.p, (uint64_t *)h->nzm.p, (uint64_t *)h->invalid.p);
        }
        BpArgs a = {};
        a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter;
        a.ms_scaling_factor = h->ms_scaling_factor;
        a.batch = nb;
        a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx;
        a.col_ptr = h->d_col_ptr; a.csc_edge = h->d_csc_edge;
        a.llr0 = h->d_llr0;
        a.A = (double *)h->msgA.p; a.C = (double *)h->msgC.p;
        a.par = (const uint64_t *)h->par.p; a.nzm = (const uint64_t *)h->nzm.p;
        a.invalid = (const uint64_t *)h->invalid.p;
        a.dec = (uint64_t *)h->dec.p;
        a.dcur = (uint64_t *)h->dcur.p;
        a.llr_t = llr ? (double *)h->llr_t.p : nullptr;
        a.iters = iters ? iters + b0 : nullptr;
        a.conv = conv ? conv + b0 : nullptr;
        a.state = (TileState *)h->tile_state.p;
        a.counters = (unsigned *)h->counter.p;
        a.handoff_list = (int32_t *)h->handoff_list.p;
        a.total_tiles = (int32_t)tiles;
        a.handoff_threshold = handoff;
        HIPCHK(hipMemsetAsync(h->counter.p, 0, 16, st));

        // Wavefronts per workgroup (one workgroup = one 64-syndrome tile).  Register variant: 128 VGPRs,
        // 16 wavefronts per CU -> 4-wave workgroups once there are >= 4 tiles per CU.  Ring variant:
        // ~70 VGPRs and 6 KiB of LDS per wavefront -> 24 wavefronts per CU as two 12-wave workgroups
        // (3 wavefronts on each SIMD; measured best on MI355X, profiles/; 6-wave workgroups place
        // unevenly on the 4 SIMDs and 8-wave ones leave a ragged last round at 1024 tiles).
        int waves = h->waves_per_wg;
        if (waves <= 0) {
            if (kern.ring_depth) waves = tiles >= 512 ? 12 : 16;
            else waves = tiles >= 1024 ? 4 : (tiles >= 512 ? 8 : 16);
        }
        if (waves > 16) waves = 16;
        // ring variant: each wavefront owns RING slots of dynamic LDS; stay below the 160 KiB of a CU
        const size_t lds_per_wave = (size_t)kern.ring_slot_bytes * (size_t)kern.ring_depth;
        while (lds_per_wave * (size_t)waves > 144u * 1024u) --waves;
        const size_t dyn_lds = lds_per_wave * (size_t)waves;
        if (dyn_lds > 48u * 1024u)
            HIPCHK(hipFuncSetAttribute((const void *)kern.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        if (h->timed) {  // fold the previous chunk's time before the events are re-recorded
            float prev = 0.f;
            HIPCHK(hipEventSynchronize(h->ev1));
            HIPCHK(hipEventElapsedTime(&prev, h->ev0, h->ev1));
            h->accumulated_ms += prev;
        }
        HIPCHK(hipEventRecord(h->ev0, st));
        SpreadArgs sa = {};
        sa.bp = a;
        unsigned parked = 0;
        int first_round = 1;  // a tile parked by the persistent kernel has completed >= 1 iteration
        if (handoff > 0 && tiles <= handoff && h->max_iter > 1) {
            // so few tiles that they would each sit on one compute unit: per-pass launches from the start
            parked = (unsigned)tiles;
            sa.n_tiles = (int32_t)parked;
            first_round = 0;
            hipLaunchKernelGGL(bp_spread_state_init_kernel, dim3((parked + 255) / 256), dim3(256), 0, st, sa);
            const dim3 gi((unsigned)((h->nnz + 63) / 64), parked);
            if (h->bp_method == LDPC_HIP_MINIMUM_SUM) hipLaunchKernelGGL((bp_spread_init_kernel<LDPC_HIP_MINIMUM_SUM, 0>), gi, dim3(256), 0, st, sa);
            else if (h->math_mode == LDPC_HIP_MATH_FAST) hipLaunchKernelGGL((bp_spread_init_kernel<LDPC_HIP_PRODUCT_SUM, 1>), gi, dim3(256), 0, st, sa);
            else hipLaunchKernelGGL((bp_spread_init_kernel<LDPC_HIP_PRODUCT_SUM, 0>), gi, dim3(256), 0, st, sa);
            HIPCHK(hipGetLastError());
        } else {
            hipLaunchKernelGGL(kern.fn, dim3((unsigned)tiles), dim3((unsigned)(waves * LDPC_WAVE)), (unsigned)dyn_lds, st, a);
            HIPCHK(hipGetLastError());
            if (handoff > 0 && h->max_iter > 1) {
                // tiles parked by the persistent kernel: the host needs their number (this is the one point where the
                // otherwise asynchronous call waits for the device)
                HIPCHK(hipMemcpyAsync(h->h_counters, h->counter.p, 16, hipMemcpyDeviceToHost, st));
                HIPCHK(hipStreamSynchronize(st));
                parked = h->h_counters[1];
            }
        }
        if (parked > 0) {
            // finish the parked tiles with chip-wide per-pass launches: check, bit, syndrome test, bookkeeping
            unsigned *live = (unsigned *)h->counter.p + 2;
            h->h_counters[3] = parked;
            HIPCHK(hipMemcpyAsync(live, &h->h_counters[3], sizeof(unsigned), hipMemcpyHostToDevice, st));
            sa.n_tiles = (int32_t)parked;
            sa.nodes = parked <= 8 ? 1 : 4;
            spread_kernel_t kc, kb;
            pick_spread(h, kc, kb);
            const unsigned per_wg = 4u * (unsigned)sa.nodes;
            const dim3 gc((unsigned)((h->m + per_wg - 1) / per_wg), parked), gb((unsigned)((h->n + per_wg - 1) / per_wg), parked);
            const dim3 gs((unsigned)((h->m + 255) / 256), parked), gf((unsigned)((h->n + 63) / 64), parked);
            const int rounds = h->max_iter - first_round;
            for (int round = 0; round < rounds; ++round) {
                sa.round = round;
                hipLaunchKernelGGL(kc, gc, dim3(256), 0, st, sa);
                hipLaunchKernelGGL(kb, gb, dim3(256), 0, st, sa);
                hipLaunchKernelGGL(bp_spread_synd_kernel, gs, dim3(256), 0, st, sa);
                hipLaunchKernelGGL(bp_spread_finish_kernel, gf, dim3(256), 0, st, sa, live);
                if ((round & 7) == 7 && round + 1 < rounds) {  // everything converged early?
                    HIPCHK(hipMemcpyAsync(&h->h_counters[2], live, sizeof(unsigned), hipMemcpyDeviceToHost, st));
                    HIPCHK(hipStreamSynchronize(st));
                    if (h->h_counters[2] == 0) break;
                }
            }
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(h->ev1, st));
        h->timed = true;
        HIPCHK(hipGetLastError());

        if (h->n > 0) {
            dim3 g((unsigned)((h->n + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(unpack_decoding_kernel, g, dim3(256), 0, st,
                               (const uint64_t *)h->dec.p, nb, h->n, decoding + b0 * h->n);
            if (llr) {
                dim3 gt((unsigned)((h->n + LDPC_WAVE - 1) / LDPC_WAVE), (unsigned)tiles);
                hipLaunchKernelGGL(transpose_llr_kernel, gt, dim3(256), 0, st,
                                   (const double *)h->llr_t.p, nb, h->n, llr + (size_t)b0 * h->n);
            }
        }
        HIPCHK(hipGetLastError());
    }
    return LDPC_HIP_OK;
}


// BP, then OSD-0 on the rows BP left unconverged; device pointers, on h->stream
static int bposd_device(ldpc_hip_bp *h, int osd_method, int osd_order, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                        double *llr, int32_t *iters, uint8_t *conv) {
    if (osd_method == 0)  // OSD_OFF: BpOsdDecoder still calls OsdDecoder::decode, which then has no LU object -- refuse instead
        return fail(LDPC_HIP_ERR_INVALID, "osd_method is OSD_OFF");
    const bool higher = osd_method >= 2 && osd_order > 0;  // osd_order == 0 takes the OSD-0 branch whatever the method (osd.hpp:114)
    const size_t B = (size_t)batch, n = (size_t)h->n;
    int rc;
    if (!llr) { if ((rc = h->osd_llr.ensure(B * n * 8 ? B * n * 8 : 1))) return rc; llr = (double *)h->osd_llr.p; }
    if (!conv) { if ((rc = h->osd_conv.ensure(B ? B : 1))) return rc; conv = (uint8_t *)h->osd_conv.p; }
    if ((rc = decode_device(h, synd, batch, decoding, llr, iters, conv))) return rc;
    if (h->m == 0 || h->n == 0) return LDPC_HIP_OK;
    OsdArgs a = {};
    a.m = h->m; a.n = h->n; a.words = (h->n + 1 + 63) / 64;
    a.batch = batch;
    a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx;
    a.synd = synd; a.llr = llr; a.conv = conv; a.decoding = decoding;
    a.method = osd_method; a.order = osd_order; a.wt = h->d_osd_wt;
    size_t per_wave = higher ? (size_t)a.m * a.words * 8 + (size_t)a.m * 8 + (size_t)a.n * 8 + 3 * (size_t)a.n * 4 + (size_t)a.m * 4
                             : (size_t)a.m * a.words * 8 + (size_t)a.n * 8 + (size_t)a.n * 4 + (size_t)a.m * 4 + (size_t)a.n;
    per_wave = (per_wave + 15) & ~(size_t)15;
    if (per_wave > 150u * 1024u)
        return fail(LDPC_HIP_ERR_UNSUPPORTED,
                    "OSD on the device keeps the bit-packed [H|s] of one syndrome in LDS: %zu bytes needed, 150 KiB available", per_wave);
    int waves = (int)((150u * 1024u) / per_wave);
    if (waves > 4) waves = 4;
    a.lds_per_wave = (int32_t)per_wave;
    const size_t dyn = per_wave * (size_t)waves;
    const void *fn = higher ? (const void *)osdw_kernel : (const void *)osd0_kernel;
    if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    const int64_t blocks = (batch + waves - 1) / waves;
    if (higher) hipLaunchKernelGGL(osdw_kernel, dim3((unsigned)blocks), dim3((unsigned)(waves * 64)), (unsigned)dyn, h->stream, a);
    else hipLaunchKernelGGL(osd0_kernel, dim3((unsigned)blocks), dim3((unsigned)(waves * 64)), (unsigned)dyn, h->stream, a);
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

extern "C" {

int ldpc_hip_bposd0_decode_batch_async(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch,
                                       uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    HIPCHK(hipSetDevice(h->device));
    return bposd_device(h, 1, 0, synd, batch, decoding, llr, iters, conv);
}

int ldpc_hip_bp_set_osd(ldpc_hip_bp *h, int32_t osd_method, int32_t osd_order) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (osd_method < 0 || osd_method > 3) return fail(LDPC_HIP_ERR_INVALID, "osd_method must be 0 (off), 1 (OSD_0), 2 (OSD_E) or 3 (OSD_CS)");
    if (osd_order < 0) return fail(LDPC_HIP_ERR_INVALID, "osd_order must not be negative");  // _bposd_decoder.pyx:222-223
    if (osd_method == 1 && osd_order != 0) return fail(LDPC_HIP_ERR_INVALID, "osd_method OSD_0 requires osd_order 0");  // pyx:225-226
    if (osd_method == 2 && osd_order > 24)
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "OSD_E with osd_order > 24 (more than 16 million candidates per syndrome) is not available");
    if (osd_method == 3 && osd_order > 64)
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "OSD_CS with osd_order > 64 is not available on the device");
    h->osd_method = osd_method;
    h->osd_order = osd_order;
    return LDPC_HIP_OK;
}

int ldpc_hip_bposd_decode_batch_async(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch,
                                      uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    HIPCHK(hipSetDevice(h->device));
    return bposd_device(h, h->osd_method, h->osd_order, synd, batch, decoding, llr, iters, conv);
}

int ldpc_hip_bp_decode_batch_async(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch,
                                   uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    if (batch > (1ll << 40)) return fail(LDPC_HIP_ERR_INVALID, "batch too large");
    HIPCHK(hipSetDevice(h->device));
    return decode_device(h, synd, batch, decoding, llr, iters, conv);
}

// osd: -1 BP only, 0 BP + OSD-0, 1 BP + the handle's osd_method / osd_order
static int decode_batch_staged(ldpc_hip_bp *h, int osd, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                               double *llr, int32_t *iters, uint8_t *conv);

int ldpc_hip_bp_decode_batch(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                             double *llr, int32_t *iters, uint8_t *conv) {
    return decode_batch_staged(h, -1, synd, batch, decoding, llr, iters, conv);
}

int ldpc_hip_bposd0_decode_batch(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                                 double *llr, int32_t *iters, uint8_t *conv) {
    return decode_batch_staged(h, 0, synd, batch, decoding, llr, iters, conv);
}

int ldpc_hip_bposd_decode_batch(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                                double *llr, int32_t *iters, uint8_t *conv) {
    return decode_batch_staged(h, 1, synd, batch, decoding, llr, iters, conv);
}

static int decode_batch_staged(ldpc_hip_bp *h, int osd, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                               double *llr, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    HIPCHK(hipSetDevice(h->device));
    const size_t B = (size_t)batch, m = (size_t)h->m, n = (size_t)h->n;
    const uint8_t *d_synd = synd;
    uint8_t *d_dec = decoding;
    double *d_llr = llr;
    int32_t *d_it = iters;
    uint8_t *d_cv = conv;
    int rc;
    const bool h_synd = !is_device_ptr(synd), h_dec = !is_device_ptr(decoding);
    const bool h_llr = llr && !is_device_ptr(llr), h_it = iters && !is_device_ptr(iters);
    const bool h_cv = conv && !is_device_ptr(conv);
    if (h_synd) {
        if ((rc = h->st_synd.ensure(B * m ? B * m : 1))) return rc;
        HIPCHK(hipMemcpyAsync(h->st_synd.p, synd, B * m, hipMemcpyHostToDevice, h->stream));
        d_synd = (const uint8_t *)h->st_synd.p;
    }
    if (h_dec) { if ((rc = h->st_dec.ensure(B * n ? B * n : 1))) return rc; d_dec = (uint8_t *)h->st_dec.p; }
    if (h_llr) { if ((rc = h->st_llr.ensure(B * n * 8 ? B * n * 8 : 1))) return rc; d_llr = (double *)h->st_llr.p; }
    if (h_it) { if ((rc = h->st_iters.ensure(B * 4))) return rc; d_it = (int32_t *)h->st_iters.p; }
    if (h_cv) { if ((rc = h->st_conv.ensure(B))) return rc; d_cv = (uint8_t *)h->st_conv.p; }

    if ((rc = osd >= 0 ? bposd_device(h, osd ? h->osd_method : 1, osd ? h->osd_order : 0, d_synd, batch, d_dec, d_llr, d_it, d_cv)
                       : decode_device(h, d_synd, batch, d_dec, d_llr, d_it, d_cv))) return rc;

    if (h_dec) HIPCHK(hipMemcpyAsync(decoding, d_dec, B * n, hipMemcpyDeviceToHost, h->stream));
    if (h_llr) HIPCHK(hipMemcpyAsync(llr, d_llr, B * n * 8, hipMemcpyDeviceToHost, h->stream));
    if (h_it) HIPCHK(hipMemcpyAsync(iters, d_it, B * 4, hipMemcpyDeviceToHost, h->stream));
    if (h_cv) HIPCHK(hipMemcpyAsync(conv, d_cv, B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LDPC_HIP_OK;
}

int ldpc_hip_gf2_mulvec_batch(ldpc_hip_bp *h, const uint8_t *vectors, int64_t batch, uint8_t *out) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0 || h->m == 0) return LDPC_HIP_OK;
    if (!vectors || !out) return fail(LDPC_HIP_ERR_INVALID, "null buffer");
    HIPCHK(hipSetDevice(h->device));
    const size_t B = (size_t)batch, m = (size_t)h->m, n = (size_t)h->n;
    const uint8_t *d_in = vectors;
    uint8_t *d_out = out;
    int rc;
    const bool h_in = !is_device_ptr(vectors), h_out = !is_device_ptr(out);
    if (h_in) {
        if ((rc = h->st_misc.ensure(B * n ? B * n : 1))) return rc;
        HIPCHK(hipMemcpyAsync(h->st_misc.p, vectors, B * n, hipMemcpyHostToDevice, h->stream));
        d_in = (const uint8_t *)h->st_misc.p;
    }
    if (h_out) { if ((rc = h->st_synd.ensure(B * m))) return rc; d_out = (uint8_t *)h->st_synd.p; }
    const int64_t total = batch * h->m;
    hipLaunchKernelGGL(gf2_mulvec_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream,
                       h->d_row_ptr, h->d_col_idx, h->m, h->n, d_in, batch, d_out);
    HIPCHK(hipGetLastError());
    if (h_out) HIPCHK(hipMemcpyAsync(out, d_out, B * m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_soft_info_decode_batch(ldpc_hip_bp *h, const double *soft_syndromes, int64_t batch, double cutoff, double sigma,
                                       uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv, double *soft_syndromes_out) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!soft_syndromes || !decoding) return fail(LDPC_HIP_ERR_INVALID, "soft syndromes and decoding must not be NULL");
    if (!(sigma > 0)) return fail(LDPC_HIP_ERR_INVALID, "The sigma value must be a float greater than 0.");  // _bp_decoder.pyx:748-749
    HIPCHK(hipSetDevice(h->device));
    const size_t B = (size_t)batch, m = (size_t)h->m, n = (size_t)h->n;
    int rc;
    const double *d_soft = soft_syndromes;
    uint8_t *d_dec = decoding;
    double *d_llr = llr, *d_so = soft_syndromes_out;
    int32_t *d_it = iters;
    uint8_t *d_cv = conv;
    const bool h_soft = !is_device_ptr(soft_syndromes), h_dec = !is_device_ptr(decoding);
    const bool h_llr = llr && !is_device_ptr(llr), h_it = iters && !is_device_ptr(iters), h_cv = conv && !is_device_ptr(conv);
    const bool h_so = soft_syndromes_out && !is_device_ptr(soft_syndromes_out);
    if (h_soft) {
        if ((rc = h->soft_in.ensure(B * m * 8 ? B * m * 8 : 1))) return rc;
        HIPCHK(hipMemcpyAsync(h->soft_in.p, soft_syndromes, B * m * 8, hipMemcpyHostToDevice, h->stream));
        d_soft = (const double *)h->soft_in.p;
    }
    if (h_dec) { if ((rc = h->st_dec.ensure(B * n ? B * n : 1))) return rc; d_dec = (uint8_t *)h->st_dec.p; }
    if (h_llr) { if ((rc = h->st_llr.ensure(B * n * 8 ? B * n * 8 : 1))) return rc; d_llr = (double *)h->st_llr.p; }
    if (h_it) { if ((rc = h->st_iters.ensure(B * 4))) return rc; d_it = (int32_t *)h->st_iters.p; }
    if (h_cv) { if ((rc = h->st_conv.ensure(B))) return rc; d_cv = (uint8_t *)h->st_conv.p; }
    if (h_so) { if ((rc = h->soft_out.ensure(B * m * 8 ? B * m * 8 : 1))) return rc; d_so = (double *)h->soft_out.p; }
    if ((rc = soft_info_device(h, d_soft, batch, cutoff, sigma, d_dec, d_llr, d_it, d_cv, d_so))) return rc;
    if (h_dec) HIPCHK(hipMemcpyAsync(decoding, d_dec, B * n, hipMemcpyDeviceToHost, h->stream));
    if (h_llr) HIPCHK(hipMemcpyAsync(llr, d_llr, B * n * 8, hipMemcpyDeviceToHost, h->stream));
    if (h_it) HIPCHK(hipMemcpyAsync(iters, d_it, B * 4, hipMemcpyDeviceToHost, h->stream));
    if (h_cv) HIPCHK(hipMemcpyAsync(conv, d_cv, B, hipMemcpyDeviceToHost, h->stream));
    if (h_so) HIPCHK(hipMemcpyAsync(soft_syndromes_out, d_so, B * m * 8, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_observables(ldpc_hip_bp *h, int32_t k, const int32_t *csr_row_ptr, const int32_t *csr_col_idx) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (k < 0 || !csr_row_ptr) return fail(LDPC_HIP_ERR_INVALID, "observables matrix: k < 0 or null row pointer");
    if (csr_row_ptr[0] != 0) return fail(LDPC_HIP_ERR_INVALID, "observables matrix: csr_row_ptr[0] must be 0");
    for (int i = 0; i < k; ++i)
        if (csr_row_ptr[i + 1] < csr_row_ptr[i]) return fail(LDPC_HIP_ERR_INVALID, "observables matrix: csr_row_ptr must not decrease");
    const int32_t nnz = csr_row_ptr[k];
    if (nnz > 0 && !csr_col_idx) return fail(LDPC_HIP_ERR_INVALID, "observables matrix: null column indices");
    for (int e = 0; e < nnz; ++e)
        if (csr_col_idx[e] < 0 || csr_col_idx[e] >= h->n) return fail(LDPC_HIP_ERR_INVALID, "observables matrix: column index out of range");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    int rc;
    if ((rc = h->obs_row_ptr.ensure(sizeof(int32_t) * (size_t)(k + 1)))) return rc;
    if ((rc = h->obs_col_idx.ensure(sizeof(int32_t) * (size_t)(nnz ? nnz : 1)))) return rc;
    HIPCHK(hipMemcpy(h->obs_row_ptr.p, csr_row_ptr, sizeof(int32_t) * (size_t)(k + 1), hipMemcpyHostToDevice));
    if (nnz) HIPCHK(hipMemcpy(h->obs_col_idx.p, csr_col_idx, sizeof(int32_t) * (size_t)nnz, hipMemcpyHostToDevice));
    h->obs_k = k;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_decode_b8(ldpc_hip_bp *h, const uint8_t *dets_b8, int64_t batch, int32_t with_osd, uint8_t *obs_b8,
                          uint8_t *decoding_b8, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!dets_b8) return fail(LDPC_HIP_ERR_INVALID, "null detection-event buffer");
    if (!obs_b8 && !decoding_b8) return fail(LDPC_HIP_ERR_INVALID, "neither obs_b8 nor decoding_b8 requested");
    if (obs_b8 && h->obs_k < 0) return fail(LDPC_HIP_ERR_INVALID, "obs_b8 requested but ldpc_hip_bp_set_observables was never called");
    HIPCHK(hipSetDevice(h->device));
    const size_t B = (size_t)batch, m = (size_t)h->m, n = (size_t)h->n;
    const size_t mb = (m + 7) / 8, nb = (n + 7) / 8, kb = obs_b8 ? ((size_t)h->obs_k + 7) / 8 : 0;
    int rc;
    const uint8_t *d_in = dets_b8;
    if (!is_device_ptr(dets_b8)) {
        if ((rc = h->b8_in.ensure(B * mb ? B * mb : 1))) return rc;
        HIPCHK(hipMemcpyAsync(h->b8_in.p, dets_b8, B * mb, hipMemcpyHostToDevice, h->stream));
        d_in = (const uint8_t *)h->b8_in.p;
    }
    if ((rc = h->b8_synd.ensure(B * m ? B * m : 1))) return rc;
    if ((rc = h->b8_dec.ensure(B * n ? B * n : 1))) return rc;
    uint8_t *d_synd = (uint8_t *)h->b8_synd.p, *d_dec = (uint8_t *)h->b8_dec.p;
    if (m) hipLaunchKernelGGL(unpack_b8_kernel, dim3((unsigned)((B * m + 255) / 256)), dim3(256), 0, h->stream, d_in, batch, h->m, d_synd);
    HIPCHK(hipGetLastError());
    const bool h_it = iters && !is_device_ptr(iters), h_cv = conv && !is_device_ptr(conv);
    int32_t *d_it = iters;
    uint8_t *d_cv = conv;
    if (h_it) { if ((rc = h->st_iters.ensure(B * 4))) return rc; d_it = (int32_t *)h->st_iters.p; }
    if (h_cv) { if ((rc = h->st_conv.ensure(B))) return rc; d_cv = (uint8_t *)h->st_conv.p; }
    if ((rc = with_osd ? bposd_device(h, h->osd_method, h->osd_order, d_synd, batch, d_dec, nullptr, d_it, d_cv)
                       : decode_device(h, d_synd, batch, d_dec, nullptr, d_it, d_cv))) return rc;
    hipLaunchKernelGGL(zero_shot_shortcut_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, h->stream, d_in, batch, h->m, h->n,
                       d_dec, d_it, d_cv);
    size_t off = 0;
    if ((rc = h->b8_out.ensure(B * (kb + nb) ? B * (kb + nb) : 1))) return rc;
    uint8_t *d_obs = obs_b8, *d_dec8 = decoding_b8;
    const bool h_obs = obs_b8 && !is_device_ptr(obs_b8), h_dec8 = decoding_b8 && !is_device_ptr(decoding_b8);
    if (h_obs) { d_obs = (uint8_t *)h->b8_out.p; off = B * kb; }
    if (h_dec8) d_dec8 = (uint8_t *)h->b8_out.p + off;
    if (obs_b8 && kb)
        hipLaunchKernelGGL(observables_b8_kernel, dim3((unsigned)((B * kb + 255) / 256)), dim3(256), 0, h->stream,
                           (const int32_t *)h->obs_row_ptr.p, (const int32_t *)h->obs_col_idx.p, h->obs_k, h->n, d_dec, batch, d_obs);
    if (decoding_b8 && nb)
        hipLaunchKernelGGL(pack_b8_kernel, dim3((unsigned)((B * nb + 255) / 256)), dim3(256), 0, h->stream, d_dec, batch, h->n, d_dec8);
    HIPCHK(hipGetLastError());
    if (h_obs && kb) HIPCHK(hipMemcpyAsync(obs_b8, d_obs, B * kb, hipMemcpyDeviceToHost, h->stream));
    if (h_dec8 && nb) HIPCHK(hipMemcpyAsync(decoding_b8, d_dec8, B * nb, hipMemcpyDeviceToHost, h->stream));
    if (h_it) HIPCHK(hipMemcpyAsync(iters, d_it, B * 4, hipMemcpyDeviceToHost, h->stream));
    if (h_cv) HIPCHK(hipMemcpyAsync(conv, d_cv, B, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LDPC_HIP_OK;
}

int ldpc_hip_gen_bsc_syndromes(ldpc_hip_bp *h, uint64_t seed, uint64_t threshold, int64_t shot0,
                               int64_t batch, uint8_t *syndromes, uint8_t *errors) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0 || shot0 < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch or shot0");
    if (batch == 0) return LDPC_HIP_OK;
    if (!syndromes) return fail(LDPC_HIP_ERR_INVALID, "null syndromes buffer");
    HIPCHK(hipSetDevice(h->device));
    const size_t B = (size_t)batch, m = (size_t)h->m, n = (size_t)h->n;
    uint8_t *d_s = syndromes, *d_e = errors;
    int rc;
    const bool h_s = !is_device_ptr(syndromes), h_e = errors && !is_device_ptr(errors);
    if (h_s) { if ((rc = h->st_synd.ensure(B * m ? B * m : 1))) return rc; d_s = (uint8_t *)h->st_synd.p; }
    if (h_e) { if ((rc = h->st_misc.ensure(B * n ? B * n : 1))) return rc; d_e = (uint8_t *)h->st_misc.p; }
    if (h->m > 0) {
        const int64_t total = batch * h->m;
        hipLaunchKernelGGL(gen_bsc_syndromes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           h->stream, h->d_row_ptr, h->d_col_idx, h->m, h->n, seed, threshold, shot0,
                           batch, d_s);
    }
    if (errors && h->n > 0) {
        const int64_t total = batch * h->n;
        hipLaunchKernelGGL(gen_bsc_errors_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           h->stream, h->n, seed, threshold, shot0, batch, d_e);
    }
    HIPCHK(hipGetLastError());
    if (h_s) HIPCHK(hipMemcpyAsync(syndromes, d_s, B * m, hipMemcpyDeviceToHost, h->stream));
    if (h_e) HIPCHK(hipMemcpyAsync(errors, d_e, B * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LDPC_HIP_OK;
}

}  // extern "C"
