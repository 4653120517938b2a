// bp_serial_stream_kernel.h -- the serial schedule (bp.hpp:451-545) streamed from HBM at the rate of the flooding kernel
// Part of libldpc_hip.so (translation unit tu_serial.hip).
#pragma once

#include "bp_device_common.h"
#include "bp_serial_kernels.h"

// The level-parallel form of the serial schedule (bp_serial_level_kernel) for codes far beyond LDS -- the (3,6) n = 10 000 code has 35
// levels of ~286 check-disjoint bits -- with the message traffic organised like the flooding kernel's (bp_stream_kernel.h):
//
//   * a bit update reads the OTHER entries of the bit's DC check rows, DC (DR - 1) message segments of 512 bytes, and writes the bit's DC
//     own ones; nothing else moves (the check->bit messages never leave the registers): 18 segments per bit on a (6,3) code, i.e.
//     1.5 x the flooding schedule's 4 per edge and iteration -- for half as many iterations;
//   * the schedule is static, so the host lays it out as one RECORD per position of the level-major order: the edge numbers of the
//     segments to fetch, of the segments to write, and the bit -- two scalar-cache lines, read with two scalar loads one and two steps
//     ahead of their use, no index arithmetic on the vector unit;
//   * every wavefront keeps RING positions' segments in flight into a private LDS ring with `buffer_load_dwordx4 ... lds` (two
//     arbitrary segments per instruction: lanes 0-31 fetch one, lanes 32-63 the other; an odd last segment pairs with an address
//     beyond the buffer, which the range check turns into zeros without a memory access) and waits with counted `s_waitcnt vmcnt(N)`;
//     positions of one level touch disjoint rows, so a wavefront runs ahead freely inside a level; a level ends with one workgroup
//     barrier (the next level reads what this one wrote);
//   * the first iteration needs no initial messages in memory: an entry of a row that no earlier position of the schedule has written
//     still holds its initial value tanh(llr0 / 2) | llr0, which is the same in all 64 lanes -- the record carries a mask of the
//     entries already written, the others are taken from a table of initial values per position (SerialArgs::pos_e0) through the
//     scalar cache, their segments are neither written beforehand nor fetched (a sixth of an iteration's traffic for the writes, on average half
//     of the first iteration's reads);
//   * a pass can start from the state an earlier pass left (SerialArgs::it_start > 0: lanes compacted out of the tiles of a first pass).
//
// Same operations on the same operands in the same order as bp_serial_kernel's walk (levels: bits that share no check commute), hence
// the same bits.  Matrices with a single row weight DR and a single column weight DC only (the host side checks); everything else keeps
// bp_serial_level_kernel.
//
// Record of a position, int32[32], 128-byte aligned:
//   [0, NO)          CSR edge numbers of the other entries of the bit's rows, row by row (rows ascending, entries ascending): NO = DC (DR - 1) <= 15
//   [15]             mask: bit t set = entry t of [0, NO) has been written by an earlier position of the schedule (first iteration)
//   [16, 16 + DC)    the bit's own edges, rows ascending
//   [16 + DC]        the bit
//   [16 + DC + 1]    the mask once more (the line the bit update reads)
constexpr int SERIAL_STREAM_REC = 32;
constexpr int serial_stream_slot_bytes(int dr, int dc) { return (dc * (dr - 1) + 1) / 2 * 1024; }

#ifdef LDPC_SER_PROF  // measurement build (tools/serial_stream_phases.py): where a wavefront's cycles go, summed over all wavefronts
__device__ unsigned long long ser_phase_clocks[8];
__device__ unsigned long long ser_wg_trace[4096][4];  // per workgroup: start / end (constant-rate ticks), HW_ID, XCC_ID
#define SER_PROF_DECL unsigned long long prof_t_ = __builtin_readcyclecounter(), prof_acc_[8] = {}; \
    if (threadIdx.x == 0 && blockIdx.x < 4096) { unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
        ser_wg_trace[blockIdx.x][0] = __builtin_readsteadycounter(); ser_wg_trace[blockIdx.x][2] = hw_; ser_wg_trace[blockIdx.x][3] = xcc_; }
#define SER_PROF(slot) do { const unsigned long long now_ = __builtin_readcyclecounter(); prof_acc_[slot] += now_ - prof_t_; prof_t_ = now_; } while (0)
#define SER_PROF_FLUSH do { if (lane == 0) for (int q_ = 0; q_ < 8; ++q_) atomicAdd(&ser_phase_clocks[q_], prof_acc_[q_]); if (threadIdx.x == 0 && blockIdx.x < 4096) ser_wg_trace[blockIdx.x][1] = __builtin_readsteadycounter(); } while (0)
#else
#define SER_PROF_DECL
#define SER_PROF(slot)
#define SER_PROF_FLUSH
#endif

typedef int ldpc_v16i __attribute__((ext_vector_type(16)));
typedef int ldpc_v8i __attribute__((ext_vector_type(8)));
typedef double ldpc_v8d __attribute__((ext_vector_type(8)));

// [n] what an edge of column j holds before the first iteration: tanh(llr0[j] / 2) | llr0[j]
template <int METHOD, int MATH>
__global__ void __launch_bounds__(256) serial_edge0_kernel(const double *llr0, int n, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = edge_form<METHOD, MATH>(llr0[j]);
}
// [positions][16] the initial values of the other entries of every position (what the first iteration takes instead of an entry nobody has written yet)
__global__ void __launch_bounds__(256) serial_pos_e0_kernel(const int32_t *__restrict__ pos_tab, const int32_t *__restrict__ col_idx, const double *__restrict__ edge0,
                                                            int positions, int no, double *__restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= positions * 16) return;
    const int p = q >> 4, t = q & 15;
    out[q] = t < no ? edge0[col_idx[pos_tab[(size_t)p * SERIAL_STREAM_REC + t]]] : 0.0;
}

// The DC check->bit messages of one bit, product-sum with the libm-exact `log`: the fast path of the flooding kernel's check row
// (check_row_ps_exact_fast, bp_device_common.h) for the serial schedule's bit update.  Every lane evaluates the TABLE branch of the `log`
// for each of its messages; lanes whose argument is near 1 park it in a wave-private LDS buffer, compacted over the bit's DC messages,
// and the near-1 branch runs once per 64 parked arguments (usually once per bit instead of DC times).  Preconditions, tested once per
// bit over the LIVE lanes: every tanh value that enters has magnitude below 1 and is no NaN -- then each product x has |x| < 1 and
// q = (1 + x) / (1 - x) is a normal number (no 0 / inf / NaN patches).  Same operations on the same operands: same bits.  false: take
// the generic routine.
template <int DC, int DRM1>
__device__ __forceinline__ bool bit_messages_ps_exact_fast(const double (&v)[DC * DRM1], const bool (&odd)[DC], double (&c)[DC], const double *log_tab,
                                                           uint64_t live, double *near_buf) {
    bool bad = false;
#pragma unroll
    for (int t = 0; t < DC * DRM1; ++t) bad = bad || !(__builtin_fabs(v[t]) < 1.0);
    if (__builtin_amdgcn_ballot_w64(bad) & live) return false;
    const int lane = (int)(threadIdx.x & (LDPC_WAVE - 1));
    const bool lane_live = (live >> lane) & 1ull;
    uint64_t parked[DC];
    int first[DC];
    int total = 0;
#pragma unroll
    for (int k = 0; k < DC; ++k) {
        double x = 1.0;  // bp.hpp:492-498: the product over the row's other entries, in the row's order
#pragma unroll
        for (int q = 0; q < DRM1; ++q) x *= v[k * DRM1 + q];
        const double q = ldpc_math::div_cr(1.0 + x, 1.0 - x);
        const bool near_any = ldpc_math::log_near_one(q);
        double y = ldpc_math::log_libm_general(q, log_tab);
        const uint64_t mask = __builtin_amdgcn_ballot_w64(near_any) & live;  // (dead lanes park nothing)
        const bool near = near_any && lane_live;
        parked[k] = 0;
        first[k] = 0;
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
        c[k] = y;
        LDPC_EDGE_FENCE();
    }
    if (total) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // LDS operations of a wavefront execute in order; this only pins the compiler
        for (int s = lane; s < total; s += LDPC_WAVE) near_buf[s] = ldpc_math::log_libm_near_one(near_buf[s]);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < DC; ++k)
            if (parked[k]) {
                if ((parked[k] >> lane) & 1ull) c[k] = near_buf[first[k] + lane_rank(parked[k])];
            }
    }
#pragma unroll
    for (int k = 0; k < DC; ++k) c[k] = ldpc_math::as_f64(ldpc_math::as_u64(c[k]) ^ (odd[k] ? 0x8000000000000000ull : 0ull));  // pow(-1, syndrome byte) (bp.hpp:499)
    return true;
}

// Scalar data is fetched one and two steps AHEAD of the step that uses it (loop-carried in SGPRs): a record line is a scalar-cache miss
// by construction (1.3 MB streamed once per iteration), the syndrome word and the prior hang off it, and a wavefront that asked for them
// where it needs them stood still for three dependent round trips per bit.
template <int METHOD, int MATH, int DR, int DC, int RING>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 5))) bp_serial_stream_kernel(const SerialArgs a) {
    constexpr int NO = DC * (DR - 1);
    static_assert(NO <= 15 && DC <= 4, "one 16-entry record line of other entries (+ mask), one line of own edges");
    constexpr int REC = SERIAL_STREAM_REC;
    constexpr int NDMA = (NO + 1) / 2;
    constexpr int SLOT_BYTES = NDMA * 1024;
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
    const unsigned ring_addr = (unsigned)(uintptr_t)ldpc_dyn_lds + (unsigned)wave * (RING * SLOT_BYTES);
    const double *ringp = reinterpret_cast<const double *>(ldpc_dyn_lds + (size_t)wave * (RING * SLOT_BYTES));
    // parking space of the exact product-sum messages (bit_messages_ps_exact_fast): behind the rings, LDPC_NEAR_BYTES per wavefront
    double *near_buf = reinterpret_cast<double *>(ldpc_dyn_lds + (size_t)nwaves * (RING * SLOT_BYTES) + (size_t)wave * LDPC_NEAR_BYTES);
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
    const bool implicit_init = a.edge0 != nullptr && a.it_start == 0;
    if (!implicit_init && a.it_start == 0)
        for (int e = wave; e < nnz; e += nwaves) At.st(l8, e, edge_form<METHOD, MATH>(sload(a.llr0 + sload(a.col_idx + e))));
    __syncthreads();
    SER_PROF_DECL;

    for (int it = a.it_start + 1; it <= a.max_iter; ++it) {
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        const bool lane_live = !((done >> lane) & 1ull);

        // One level: this wavefront's positions p0 + wave, + nwaves, ...  FRESH = the first iteration of an implicitly initialised decode.
        auto run_level = [&](auto fresh_tag, const int p0, const int p1) {
            constexpr bool FRESH = decltype(fresh_tag)::value;
            const int mine = p1 - p0 - wave;
            const int nsteps = mine > 0 ? (mine + nwaves - 1) / nwaves : 0;
            if (nsteps == 0) return;
            const int plast = p0 + wave + (nsteps - 1) * nwaves;
            auto rec_of = [&](int idx) {  // (steps beyond the last one read the last one's record again: harmless, and no branch)
                const int p = p0 + wave + idx * nwaves;
                return a.pos_tab + (size_t)(p < plast ? p : plast) * REC;
            };
            // the segments of a position into a ring slot: NDMA instructions whatever the mask says (the counted waits need a fixed number;
            // an entry nobody has written yet gets the address beyond the buffer: zeros, no memory access)
            auto issue = [&](const ldpc_v16i &oth, int slot) {
                const unsigned written = FRESH ? (unsigned)oth[15] : ~0u;
#pragma unroll
                for (int c = 0; c < NDMA; ++c) {
                    const unsigned ea = (!FRESH || ((written >> (2 * c)) & 1u)) ? (unsigned)oth[2 * c] << 9 : beyond;
                    const unsigned eb = (2 * c + 1 < NO && (!FRESH || ((written >> (2 * c + 1)) & 1u))) ? (unsigned)oth[2 * c + 1] << 9 : beyond;
                    lds_dma16(At.rsrc, (upper ? eb : ea) + l16, 0u, ring_addr + slot * SLOT_BYTES + c * 1024);
                }
            };
            struct Node {  // what a bit update needs besides the message segments, all wave-uniform
                ldpc_v8i own;          // own edges, the bit, the mask
                uint64_t par[DC];      // the syndrome words of its checks
                double llr0;
                ldpc_v8d e0a, e0b;     // FRESH: initial values of the other entries
            };
            auto load_own = [&](Node &nd, int idx) { nd.own = sload(reinterpret_cast<const ldpc_v8i *>(rec_of(idx) + 16)); };
            auto load_rest = [&](Node &nd, int idx) {  // (needs nd.own)
#pragma unroll
                for (int k = 0; k < DC; ++k) nd.par[k] = sload(par + nd.own[k] / DR);  // (every row has DR entries: row i starts at edge i DR)
                nd.llr0 = sload(a.llr0 + nd.own[DC]);
                if (FRESH) {
                    const int p = p0 + wave + idx * nwaves;
                    const double *e0 = a.pos_e0 + (size_t)(p < plast ? p : plast) * 16;
                    nd.e0a = sload(reinterpret_cast<const ldpc_v8d *>(e0));
                    nd.e0b = sload(reinterpret_cast<const ldpc_v8d *>(e0 + 8));
                }
            };
            ldpc_v16i oth;
#pragma unroll
            for (int r = 0; r < RING; ++r)
                if (r < nsteps) {
                    oth = sload(reinterpret_cast<const ldpc_v16i *>(rec_of(r)));
                    issue(oth, r);
                }
            oth = sload(reinterpret_cast<const ldpc_v16i *>(rec_of(RING)));
            Node cur, nxt;
            load_own(cur, 0);
            load_own(nxt, 1);
            load_rest(cur, 0);
            int slot = 0;
            SER_PROF(0);  // level prologue (issue, records)
            for (int idx = 0; idx < nsteps; ++idx) {
                // behind the wanted loads sit, per position issued since, NDMA loads and DC + 1 (+ 1 with log-ratios) stores
                if (idx >= RING && idx + RING - 1 < nsteps) {
                    if (want_llr) wait_vmcnt<RING * (DC + 2) + (RING - 1) * NDMA>(); else wait_vmcnt<RING * (DC + 1) + (RING - 1) * NDMA>();
                } else {
                    wait_vmcnt<0>();
                }
                SER_PROF(1);  // waiting for the segments
                double v[NO];
#pragma unroll
                for (int t = 0; t < NO; ++t) v[t] = ringp[slot * (SLOT_BYTES / 8) + t * LDPC_WAVE + lane];
                wait_lds_reads();  // the slot is free once its values sit in registers (and last step's scalar loads have landed)
                SER_PROF(2);  // LDS reads (+ last step's scalar loads)
                if (idx + RING < nsteps) issue(oth, slot);
                SER_PROF(3);  // issue
                if (FRESH) {  // entries no earlier position has written hold their initial value (the same in all lanes)
                    const unsigned written = (unsigned)cur.own[DC + 1];
#pragma unroll
                    for (int t = 0; t < NO; ++t)
                        if (!((written >> t) & 1u)) v[t] = t < 8 ? cur.e0a[t & 7] : cur.e0b[t & 7];
                }
                const int bit = cur.own[DC];
                double llr = cur.llr0;  // bp.hpp:488
                double c[DC], pre[DC];
                bool odd_k[DC];
#pragma unroll
                for (int k = 0; k < DC; ++k) odd_k[k] = (cur.par[k] >> lane) & 1ull;  // pow(-1, syndrome byte) / syndrome parity
                bool have_c = false;
                if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0) have_c = bit_messages_ps_exact_fast<DC, DR - 1>(v, odd_k, c, log_tab, ~done, near_buf);
#pragma unroll
                for (int k = 0; k < DC; ++k) {
                    const bool odd = odd_k[k];
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) {
                        if (!have_c) {  // (wave-uniform)
                            double x = 1.0;  // bp.hpp:492-498: the product over the row's other entries, in the row's order
#pragma unroll
                            for (int q = 0; q < DR - 1; ++q) x *= v[k * (DR - 1) + q];
                            c[k] = ps_message<MATH>(x, odd, log_tab);
                        }
                    } else {
                        int sgn = odd ? 1 : 0;  // bp.hpp:505-519
                        double temp = DBL_MAX;
#pragma unroll
                        for (int q = 0; q < DR - 1; ++q) {
                            const double b = v[k * (DR - 1) + q];
                            const double ab = fabs(b);
                            if (ab < temp) temp = ab;
                            if (b <= 0) sgn ^= 1;
                        }
                        c[k] = (alpha * (sgn ? -1.0 : 1.0)) * temp;
                    }
                    pre[k] = llr;  // bp.hpp:500-501 / 520-521
                    llr += c[k];
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) LDPC_EDGE_FENCE();
                }
                double out[DC];
                double temp = 0.0;  // bp.hpp:530-534
#pragma unroll
                for (int k = DC - 1; k >= 0; --k) {
                    out[k] = edge_form<METHOD, MATH>(pre[k] + temp);
                    temp += c[k];
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) LDPC_EDGE_FENCE();
                }
                // the scalar data of the steps to come, asked for now and looked at one step later: in flight across the stores and the
                // wait for the next position's segments
                SER_PROF(4);  // arithmetic
                Node nn;
                oth = sload(reinterpret_cast<const ldpc_v16i *>(rec_of(idx + 1 + RING)));
                load_own(nn, idx + 2);
                load_rest(nxt, idx + 1);
#pragma unroll
                for (int k = DC - 1; k >= 0; --k) At.st(l8, cur.own[k], out[k]);
                const uint64_t hard = __ballot(llr <= 0);  // bp.hpp:525-529
                if (lane == 0) dcur[bit] = hard;
                if (want_llr && lane_live) Lt.st(l8, bit, llr);
                SER_PROF(5);  // scalar loads asked for, stores issued
                cur = nxt;
                nxt.own = nn.own;
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
            wait_vmcnt<0>();
            SER_PROF(6);  // drain
        };

        const bool fresh = implicit_init && it == 1;
        for (int l = 0; l < a.n_levels; ++l) {
            const int p0 = sload(a.lvl_ptr + l), p1 = sload(a.lvl_ptr + l + 1);
            if (fresh) run_level(std::true_type{}, p0, p1);
            else run_level(std::false_type{}, p0, p1);
            __syncthreads();  // the next level reads what this one wrote
            SER_PROF(7);  // level barrier
        }
        // candidate syndrome of this iteration's hard decision vs the syndrome bytes (bp.hpp:537-543)
        uint64_t unsat = 0;
        for (int i = threadIdx.x; i < m; i += blockDim.x) {
            uint64_t cand = 0;
#pragma unroll
            for (int q = 0; q < DR; ++q) cand ^= dcur[a.col_idx[i * DR + q]];
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
    SER_PROF_FLUSH;
    if (threadIdx.x == 0) clock_probe_end(a.clk, clk_stamp);
}

// ---- the same schedule for a HANDFUL of syndromes: one workgroup per syndrome, lane = bit ---------------------------------------------
// A 64-syndrome tile moves 64 lanes' segments whatever the number of lanes still decoding, and its levels are a chain of ~35 barriers per
// iteration on ONE compute unit: the one hopeless syndrome of a batch (1 in 65 536 at the headline's early-exit point) kept its tile
// going for 46 more iterations of ~2 ms while the rest of the chip had nothing to do -- as long as the whole first pass.  Few syndromes
// are decoded here instead: a syndrome's messages as one row-major array [nnz] (240 KB on the n = 10 000 code: it lives in L2), the
// bits of a level one per LANE (a level's ~286 bits are one step of a 512-thread workgroup), the same position records, the same
// operations on the same operands in the same order -- the same bits (how the lanes share a bit: at the kernel).  Used for the rows a streamed pass left (decode_serial, host_serial.h:
// state taken over lane by lane through serial_rows_from_tiles_kernel) and for small batches from the start (it_start = 0).
struct SerialLaneArgs {
    int32_t m, n, nnz, max_iter, it_start, n_levels;
    double ms_scaling_factor;
    int64_t rows;
    const int32_t *col_idx, *lvl_ptr, *pos_tab;
    const double *llr0;
    double *A;            // [rows][nnz] tanh(b2c / 2) | b2c, row-major per syndrome
    const uint8_t *synd;  // [rows][m]
    uint8_t *decoding;    // [rows][n]
    double *llr;          // [rows][n] or nullptr
    int32_t *iters;
    uint8_t *conv;
};

// A_rows[r][e] = A_tiles[list[r] / 64][e][list[r] % 64]  (list == nullptr: row r itself)
__global__ void __launch_bounds__(256) serial_rows_from_tiles_kernel(const double *__restrict__ tiles, const int32_t *__restrict__ list, int64_t rows, int nnz,
                                                                     double *__restrict__ out) {
    const int64_t r = blockIdx.y;
    const int64_t b = list ? (int64_t)list[r] : r;
    const double *from = tiles + ((b >> 6) * (int64_t)nnz) * 64 + (b & 63);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += gridDim.x * blockDim.x) out[r * (int64_t)nnz + e] = from[(int64_t)e * 64];
}

// lane = (bit of the level, one of its DC checks): DC neighbouring lanes share a bit.  Each forms ITS check's message (the product or
// minimum over the row's other entries, the `log`), the DC messages go round the group by lane permutation, every lane of the group adds
// them up in the reference's order (bp.hpp:500-501, 530-534 -- a handful of additions, done DC times over rather than waited for) and
// evaluates its own edge's new bit->check message (the `tanh`): the dependent chain of a level is one `log` and one `tanh` long instead of DC of each.
template <int METHOD, int MATH, int DR, int DC>
__global__ void __launch_bounds__(1024) bp_serial_lane_kernel(const SerialLaneArgs a) {
    constexpr int REC = SERIAL_STREAM_REC;
    constexpr int PER_WAVE = LDPC_WAVE / DC;  // bits per wavefront step (lanes beyond PER_WAVE * DC idle)
    extern __shared__ __attribute__((aligned(16))) unsigned char lane_lds[];
    uint8_t *dbit = lane_lds;  // [n] this iteration's hard decisions
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwaves = T >> 6;
    const int grp = lane / DC, k = lane - grp * DC;  // this lane's bit within the wavefront's step, and which of the bit's checks
    const bool active = grp < PER_WAVE;
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
    bool converged = false;
    int it_done = a.max_iter;
    for (int it = a.it_start + 1; it <= a.max_iter; ++it) {
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        for (int l = 0; l < a.n_levels; ++l) {
            const int p0 = a.lvl_ptr[l], p1 = a.lvl_ptr[l + 1];
            for (int pb = p0 + wave * PER_WAVE; pb < p1; pb += nwaves * PER_WAVE) {  // (wave-uniform trip count: the permutations below need whole wavefronts)
                const int p = pb + grp;
                const bool on = active && p < p1;
                const int32_t *rec = a.pos_tab + (size_t)(on ? p : p0) * REC;
                const int own = rec[16 + k], bit = rec[16 + DC];
                double c;
                {
                    double v[DR - 1];
#pragma unroll
                    for (int q = 0; q < DR - 1; ++q) v[q] = A[rec[k * (DR - 1) + q]];
                    const bool odd = synd[own / DR] & 1;  // pow(-1, syndrome byte) / syndrome parity
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) {
                        double x = 1.0;  // bp.hpp:492-498
#pragma unroll
                        for (int q = 0; q < DR - 1; ++q) x *= v[q];
                        c = ps_message<MATH>(x, odd, log_tab);
                    } else {
                        int sgn = odd ? 1 : 0;  // bp.hpp:505-519
                        double temp = DBL_MAX;
#pragma unroll
                        for (int q = 0; q < DR - 1; ++q) {
                            const double ab = fabs(v[q]);
                            if (ab < temp) temp = ab;
                            if (v[q] <= 0) sgn ^= 1;
                        }
                        c = (alpha * (sgn ? -1.0 : 1.0)) * temp;
                    }
                }
                double cs[DC];
#pragma unroll
                for (int j = 0; j < DC; ++j) cs[j] = __shfl(c, grp * DC + j, LDPC_WAVE);
                double llr = a.llr0[bit], pre = 0.0;  // bp.hpp:488, 500-501
#pragma unroll
                for (int j = 0; j < DC; ++j) {
                    if (j == k) pre = llr;
                    llr += cs[j];
                }
                double temp = 0.0;  // bp.hpp:530-534: what the entries after this one add
#pragma unroll
                for (int j = DC - 1; j >= 0; --j)
                    if (j > k) temp += cs[j];
                const double out = edge_form<METHOD, MATH>(pre + temp);
                if (on) {
                    A[own] = out;
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
#pragma unroll
            for (int q = 0; q < DR; ++q) cand ^= dbit[a.col_idx[i * DR + q]];
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
