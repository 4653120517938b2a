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
//   * the schedule is static, so the host lays it out as one RECORD per position of the level-major order (serial_stream_record):
//     the edge numbers of the segments to fetch, of the segments to write, and the bit -- one scalar-cache line and a bit, read with
//     two scalar loads, no index arithmetic on the vector unit;
//   * every wavefront keeps RING positions' segments in flight into a private LDS ring with `buffer_load_dwordx4 ... lds` (two
//     arbitrary segments per instruction: lanes 0-31 fetch one, lanes 32-63 the other; an odd last segment pairs with an address
//     beyond the buffer, which the range check turns into zeros without a memory access) and waits with counted `s_waitcnt vmcnt(N)`;
//     positions of one level touch disjoint rows, so a wavefront runs ahead freely inside a level; a level ends with one workgroup
//     barrier (the next level reads what this one wrote);
//   * the first iteration needs no initial messages in memory: an entry of a row that no earlier position of the schedule has written
//     still holds its initial value tanh(llr0 / 2) | llr0, which is the same in all 64 lanes -- the record carries a mask of the
//     entries already written, the others are taken from the table of initial values (SerialArgs::edge0) through the scalar cache,
//     their segments are neither written beforehand nor fetched (a sixth of an iteration's traffic for the writes, on average half
//     of the first iteration's reads);
//   * a pass can start from the state an earlier pass left (SerialArgs::it_start > 0: lanes compacted out of the tiles of a first pass).
//
// Same operations on the same operands in the same order as bp_serial_kernel's walk (levels: bits that share no check commute), hence
// the same bits.  Matrices with a single row weight DR and a single column weight DC only (the host side checks); everything else keeps
// bp_serial_level_kernel.
//
// Record of a position, int32[serial_stream_rec(DR, DC)], 64-byte aligned:
//   [0, NO)            CSR edge numbers of the other entries of the bit's rows, row by row (rows ascending, entries ascending): NO = DC (DR - 1)
//   [NO_PAD, +DC)      the bit's own edges, rows ascending           (NO_PAD = NO rounded up to a multiple of 16)
//   [NO_PAD + DC]      the bit
//   [NO_PAD + DC + 1]  mask: bit t set = entry t of [0, NO) has been written by an earlier position of the schedule (first iteration)
constexpr int serial_stream_no_pad(int dr, int dc) { return (dc * (dr - 1) + 15) / 16 * 16; }
constexpr int serial_stream_rec(int dr, int dc) { return (serial_stream_no_pad(dr, dc) + dc + 2 + 15) / 16 * 16; }
constexpr int serial_stream_slot_bytes(int dr, int dc) { return (dc * (dr - 1) + 1) / 2 * 1024; }

// [n] what an edge of column j holds before the first iteration: tanh(llr0[j] / 2) | llr0[j]
template <int METHOD, int MATH>
__global__ void __launch_bounds__(256) serial_edge0_kernel(const double *llr0, int n, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < n) out[j] = edge_form<METHOD, MATH>(llr0[j]);
}

typedef int ldpc_v16i __attribute__((ext_vector_type(16)));
typedef int ldpc_v4i __attribute__((ext_vector_type(4)));

template <int METHOD, int MATH, int DR, int DC, int RING>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 5))) bp_serial_stream_kernel(const SerialArgs a) {
    constexpr int NO = DC * (DR - 1);
    static_assert(NO <= 16 && DC <= 4, "one 16-entry record line of other entries, one 4-entry line of own edges");
    constexpr int NO_PAD = serial_stream_no_pad(DR, DC);
    constexpr int REC = serial_stream_rec(DR, DC);
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
    const unsigned l16 = (unsigned)(lane & 31) * 16u;
    const bool upper = lane >= 32;
    const unsigned beyond = (unsigned)nnz << 9;  // an offset the buffer's range check rejects: zeros, no memory access

    const int64_t valid = a.batch - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    const uint64_t never = a.invalid[tile];
    int my_iter = 0;
    const bool implicit_init = a.edge0 != nullptr && a.it_start == 0;
    if (!implicit_init && a.it_start == 0)
        for (int e = wave; e < nnz; e += nwaves) At.st(l8, e, edge_form<METHOD, MATH>(sload(a.llr0 + sload(a.col_idx + e))));
    __syncthreads();

    // the segments of position p into a ring slot; `fresh`: the first iteration of an implicitly initialised decode (entries nobody has
    // written yet are not fetched)
    auto issue = [&](int p, int slot, bool fresh) {
        const int32_t *rec = a.pos_tab + (size_t)p * REC;
        const ldpc_v16i oth = sload(reinterpret_cast<const ldpc_v16i *>(rec));
        unsigned written = ~0u;
        if (fresh) written = (unsigned)sload(rec + NO_PAD + DC + 1);
#pragma unroll
        for (int c = 0; c < NDMA; ++c) {
            const unsigned ea = ((written >> (2 * c)) & 1u) ? (unsigned)oth[2 * c] << 9 : beyond;
            const unsigned eb = (2 * c + 1 < NO && ((written >> (2 * c + 1)) & 1u)) ? (unsigned)oth[2 * c + 1] << 9 : beyond;
            if (fresh && ea == beyond && eb == beyond) continue;  // (wave-uniform; the counted waits are not used in that iteration)
            lds_dma16(At.rsrc, (upper ? eb : ea) + l16, 0u, ring_addr + slot * SLOT_BYTES + c * 1024);
        }
    };

    for (int it = a.it_start + 1; it <= a.max_iter; ++it) {
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        const bool lane_live = !((done >> lane) & 1ull);
        const bool fresh = implicit_init && it == 1;
        for (int l = 0; l < a.n_levels; ++l) {
            const int p0 = sload(a.lvl_ptr + l), p1 = sload(a.lvl_ptr + l + 1);
            const int mine = p1 - p0 - wave;
            const int nsteps = mine > 0 ? (mine + nwaves - 1) / nwaves : 0;
#pragma unroll
            for (int r = 0; r < RING; ++r)
                if (r < nsteps) issue(p0 + wave + r * nwaves, r, fresh);
            int slot = 0;
            for (int idx = 0; idx < nsteps; ++idx) {
                const int p = p0 + wave + idx * nwaves;
                // behind the wanted loads sit, per position issued since, NDMA loads and DC + 1 (+ 1 with log-ratios) stores
                if (!fresh && idx >= RING && idx + RING - 1 < nsteps) {
                    if (want_llr) wait_vmcnt<RING * (DC + 2) + (RING - 1) * NDMA>(); else wait_vmcnt<RING * (DC + 1) + (RING - 1) * NDMA>();
                } else {
                    wait_vmcnt<0>();
                }
                const int32_t *rec = a.pos_tab + (size_t)p * REC;
                double v[NO];
#pragma unroll
                for (int t = 0; t < NO; ++t) v[t] = ringp[slot * (SLOT_BYTES / 8) + t * LDPC_WAVE + lane];
                wait_lds_reads();  // the slot is free once its values sit in registers
                if (idx + RING < nsteps) issue(p + RING * nwaves, slot, fresh);
                const ldpc_v4i own = sload(reinterpret_cast<const ldpc_v4i *>(rec + NO_PAD));
                const int bit = sload(rec + NO_PAD + DC);
                if (fresh) {  // entries no earlier position has written hold their initial value (the same in all lanes)
                    const unsigned written = (unsigned)sload(rec + NO_PAD + DC + 1);
                    const ldpc_v16i oth = sload(reinterpret_cast<const ldpc_v16i *>(rec));
#pragma unroll
                    for (int t = 0; t < NO; ++t)
                        if (!((written >> t) & 1u)) v[t] = sload(a.edge0 + sload(a.col_idx + oth[t]));
                }
                double llr = sload(a.llr0 + bit);  // bp.hpp:488
                double c[DC], pre[DC];
#pragma unroll
                for (int k = 0; k < DC; ++k) {
                    const int chk = own[k] / DR;  // (every row has DR entries: row i starts at edge i DR)
                    const bool odd = (sload(par + chk) >> lane) & 1ull;  // pow(-1, syndrome byte) / syndrome parity
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) {
                        double x = 1.0;  // bp.hpp:492-498: the product over the row's other entries, in the row's order
#pragma unroll
                        for (int q = 0; q < DR - 1; ++q) x *= v[k * (DR - 1) + q];
                        c[k] = ps_message<MATH>(x, odd, log_tab);
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
                double temp = 0.0;  // bp.hpp:530-534
#pragma unroll
                for (int k = DC - 1; k >= 0; --k) {
                    At.st(l8, own[k], edge_form<METHOD, MATH>(pre[k] + temp));
                    temp += c[k];
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) LDPC_EDGE_FENCE();
                }
                const uint64_t hard = __ballot(llr <= 0);  // bp.hpp:525-529
                if (lane == 0) dcur[bit] = hard;
                if (want_llr && lane_live) Lt.st(l8, bit, llr);
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
            wait_vmcnt<0>();
            __syncthreads();  // the next level reads what this one wrote
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
    if (threadIdx.x == 0) clock_probe_end(a.clk, clk_stamp);
}
