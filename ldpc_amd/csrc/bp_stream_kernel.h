// bp_stream_kernel.h -- bp_decode_kernel: the persistent one-workgroup-per-tile streaming kernel (bp.hpp:192-325)
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// One workgroup owns one 64-syndrome tile for the whole decode: check pass, barrier, bit pass, barrier, syndrome
// test, up to max_iter times (design notes at the top of bp_hip.hip and in DESIGN.md section 4).
//   METHOD  LDPC_HIP_PRODUCT_SUM | LDPC_HIP_MINIMUM_SUM      MATH  0 libm-exact | 1 fast (product-sum only)
//   DR, DC  register bounds on row / column weight; heavier nodes are streamed through memory in two sweeps
//   RING    0: next row / bits prefetched into VGPRs;  2 or 3: slots of the per-wavefront LDS ring filled by
//           `buffer_load_dwordx4 ... lds` (only for matrices with a single row weight DR and column weight DC);
//           LDPC_RING_VAR: the ring as a queue of 1 KiB units for rows of 0 .. DR entries and column pairs of 0 .. 2 DC entries
//           (irregular matrices; "variable-degree ring" below)
// Register budget: the ring variants fit 80 VGPRs (six wavefronts per SIMD, 12-wavefront workgroups); the register-prefetch variants get
// 128 (four per SIMD) up to DR = 6, 168 at DR = 8 (workgroups of at most 12 wavefronts) and 256 at DR = 16 (at most 8): a row of 16
// entries with its prefix products and one exact transcendental in flight does not fit 128 without spilling into the row loop
// (host side: stream_max_waves).
// Variable-degree ring.  A wavefront's LDS is a circular queue of U units of 1 KiB (one DMA instruction = two 512-byte edge segments).
// An item -- a check row (its entries are consecutive edges of A) or a pair of bit columns (their entries are gathered from C through
// csc_edge, the second column's behind the first's) -- takes ceil(entries / 2) units at the queue's head.  The wavefront copies the
// current item from LDS into registers, frees its units, queues the DMAs of up to two items ahead as far as they fit, and computes.
// Items come in an order the host chose so that heavy and light ones alternate along each wavefront's sequence (BpArgs::row_items,
// pair_items): two consecutive ones then fit the queue nearly always.  The wait before an item is read is the counted one of the
// fixed-degree ring, with the count kept at run time: `ops` counts the vector-memory instructions this wavefront has issued (DMAs,
// message stores, decision words -- the posterior stores, which depend on the tile's state, are left out: a count that is too small
// only waits longer), an item remembers `ops` after its last DMA, and the difference is what may still be outstanding.
// 168 VGPRs (three wavefronts per SIMD, 12 per workgroup): a row of 16 entries with its prefix products and the results of the table
// branch of the exact logarithm fits without spilling.
#define LDPC_RING_VAR 9
typedef int ldpc_v4i __attribute__((ext_vector_type(4)));
constexpr int stream_max_waves(int dr, int ring) { return ring == LDPC_RING_VAR ? 12 : ring ? 16 : dr > 8 ? 8 : dr > 6 ? 12 : 16; }
// (fixed-degree ring variants with rows of more than 8 entries -- (5,10)-regular codes, round 6 -- get 128 VGPRs: a row of 10 with its prefix products does not fit 80)
constexpr int stream_waves_per_eu(int dr, int ring) { return ring == LDPC_RING_VAR ? 3 : ring ? (dr > 8 ? 4 : 6) : dr > 8 ? 2 : dr > 6 ? 3 : 4; }
template <int METHOD, int MATH, int DR, int DC, int RING>
__global__ void __launch_bounds__(64 * stream_max_waves(DR, RING)) __attribute__((amdgpu_waves_per_eu(stream_waves_per_eu(DR, RING)))) bp_decode_kernel(const BpArgs a) {
    constexpr int UB = DC <= 4 ? 4 : (DC <= 8 ? 2 : 1);  // bits in flight per wavefront (register variant)
    const int lane = threadIdx.x & (LDPC_WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int nwaves = (int)(blockDim.x >> 6);
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    if (a.rows_dev && tile >= (int64_t)sload(a.rows_dev + 1)) return;  // (rows known to the device only: this tile does not exist)

    const int32_t *__restrict__ row_ptr = a.row_ptr;
    const int32_t *__restrict__ col_idx = a.col_idx;
    const int32_t *__restrict__ col_ptr = a.col_ptr;
    const int32_t *__restrict__ csc_edge = a.csc_edge;
    const double *__restrict__ llr0 = a.llr0;
    const uint64_t *__restrict__ par = a.par + tile * m;
    const uint64_t *__restrict__ nzm = a.nzm + tile * m;

    // (non-temporal policy: this kernel only runs batches of hundreds of tiles, whose messages no cache can hold)
    const MsgBufNT At = make_msgbuf<MsgBufNT>(a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    const MsgBufNT Ct = make_msgbuf<MsgBufNT>(a.C + (size_t)tile * (size_t)nnz * LDPC_WAVE, (unsigned)nnz);
    uint64_t *dec = a.dec + tile * n;    // frozen decisions of converged syndromes (zero-initialised)
    uint64_t *dcur = a.dcur + tile * n;  // this iteration's hard decisions, all lanes
    const bool want_llr = a.llr_t != nullptr;
    const MsgBufNT Lt = make_msgbuf<MsgBufNT>(want_llr ? a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE : a.A, want_llr ? (unsigned)n : 0u);
    const int l8 = lane * 8;

    __shared__ uint64_t red[2][16];
    __shared__ int red_i;
    __shared__ unsigned long long clk_stamp[2];
    if (threadIdx.x == 0) clock_probe_begin(clk_stamp);
    __shared__ __attribute__((aligned(16))) double log_tab[256];  // glibc log's {1/c, log c} table, LDS-resident
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = threadIdx.x; q < 256; q += blockDim.x) log_tab[q] = ldpc_math::k_log_tab[q];

    // ring geometry (RING variant): one slot holds a check row (DR segments) or a pair of bit columns (2*DC)
    constexpr int ROW_DMAS = (DR + 1) / 2;                       // 1 KiB DMA instructions per row
    constexpr int SLOT_BYTES = (ROW_DMAS > DC ? ROW_DMAS : DC) * 1024;
    constexpr int N_CHECK = RING * DR + (RING - 1) * ROW_DMAS;
    constexpr int N_BIT = RING * (2 * DC + 2) + (RING - 1) * DC;
    constexpr int N_BIT_QUIET = RING * 2 + (RING - 1) * DC;  // the decode's last bit pass sends no messages: only the two decision words per step are stored
    constexpr bool VAR = RING == LDPC_RING_VAR;
    static_assert(!VAR || (DR % 2 == 0 && DR <= 16 && DC <= 8), "an item of the variable-degree ring is at most 8 units");
    const int U = VAR ? a.ring_units : 0;  // units of the variable-degree queue
    const unsigned ring_bytes = VAR ? (unsigned)U * 1024u : (unsigned)(RING * SLOT_BYTES);  // per wavefront
    const unsigned ring_addr = (unsigned)(uintptr_t)ldpc_dyn_lds + (unsigned)wave * ring_bytes;
    const double *ringp = reinterpret_cast<const double *>(ldpc_dyn_lds + (size_t)wave * ring_bytes);
    // parking space of the exact product-sum check row (check_row_ps_exact_fast): behind the rings, LDPC_NEAR_BYTES per wavefront
    double *near_buf = reinterpret_cast<double *>(ldpc_dyn_lds + (size_t)nwaves * ring_bytes + (size_t)wave * LDPC_NEAR_BYTES);
    const unsigned l16 = (unsigned)lane * 16u;

    bool llr_each = false;  // tile-uniform: posteriors are stored by every bit pass (set at the first convergence event)
    // lanes beyond the batch (partial last tile) are born "done"
    const int64_t valid = (a.rows_dev ? (int64_t)sload(a.rows_dev) : a.batch) - tile * LDPC_WAVE;
    uint64_t done = valid >= LDPC_WAVE ? 0ull : ~((1ull << valid) - 1ull);
    const uint64_t never = a.invalid[tile];
    int my_iter = 0;  // meaningful in wave 0: iteration at which this lane's syndrome converged

    // initialise_log_domain_bp (bp.hpp:147-157): every edge of column j starts at llr0[j] -- written out, or (ring variant, a.edge0)
    // left implicit: the first check pass reads the table of initial values instead of the message array
    const bool implicit_init = RING != 0 && !VAR && a.edge0 != nullptr;
    if (!implicit_init && a.it_start == 0) {
        for (int e = wave; e < nnz; e += nwaves) At.st(l8, e, edge_form<METHOD, MATH>(sload(llr0 + sload(col_idx + e))));
        __syncthreads();
    }

    // Second pass of a compacted decode whose rows (known to the device only) fill no more tiles than the hand-off threshold: a tile
    // would sit alone on a compute unit (~6 ms for its first iteration on the n = 10 000 code against ~1.4 ms when the chip shares it),
    // and the host, not knowing the count, could not send the batch to the per-pass kernels itself -- so the tile parks at once, before
    // its first iteration, exactly as a straggler parks further down.
    if (a.rows_dev && a.handoff_threshold > 0 && a.it_start > 0 && a.it_start < a.max_iter && (int)sload(a.rows_dev + 1) <= a.handoff_threshold) {
        TileState *stt = a.state + tile;
        if (wave == 0) stt->lane_iter[lane] = 0;
        if (threadIdx.x == 0) {
            stt->done[0] = done;
            stt->it0 = a.it_start;
            stt->end_round = INT32_MAX;
            stt->unsat[0] = stt->unsat[1] = 0ull;
            stt->llr_each[0] = 1;  // (lanes compacted out of a first pass are the ones about to converge: as bp_spread_state_init_kernel)
            a.handoff_list[atomicAdd(&a.counters[1], 1u)] = (int32_t)tile;
            atomicAdd(&a.counters[2], 1u);
            clock_probe_end(a.clk, clk_stamp);
        }
        return;
    }

    for (int it = a.it_start + 1; it <= a.max_iter; ++it) {
        // ---------------- check pass (bp.hpp:201-273) ----------------
        double alpha = 0.0;
        if (METHOD == LDPC_HIP_MINIMUM_SUM)
            alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;

        if (VAR) {
            // ---- variable-degree ring: rows of 0 .. DR entries (see the top of the file)
            const int nsteps = wave < m ? (m - wave + nwaves - 1) / nwaves : 0;
            const ldpc_v4i *items = reinterpret_cast<const ldpc_v4i *>(a.row_items);
            int head = 0;
            unsigned ops = 0;
            auto issue_row = [&](const ldpc_v4i &item, int &pos, unsigned &mark) {
                const int units = (item.z + 1) >> 1;
                pos = head;
#pragma unroll
                for (int c = 0; c < DR / 2; ++c)
                    if (c < units) {
                        int u = head + c;
                        if (u >= U) u -= U;
                        lds_dma16(At.rsrc, l16, (unsigned)(item.x + 2 * c) << 9, ring_addr + (unsigned)u * 1024u);
                    }
                head += units;
                if (head >= U) head -= U;
                ops += (unsigned)units;
                mark = ops;
            };
            ldpc_v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, nx = c0;  // current item, the (up to) two queued behind it, the first not queued
            int p0 = 0, p1 = 0, p2 = 0, ahead = 0, queued = 0;
            unsigned m0 = 0, m1 = 0, m2 = 0;
            if (nsteps > 0) {
                c0 = sload(items + wave);
                issue_row(c0, p0, m0);
                queued = 1;
                if (nsteps > 1) nx = sload(items + wave + nwaves);
            }
            for (int idx = 0; idx < nsteps; ++idx) {
                wait_vmcnt_dyn((int)(ops - m0));
                const int d = c0.z;
                double cur[DR];
#pragma unroll
                for (int k = 0; k < DR; ++k)
                    if (k < d) {
                        int u = p0 + (k >> 1);
                        if (u >= U) u -= U;
                        cur[k] = ringp[u * 128 + (k & 1) * LDPC_WAVE + lane];
                    }
                wait_lds_reads();  // the item's units are free
                if (ahead == 0 && queued < nsteps) {
                    c1 = nx;
                    issue_row(c1, p1, m1);
                    ahead = 1;
                    if (++queued < nsteps) nx = sload(items + wave + queued * nwaves);
                }
                if (ahead == 1 && queued < nsteps && ((c1.z + 1) >> 1) + ((nx.z + 1) >> 1) <= U) {
                    c2 = nx;
                    issue_row(c2, p2, m2);
                    ahead = 2;
                    if (++queued < nsteps) nx = sload(items + wave + queued * nwaves);
                }
                const int i = c0.y;
                const bool neg = (sload(nzm + i) >> lane) & 1ull;
                const int parity = (int)((sload(par + i) >> lane) & 1ull);
                ops += (unsigned)check_row_live<METHOD, MATH, DR>(cur, d, c0.x, neg, parity, alpha, Ct, l8, log_tab, ~done, near_buf);  // the stores the routine that ran REPORTS (one per entry)
                c0 = c1; p0 = p1; m0 = m1;
                c1 = c2; p1 = p2; m1 = m2;
                if (ahead > 0) --ahead;
            }
        } else if (RING && implicit_init && it == 1) {
            for (int i = wave; i < m; i += nwaves) {
                double cur[DR];
#pragma unroll
                for (int k = 0; k < DR; ++k) cur[k] = sload(a.edge0 + sload(col_idx + i * DR + k));  // (the same in all 64 lanes)
                const bool neg = (sload(nzm + i) >> lane) & 1ull;
                const int parity = (int)((sload(par + i) >> lane) & 1ull);
                check_row_live<METHOD, MATH, DR>(cur, DR, i * DR, neg, parity, alpha, Ct, l8, log_tab, ~done, near_buf);
            }
        } else if (RING) {
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
                check_row_live<METHOD, MATH, DR>(cur, DR, i * DR, neg, parity, alpha, Ct, l8, log_tab, ~done, near_buf);
                slot = slot + 1 == RING ? 0 : slot + 1;
            }
        } else if (DR > 8) {
            // rows of up to 16 entries in registers: 2 x 16 values for the row and its prefix products leave no room for a second row in
            // flight (a register double buffer spilled 84 - 1 231 VGPRs here) -- the other wavefronts of the SIMD cover the load latency
            for (int i = wave; i < m; i += nwaves) {
                const int rs = sload(row_ptr + i), d = sload(row_ptr + i + 1) - rs;
                const bool neg = (sload(nzm + i) >> lane) & 1ull;
                const int parity = (int)((sload(par + i) >> lane) & 1ull);
                if (d <= DR) {
                    double cur[DR];
#pragma unroll
                    for (int k = 0; k < DR; ++k)
                        if (k < d) cur[k] = At.ld(l8, rs + k);
                    check_row_live<METHOD, MATH, DR>(cur, d, rs, neg, parity, alpha, Ct, l8, log_tab, ~done, near_buf);
                } else {
                    check_row_streamed<METHOD, MATH>(d, rs, neg, parity, alpha, At, Ct, l8, log_tab);
                }
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
                if (d <= DR) check_row_live<METHOD, MATH, DR>(cur, d, rs, neg, parity, alpha, Ct, l8, log_tab, ~done, near_buf);
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
        if (VAR) {
            // ---- variable-degree ring: pairs of columns of 0 .. DC entries each; the pair's entries are positions
            // [item.x, item.x + d0 + d1) of csc_edge, a DMA instruction gathers two of them (lanes 0-31 one segment, 32-63 the next)
            const int ngroups = (n + 1) / 2;
            const int nsteps = wave < ngroups ? (ngroups - wave + nwaves - 1) / nwaves : 0;
            const ldpc_v4i *items = reinterpret_cast<const ldpc_v4i *>(a.pair_items);
            const bool sends = !last || a.keep_state != 0;
            int head = 0;
            unsigned ops = 0;
            auto entries = [](const ldpc_v4i &item) { return item.z + (item.w > 0 ? item.w : 0); };
            auto issue_pair = [&](const ldpc_v4i &item, int &pos, unsigned &mark) {
                const int units = (entries(item) + 1) >> 1;
                pos = head;
#pragma unroll
                for (int c = 0; c < DC; ++c)
                    if (c < units) {
                        const int q0 = item.x + 2 * c, q1 = q0 + 1;
                        const unsigned ea = (unsigned)sload(csc_edge + q0);
                        const unsigned eb = (unsigned)sload(csc_edge + (q1 < nnz ? q1 : 0));
                        const unsigned voff = ((lane < 32 ? ea : eb) << 9) + (unsigned)(lane & 31) * 16u;
                        int u = head + c;
                        if (u >= U) u -= U;
                        lds_dma16(Ct.rsrc, voff, 0u, ring_addr + (unsigned)u * 1024u);
                    }
                head += units;
                if (head >= U) head -= U;
                ops += (unsigned)units;
                mark = ops;
            };
            ldpc_v4i c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, nx = c0;
            int p0 = 0, p1 = 0, p2 = 0, ahead = 0, queued = 0;
            unsigned m0 = 0, m1 = 0, m2 = 0;
            if (nsteps > 0) {
                c0 = sload(items + wave);
                issue_pair(c0, p0, m0);
                queued = 1;
                if (nsteps > 1) nx = sload(items + wave + nwaves);
            }
            for (int idx = 0; idx < nsteps; ++idx) {
                wait_vmcnt_dyn((int)(ops - m0));
                const int d0 = c0.z, d1 = c0.w;  // (d1 < 0: the matrix has an odd number of columns and this is the last)
                double c[2][DC];
#pragma unroll
                for (int u2 = 0; u2 < 2; ++u2)
#pragma unroll
                    for (int k = 0; k < DC; ++k)
                        if (k < (u2 ? d1 : d0)) {
                            const int rel = (u2 ? d0 : 0) + k;
                            int u = p0 + (rel >> 1);
                            if (u >= U) u -= U;
                            c[u2][k] = ringp[u * 128 + (rel & 1) * LDPC_WAVE + lane];
                        }
                wait_lds_reads();
                if (ahead == 0 && queued < nsteps) {
                    c1 = nx;
                    issue_pair(c1, p1, m1);
                    ahead = 1;
                    if (++queued < nsteps) nx = sload(items + wave + queued * nwaves);
                }
                if (ahead == 1 && queued < nsteps && ((entries(c1) + 1) >> 1) + ((entries(nx) + 1) >> 1) <= U) {
                    c2 = nx;
                    issue_pair(c2, p2, m2);
                    ahead = 2;
                    if (++queued < nsteps) nx = sload(items + wave + queued * nwaves);
                }
#pragma unroll
                for (int u2 = 0; u2 < 2; ++u2) {
                    const int j = 2 * c0.y + u2, dj = u2 ? d1 : d0;
                    if (dj >= 0) {
                        int e[DC];
#pragma unroll
                        for (int k = 0; k < DC; ++k)
                            if (k < dj) e[k] = sload(csc_edge + c0.x + (u2 ? d0 : 0) + k);
                        int sent;
                        const double llr = bit_column<METHOD, MATH, DC>(c[u2], e, dj, sload(llr0 + j), At, l8, sends, &sent);
                        const uint64_t hard = __ballot(llr <= 0);  // bp.hpp:290
                        if (lane == 0) dcur[j] = hard;
                        if ((last || llr_each) && want_llr && lane_live) Lt.st(l8, j, llr);
                        ops += (unsigned)sent + 1u;  // the message stores bit_column reports + the decision word
                    }
                }
                c0 = c1; p0 = p1; m0 = m1;
                c1 = c2; p1 = p2; m1 = m2;
                if (ahead > 0) --ahead;
            }
        } else if (RING) {
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
                // (the counted wait must not exceed what was really issued behind the wanted loads, or it proves nothing: a pass without
                // message stores counts fewer -- getting this wrong reads a ring slot before its data has landed)
                const bool sends = !last || a.keep_state != 0;
                if (idx >= RING && idx + RING - 1 < nsteps) { if (sends) wait_vmcnt<N_BIT>(); else wait_vmcnt<N_BIT_QUIET>(); }
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
                        const double llr = bit_column<METHOD, MATH, DC>(c[u], e, DC, sload(llr0 + j), At, l8, sends);
                        const uint64_t hard = __ballot(llr <= 0);  // bp.hpp:290
                        if (lane == 0) dcur[j] = hard;
                        if ((last || llr_each) && want_llr && lane_live) Lt.st(l8, j, llr);
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
                        llr = bit_column<METHOD, MATH, DC>(c[u], e[u], dg[u], prior, At, l8, !last || a.keep_state != 0);
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
                    if ((last || llr_each) && want_llr && lane_live) Lt.st(l8, j, llr);
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
            if (!last && want_llr && !llr_each) {
                // first convergence in this tile: rebuild the posteriors of the lanes that stop from C, once
                for (int j = wave; j < n; j += nwaves) {
                    if (lane == 0) dec[j] = (dec[j] & ~newly) | (dcur[j] & newly);
                    double temp = llr0[j];
                    for (int p = col_ptr[j]; p < col_ptr[j + 1]; ++p) temp += Ct.ld(l8, csc_edge[p]);
                    if (mine) Lt.st(l8, j, temp);
                }
            } else {
                // the bit pass has stored this iteration's posteriors already (last iteration, or llr_each)
                for (int j = threadIdx.x; j < n; j += blockDim.x) dec[j] = (dec[j] & ~newly) | (dcur[j] & newly);
            }
            // Once syndromes of a tile start to converge the others usually follow within a few iterations: from now
            // on the bit pass stores the posteriors of the live lanes every iteration (+8 % traffic) instead of this
            // tile paying a full extra sweep over C per convergence event.
            llr_each = true;
            done |= newly;
            __syncthreads();  // C is overwritten by the next check pass
        }
        if (done == ~0ull) break;
        // Few tiles left running (stragglers, a tiny batch, or the tail of the launch): a lone tile is bound to
        // ONE compute unit (~3 ms per iteration on the n = 10 000 code), so park its state and let the per-pass
        // kernels below spread its remaining iterations over the whole chip.
        if (a.handoff_threshold > 0 && it < a.max_iter) {
            if (threadIdx.x == 0)
                red_i = (a.rows_dev ? (int)a.rows_dev[1] : a.total_tiles) - (int)__hip_atomic_load(&a.counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (red_i <= a.handoff_threshold) {
                TileState *stt = a.state + tile;
                if (wave == 0) stt->lane_iter[lane] = my_iter;
                if (threadIdx.x == 0) {
                    stt->done[0] = done;
                    stt->it0 = it;
                    stt->end_round = INT32_MAX;
                    stt->unsat[0] = stt->unsat[1] = 0ull;
                    stt->llr_each[0] = llr_each ? 1 : 0;  // (the per-pass kernels carry the policy on)
                    a.handoff_list[atomicAdd(&a.counters[1], 1u)] = (int32_t)tile;
                    atomicAdd(&a.counters[2], 1u);  // live tiles of the per-pass rounds
                    clock_probe_end(a.clk, clk_stamp);
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
        if (b < (a.rows_dev ? (int64_t)a.rows_dev[0] : a.batch)) {
            const int64_t row = a.row_map ? (int64_t)a.row_map[b] : b;
            const bool cv = ((done >> lane) & 1ull) != 0;
            if (a.iters) a.iters[row] = cv ? my_iter : a.max_iter;  // bp.hpp:304
            if (a.conv) a.conv[row] = cv ? 1 : 0;
        }
    }
    if (threadIdx.x == 0 && a.counters) atomicAdd(&a.counters[0], 1u);
    if (threadIdx.x == 0) clock_probe_end(a.clk, clk_stamp);
}
