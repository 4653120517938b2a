// host_stream.h -- host side of the streamed kernels: dispatch (decode_device), persistent + per-pass launches, two-pass decode with lane compaction
// Part of libldpc_hip.so: included by bp_hip.hip (one translation unit), in the order given there.
#pragma once


// nt: non-temporal cache policy for the message traffic (tiles that outgrow the 256 MB MALL; see MsgBufT)
template <bool LOOP>
static void pick_spread(const ldpc_hip_bp *h, bool nt, spread_kernel_t &kc, spread_kernel_t &kb) {
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) pick_spread_m<LDPC_HIP_MINIMUM_SUM, 0, LOOP>(h->max_row_deg, h->max_col_deg, nt, kc, kb);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) pick_spread_m<LDPC_HIP_PRODUCT_SUM, 1, LOOP>(h->max_row_deg, h->max_col_deg, nt, kc, kb);
    else pick_spread_m<LDPC_HIP_PRODUCT_SUM, 0, LOOP>(h->max_row_deg, h->max_col_deg, nt, kc, kb);
}

// Item tables of the variable-degree ring (bp_stream_kernel.h, LDPC_RING_VAR): the check rows, and the pairs of bit columns, in the
// order the wavefronts take them -- wavefront w of a workgroup of W takes entries w, w + W, w + 2 W, ...  Blocks of W items come
// alternately from the heavy and from the light end of the items sorted by size, so that along a wavefront's sequence a heavy item
// is followed by a light one and two consecutive items fit its queue together.  (The order changes nothing in the results: the rows
// of a check pass, and the columns of a bit pass, are independent of each other.)
static int ensure_var_ring_items(ldpc_hip_bp *h, int W) {
    if (h->var_items_built) return 0;
    const int m = h->m, n = h->n, pairs = (n + 1) / 2;
    std::vector<int32_t> col_ptr((size_t)n + 1, 0);
    for (int32_t c : h->h_col_idx) ++col_ptr[(size_t)c + 1];
    for (int j = 0; j < n; ++j) col_ptr[(size_t)j + 1] += col_ptr[(size_t)j];
    auto interleave = [W](std::vector<std::array<int32_t, 4>> &items) {
        auto units = [](const std::array<int32_t, 4> &it) { return (it[2] + (it[3] > 0 ? it[3] : 0) + 1) / 2; };
        std::stable_sort(items.begin(), items.end(), [&](const std::array<int32_t, 4> &x, const std::array<int32_t, 4> &y) { return units(x) > units(y); });
        std::vector<std::array<int32_t, 4>> out;
        out.reserve(items.size());
        size_t lo = 0, hi = items.size();
        for (bool heavy = true; lo < hi; heavy = !heavy)
            for (int k = 0; k < W && lo < hi; ++k) out.push_back(heavy ? items[lo++] : items[--hi]);
        items.swap(out);
    };
    std::vector<std::array<int32_t, 4>> rows((size_t)m), prs((size_t)pairs);
    for (int i = 0; i < m; ++i) rows[(size_t)i] = {h->h_row_ptr[(size_t)i], i, h->h_row_ptr[(size_t)i + 1] - h->h_row_ptr[(size_t)i], 0};
    for (int g = 0; g < pairs; ++g) {
        const int j0 = 2 * g, j1 = j0 + 1;
        prs[(size_t)g] = {col_ptr[(size_t)j0], g, col_ptr[(size_t)j0 + 1] - col_ptr[(size_t)j0], j1 < n ? col_ptr[(size_t)j1 + 1] - col_ptr[(size_t)j1] : -1};
    }
    interleave(rows);
    interleave(prs);
    int rc;
    if ((rc = h->var_row_items.ensure(16 * (size_t)(m ? m : 1)))) return rc;
    if ((rc = h->var_pair_items.ensure(16 * (size_t)(pairs ? pairs : 1)))) return rc;
    if (m) HIPCHK(hipMemcpy(h->var_row_items.p, rows.data(), 16 * (size_t)m, hipMemcpyHostToDevice));
    if (pairs) HIPCHK(hipMemcpy(h->var_pair_items.p, prs.data(), 16 * (size_t)pairs, hipMemcpyHostToDevice));
    h->var_items_built = true;
    return 0;
}

// Everything below runs on h->stream with device pointers only.
static int decode_stream_repacked(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                                  double *llr, int32_t *iters, uint8_t *conv);
int decode_device(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                  double *llr, int32_t *iters, uint8_t *conv, bool may_repack) {
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    if (tiles_total == 0) return LDPC_HIP_OK;
    if (!h->cont_A) h->timed_prev = h->timed_prev_mid = false;  // (a second pass keeps the first pass's events: ldpc_hip_bp_last_kernel_ms adds them)
    if (h->schedule == 0 || h->schedule == 2) return decode_serial(h, synd, batch, decoding, llr, iters, conv);
    {   // small code: the kernels that keep a syndrome's messages on chip (tu_onchip.hip), where one applies
        bool took = false;
        const int rc_onchip = decode_onchip(h, synd, batch, decoding, llr, iters, conv, &took);
        if (took || rc_onchip) return rc_onchip;
    }
    // streamed tiles: a tile runs until the slowest of its 64 syndromes is done.  Where most syndromes converge early
    // a short first pass + a second pass over the compacted rest does the same work in a fraction of the tile-iterations
    if (may_repack && h->repack_iters != 0 && h->max_iter >= 8 && tiles_total >= 512 && h->m > 0 && h->n > 0)
        return decode_stream_repacked(h, synd, batch, decoding, llr, iters, conv);
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
    h->last_chunk_tiles = chunk;
    if ((rc = h->tile_state.ensure(sizeof(TileState) * (size_t)chunk))) return rc;
    if ((rc = h->handoff_list.ensure(sizeof(int32_t) * (size_t)chunk))) return rc;
    if ((rc = h->counter.ensure(16))) return rc;
    if (!h->h_counters) HIPCHK(hipHostMalloc((void **)&h->h_counters, 16, hipHostMallocDefault));

    const int ring = h->regular ? h->ring_depth : 0;
    // the variable-degree ring (bp_stream_kernel.h, LDPC_RING_VAR) on request (VAR_RING 1) wherever it applies -- rows <= 16, columns <= 8.
    // Measured on the irregular n = 10 000 code (profiles/r5_irregular_paths.txt): +5 % over the register variant for product-sum, -4 % for
    // min-sum, and below the per-pass kernels for product-sum -- so it is not what runs by default anywhere.
    const bool var_ring = h->m > 0 && h->n > 0 && h->max_row_deg <= 16 && h->max_col_deg <= 8 && h->on("VAR_RING");
    KernelChoice kern;
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_kernel<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg, ring, var_ring);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_kernel<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg, ring, var_ring);
    else kern = pick_kernel<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg, ring, var_ring);
    const int var_units = !kern.var_ring ? 0 : h->sw("VAR_RING_UNITS") >= 8 ? (h->sw("VAR_RING_UNITS") <= 40 ? h->sw("VAR_RING_UNITS") : 40) : 11;
    if (kern.var_ring && (rc = ensure_var_ring_items(h, kern.max_waves))) return rc;
    // Product-sum on a matrix without a fixed-degree ring variant (irregular, or regular of another shape than (6,3) / (8,4), or the ring
    // switched off): the persistent kernel holds the check pass AND the bit pass in one register allocation -- 128 VGPRs with rows of up to 6
    // entries, 158-168 with 8 or 16: four, then three wavefronts per SIMD -- while the per-pass kernels hold one pass each (73-106 and 46-62
    // VGPRs: 4-6 and 8 per SIMD), and the exact product-sum arithmetic is a dependent chain per wavefront that needs the wavefronts: they get
    // through the same tile-iterations in 0.64 of the cycles (counter pass in profiles/r5_irregular_paths.txt).  So unless the caller set a
    // threshold such a batch takes the per-pass kernels from its first iteration: 0.45 -> 0.56 of HBM on the irregular code with rows of 3 .. 16
    // entries, 0.49 -> 0.63 and 0.46 -> 0.59 with rows of 3 .. 8, 0.57 -> 0.60 on the headline code with its ring off.  Min-sum has no such
    // chain and stays with the persistent kernel (0.69-0.77 against 0.65-0.68), and so do the ring variants (80 VGPRs: 0.65 against 0.60).
    // Bounded: a per-pass round is four launches over EVERY tile of the chunk (a row of workgroups per tile, leaving at once when the tile is
    // final: ~50 us per 256 rows and launch), queued by the host until the device reports the last tile final.  With one hopeless syndrome
    // and the reference's default max_iter = n that is thousands of full-size, empty rounds -- a cost that grows with the batch.  So only
    // decodes of at most 128 iterations start per-pass; longer ones keep the persistent kernel, whose hand-off parks at most 256 tiles
    // (the cost of an empty round is then the fixed ~0.2 ms it always was).
    const bool per_pass_first = h->handoff < 0 && h->bp_method == LDPC_HIP_PRODUCT_SUM && kern.ring_depth == 0 && !kern.var_ring &&
                                h->max_row_deg <= 16 && h->max_col_deg <= 8 && h->max_iter <= 128;
    const int handoff = h->handoff < 0 ? (per_pass_first ? INT32_MAX : 256) : h->handoff;
    if (!h->cont_extend) {
        h->accumulated_ms = 0.f;
        h->accumulated_persistent_ms = 0.f;
        h->timed = false;
        h->timed_mid = false;
    }
    hipStream_t st = h->stream;
    // second pass of a compacted decode: its rows are known to the device only -- `batch` is the most there can be, the kernels read the
    // real count (cont_rows_dev) and reach the caller's rows through cont_row_map; the grids of the tile-looping kernels follow an estimate
    const int32_t *row_map = h->cont_A ? h->cont_row_map : nullptr;
    const unsigned *rows_dev = h->cont_A ? h->cont_rows_dev : nullptr;
    if (h->cont_A && chunk < tiles_total) return fail(LDPC_HIP_ERR_NOMEM, "internal: the second pass of a compacted decode must be one chunk");

    for (int64_t t0 = 0; t0 < tiles_total; t0 += chunk) {
        const int64_t tiles = (tiles_total - t0 < chunk) ? tiles_total - t0 : chunk;
        const int64_t b0 = t0 * LDPC_WAVE;
        const int64_t nb = (batch - b0 < tiles * LDPC_WAVE) ? batch - b0 : tiles * LDPC_WAVE;

        HIPCHK(hipMemsetAsync(h->invalid.p, 0, sizeof(uint64_t) * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dec.p, 0, sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)tiles, st));
        const unsigned loop_tiles = rows_dev ? (unsigned)(h->cont_grid_tiles < tiles ? (h->cont_grid_tiles > 0 ? h->cont_grid_tiles : 1) : tiles) : (unsigned)tiles;
        if (h->m > 0) {
            dim3 g((unsigned)((h->m + 255) / 256), loop_tiles);
            hipLaunchKernelGGL(pack_syndromes_kernel, g, dim3(256), 0, st, synd + b0 * h->m, nb, h->m,
                               (uint64_t *)h->par.p, (uint64_t *)h->nzm.p, (uint64_t *)h->invalid.p, row_map, rows_dev);
        }
        BpArgs a = {};
        a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter;
        a.ms_scaling_factor = h->ms_scaling_factor;
        a.batch = nb;
        a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx;
        a.col_ptr = h->d_col_ptr; a.csc_edge = h->d_csc_edge;
        a.llr0 = h->d_llr0;
        a.A = (double *)h->msgA.p; a.C = (double *)h->msgC.p;
        if (h->cont_A) { a.A = h->cont_A; a.C = h->cont_C; a.it_start = h->cont_it_start; a.rows_dev = rows_dev; a.row_map = row_map; }
        a.keep_state = (h->keep_state || h->on("KEEP_LAST_MESSAGES")) ? 1 : 0;
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
        a.clk = h->d_clk;
        if (kern.var_ring) { a.row_items = (const int32_t *)h->var_row_items.p; a.pair_items = (const int32_t *)h->var_pair_items.p; a.ring_units = var_units; }
        HIPCHK(hipMemsetAsync(h->counter.p, 0, 16, st));

        // Wavefronts per workgroup (one workgroup = one 64-syndrome tile).  Register variant: 128 VGPRs,
        // 16 wavefronts per CU -> 4-wave workgroups once there are >= 4 tiles per CU.  Ring variant:
        // ~70 VGPRs and 6 KiB of LDS per wavefront -> 24 wavefronts per CU as two 12-wave workgroups
        // (3 wavefronts on each SIMD; measured best on MI355X, profiles/; 6-wave workgroups place
        // unevenly on the 4 SIMDs and 8-wave ones leave a ragged last round at 1024 tiles).
        int waves = h->waves_per_wg;
        if (waves <= 0) {
            if (kern.ring_depth) waves = tiles >= 512 ? 12 : 16;
            else if (kern.var_ring) waves = kern.max_waves;
            else waves = tiles >= 1024 ? 4 : (tiles >= 512 ? 8 : 16);
        }
        if (waves > kern.max_waves) waves = kern.max_waves;
        // ring variant: each wavefront owns RING slots of dynamic LDS; stay below the 160 KiB of a CU
        // + the parking space of the exact product-sum check row (LDPC_NEAR_BYTES per wavefront, behind the rings)
        const size_t near_bytes = (h->bp_method == LDPC_HIP_PRODUCT_SUM && h->math_mode == LDPC_HIP_MATH_LIBM_EXACT) ? LDPC_NEAR_BYTES : 0;
        const size_t lds_per_wave = (kern.var_ring ? (size_t)var_units * 1024u : (size_t)kern.ring_slot_bytes * (size_t)kern.ring_depth) + near_bytes;
        while (lds_per_wave * (size_t)waves > 144u * 1024u) --waves;
        const size_t dyn_lds = lds_per_wave * (size_t)waves;
        if (dyn_lds > 48u * 1024u)
            HIPCHK(hipFuncSetAttribute((const void *)kern.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        if (h->timed && !h->cont_extend) {  // fold the previous chunk's time before the events are re-recorded
            float prev = 0.f;
            HIPCHK(hipEventSynchronize(h->ev1));
            HIPCHK(hipEventElapsedTime(&prev, h->ev0, h->ev1));
            h->accumulated_ms += prev;
            if (h->timed_mid) {
                HIPCHK(hipEventElapsedTime(&prev, h->ev0, h->ev_mid));
                h->accumulated_persistent_ms += prev;
            }
        }
        if (!h->cont_extend) {  // (a third pass: the second pass's interval goes on -- its ev0 and ev_mid stay, ev1 is recorded again at the end)
            h->timed_mid = false;
            HIPCHK(hipEventRecord(h->ev0, st));
        }
        if (h->cont_A) {
            // the listed rows' message state after the first pass, lane by lane, into dense tiles (inside this pass's timed region)
            const int epw = 16;
            const dim3 gg((unsigned)((h->nnz + 4 * epw - 1) / (4 * epw)), loop_tiles);
            hipLaunchKernelGGL(gather_lane_state_kernel, gg, dim3(256), 0, st, (const double *)h->cont_C, h->cont_src_map ? h->cont_src_map : row_map, (int64_t)0, h->nnz, epw, h->cont_A, rows_dev);
            HIPCHK(hipGetLastError());
        }
        SpreadArgs sa = {};
        sa.bp = a;
        sa.host_flag = h->d_flag;
        sa.seq = ++h->flag_seq ? h->flag_seq : ++h->flag_seq;  // never 0 (the word's initial value)
        // per-pass rounds: `grid_tiles` workgroup rows; how many of them have a tile is known to the host only when the
        // batch skips the persistent kernel (sa.n_tiles >= 0), otherwise the kernels read it from counters[1]
        unsigned grid_tiles = 0;
        int first_round = 1;  // a tile parked by the persistent kernel has completed >= 1 iteration
        if (handoff > 0 && tiles <= handoff && h->max_iter - a.it_start > 1 && !rows_dev) {
            // so few tiles that they would each sit on one compute unit: per-pass launches from the start
            grid_tiles = (unsigned)tiles;
            sa.n_tiles = (int32_t)tiles;
            // rows (columns) per wavefront of a per-pass workgroup: 1 when a handful of tiles must fill the chip, 4 up to a few hundred
            // tiles (the chunks of the pipelined host path among them), 16 from 512 on -- a workgroup's start (the logarithm table into
            // LDS, the tile's state) is then paid per 64 rows instead of per 16: 0.563 against 0.535 of HBM on the irregular code's 512
            // tiles, same box (profiles/r5_irregular_paths.txt)
            sa.nodes = h->sw("SPREAD_NODES") > 0 ? h->sw("SPREAD_NODES") : tiles <= 8 ? 1 : tiles < 512 ? 4 : 16;
            first_round = 0;
            hipLaunchKernelGGL(bp_spread_state_init_kernel, dim3((grid_tiles + 255) / 256), dim3(256), 0, st, sa);
            const dim3 gi((unsigned)(h->nnz ? (h->nnz + 63) / 64 : 1), grid_tiles);  // (a grid dimension must not be 0: empty matrices)
            if (a.it_start > 0) { /* the message state is there already */ }
            else if (h->bp_method == LDPC_HIP_MINIMUM_SUM) hipLaunchKernelGGL((bp_spread_init_kernel<LDPC_HIP_MINIMUM_SUM, 0>), gi, dim3(256), 0, st, sa);
            else if (h->math_mode == LDPC_HIP_MATH_FAST) hipLaunchKernelGGL((bp_spread_init_kernel<LDPC_HIP_PRODUCT_SUM, 1>), gi, dim3(256), 0, st, sa);
            else hipLaunchKernelGGL((bp_spread_init_kernel<LDPC_HIP_PRODUCT_SUM, 0>), gi, dim3(256), 0, st, sa);
            HIPCHK(hipGetLastError());
        } else {
            if (kern.ring_depth && h->n > 0 && !h->on("EXPLICIT_INIT")) {  // the first check pass reads this table instead of initial messages
                if ((rc = h->d_edge0.ensure(sizeof(double) * (size_t)h->n))) return rc;
                const dim3 ge((unsigned)((h->n + 255) / 256));
                if (h->bp_method == LDPC_HIP_MINIMUM_SUM) hipLaunchKernelGGL((bp_edge0_kernel<LDPC_HIP_MINIMUM_SUM, 0>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
                else if (h->math_mode == LDPC_HIP_MATH_FAST) hipLaunchKernelGGL((bp_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 1>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
                else hipLaunchKernelGGL((bp_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 0>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
                a.edge0 = (const double *)h->d_edge0.p;
            }
            hipLaunchKernelGGL(kern.fn, dim3((unsigned)tiles), dim3((unsigned)(waves * LDPC_WAVE)), (unsigned)dyn_lds, st, a);
            HIPCHK(hipGetLastError());
            if (!h->cont_extend) {
                HIPCHK(hipEventRecord(h->ev_mid, st));
                h->timed_mid = true;
            }
            if (handoff > 0 && h->max_iter > 1) {
                // the persistent kernel parks at most `handoff` tiles (it starts parking when that many are unfinished);
                // how many it did park stays on the device
                grid_tiles = (unsigned)(tiles < handoff ? tiles : handoff);
                sa.n_tiles = -1;
                sa.nodes = h->sw("SPREAD_NODES2") > 0 ? h->sw("SPREAD_NODES2") : 4;  // (4, 8 and 16 measure the same on the headline's last 256 tiles)
            }
        }
        if (grid_tiles > 0) {
            // finish the parked tiles with chip-wide per-pass launches: check, bit, syndrome test, bookkeeping.  Every
            // round is queued at once; the host never waits.  A tile that is final (or a workgroup row without a tile)
            // leaves each kernel at its first instruction, and once the device has reported "nothing left" through
            // the host-mapped flag the host stops queueing -- which only matters when max_iter is far larger than
            // the iterations needed (the reference's default max_iter = n).
            spread_kernel_t kc, kb, kcl, kbl;
            // messages of the tiles in flight: 2 arrays x nnz x 512 B each; beyond ~the MALL they are streamed, not cached
            const bool nt = (double)grid_tiles * 2.0 * (double)per_tile_msg > 384.0 * 1024.0 * 1024.0;
            pick_spread<false>(h, nt, kc, kb);
            pick_spread<true>(h, nt, kcl, kbl);
            const unsigned per_wg = 4u * (unsigned)sa.nodes;
            const dim3 gc((unsigned)(h->m ? (h->m + per_wg - 1) / per_wg : 1), grid_tiles), gb((unsigned)(h->n ? (h->n + per_wg - 1) / per_wg : 1), grid_tiles);
            const dim3 gs((unsigned)(h->m ? (h->m + 255) / 256 : 1), grid_tiles), gf((unsigned)(h->n ? (h->n + 63) / 64 : 1), grid_tiles);
            const int rounds = h->max_iter - (first_round ? first_round : a.it_start);  // (a tile parked by the persistent kernel knows its own it0)
            const volatile unsigned *flag = h->h_flag;
            // Late rounds (bp_spread_kernels.h): where the steering histogram of a two-pass decode shows at most 24 rows still running 8
            // iterations into the second pass, its list of tiles is compacted on the device every 8 rounds and a round becomes 32 rows of
            // workgroups for the list's first 32 slots + 8 rows of the looping form for whatever lies beyond (normally nothing), instead of
            // `grid_tiles` rows that leave at once at ~50 us a launch.  Not elsewhere: when most tiles keep going (the headline's last 256
            // tiles run to iteration 50, a chunk of the pipelined host path likewise) the row-per-tile grid is what runs them fastest.
            const bool may_compact = rows_dev != nullptr && h->cont_late_rows >= 0 && h->cont_late_rows <= 24 && grid_tiles > 40 && !h->on("NO_SPREAD_COMPACT");
            bool compacted = false;
            for (int round = 0; round < rounds; ++round) {
                if (*flag == sa.seq) break;  // a look, not a wait
                sa.round = round;
                sa.slot0 = 0;
                if (may_compact && round >= 8 && round % 8 == 0) {
                    hipLaunchKernelGGL(bp_spread_compact_kernel, dim3(1), dim3(64), 0, st, sa);
                    sa.n_tiles = -1;  // (the count is the device's from here on: counters[1])
                    compacted = true;
                }
                if (!compacted) {
                    hipLaunchKernelGGL(kc, gc, dim3(256), 0, st, sa);
                    hipLaunchKernelGGL(kb, gb, dim3(256), 0, st, sa);
                    hipLaunchKernelGGL(bp_spread_synd_kernel<false>, gs, dim3(256), 0, st, sa);
                    hipLaunchKernelGGL(bp_spread_finish_kernel<false>, gf, dim3(256), 0, st, sa);
                } else {
                    SpreadArgs sb = sa;
                    sb.slot0 = 32;
                    hipLaunchKernelGGL(kc, dim3(gc.x, 32), dim3(256), 0, st, sa);
                    hipLaunchKernelGGL(kcl, dim3(gc.x, 8), dim3(256), 0, st, sb);
                    hipLaunchKernelGGL(kb, dim3(gb.x, 32), dim3(256), 0, st, sa);
                    hipLaunchKernelGGL(kbl, dim3(gb.x, 8), dim3(256), 0, st, sb);
                    hipLaunchKernelGGL(bp_spread_synd_kernel<false>, dim3(gs.x, 32), dim3(256), 0, st, sa);
                    hipLaunchKernelGGL(bp_spread_synd_kernel<true>, dim3(gs.x, 8), dim3(256), 0, st, sb);
                    hipLaunchKernelGGL(bp_spread_finish_kernel<false>, dim3(gf.x, 32), dim3(256), 0, st, sa);
                    hipLaunchKernelGGL(bp_spread_finish_kernel<true>, dim3(gf.x, 8), dim3(256), 0, st, sb);
                }
            }
            HIPCHK(hipGetLastError());
        }
        HIPCHK(hipEventRecord(h->ev1, st));
        h->timed = true;
        HIPCHK(hipGetLastError());
        if (h->on("DEBUG_HANDOFF")) {  // diagnostic only: waits for the device and reports what the persistent kernel parked
            unsigned c[4] = {0, 0, 0, 0};
            HIPCHK(hipStreamSynchronize(st));
            HIPCHK(hipMemcpy(c, h->counter.p, 16, hipMemcpyDeviceToHost));
            const TileState *ts = nullptr; (void)ts;
            std::vector<TileState> states((size_t)tiles);
            HIPCHK(hipMemcpy(states.data(), h->tile_state.p, sizeof(TileState) * (size_t)tiles, hipMemcpyDeviceToHost));
            std::vector<int32_t> list((size_t)tiles);
            HIPCHK(hipMemcpy(list.data(), h->handoff_list.p, sizeof(int32_t) * (size_t)tiles, hipMemcpyDeviceToHost));
            long sum_it0 = 0; int min_it0 = 1 << 30, max_it0 = 0;
            for (unsigned q = 0; q < c[1] && q < (unsigned)tiles; ++q) { const int it0 = states[(size_t)list[q]].it0; sum_it0 += it0; if (it0 < min_it0) min_it0 = it0; if (it0 > max_it0) max_it0 = it0; }
            fprintf(stderr, "[ldpc_hip] tiles %lld: finished by the persistent kernel %u, parked %u (iterations done when parked: min %d mean %.1f max %d), live afterwards %u\n",
                    (long long)tiles, c[0], c[1], c[1] ? min_it0 : 0, c[1] ? (double)sum_it0 / c[1] : 0.0, max_it0, c[2]);
        }

        if (h->n > 0) {
            dim3 g((unsigned)((h->n + 255) / 256), loop_tiles);
            hipLaunchKernelGGL(unpack_decoding_kernel, g, dim3(256), 0, st,
                               (const uint64_t *)h->dec.p, nb, h->n, decoding + b0 * h->n, row_map, rows_dev);
            if (llr) {
                dim3 gt((unsigned)((h->n + LDPC_WAVE - 1) / LDPC_WAVE), loop_tiles);
                hipLaunchKernelGGL(transpose_llr_kernel, gt, dim3(256), 0, st,
                                   (const double *)h->llr_t.p, nb, h->n, llr + (size_t)b0 * h->n, row_map, rows_dev);
            }
        }
        HIPCHK(hipGetLastError());
    }
    return LDPC_HIP_OK;
}


// Two passes of the streamed parallel schedule: k1 iterations for everyone, then the rows that have not converged are
// COMPACTED: their message state is gathered, lane by lane, out of the first pass's tiles into dense tiles, and the decode
// carries on from iteration k1 + 1 on those (same operations on the same values: same results).  A 64-syndrome tile runs until
// its slowest syndrome is done and moves all 64 lanes' messages until then; after the compaction the tiles hold live lanes
// only.  (Rounds 1 - 2 restarted the gathered rows from scratch, which only pays when almost everything has converged by k1.)
// Whether and where to cut depends on the noise, which the host cannot see -- so every streamed decode leaves a histogram of
// its iteration counts behind (one tiny kernel, copied asynchronously) and the next decode on the handle prices the
// alternatives with it, in tile-iterations per tile of the batch: F(j) = fraction converged within j iterations,
//     plain        sum_j (1 - F(j-1)^64)
//     cut at k     sum_{j<=k} (1 - F(j-1)^64)  +  gather  +  (1 - F(k)) sum_{j>k} (1 - G_k(j-1)^64),  G_k = F conditioned on > k,
// gather = reading one message array of every tile and writing the live share = (1 + (1 - F(k))) * gather_cost of an iteration (the
// flooding schedule moves four arrays per iteration: 1/4; the serial schedule six segments per edge: 1/6), plus the first pass's outputs for rows that are decoded on.  No work is wasted when nothing
// converges (the first call, and every call whose predecessor says "plain", run plain); results do not depend on any of this.
static int stream_first_pass_length(ldpc_hip_bp *h, double *live_after, double gather_cost = 0.25) {
    *live_after = 0.5;
    h->cont_late_rows = -1;
    for (int R = 0; R <= 3; ++R) h->cont_alive[R] = -1;
    if (h->repack_iters > 0) return h->repack_iters < h->max_iter ? h->repack_iters : 0;
    // The previous decode's histogram, IF its copy has landed -- a look, never a wait (the *_async entry points must not block): a
    // caller that queues decodes back to back is steered by the last histogram that did land
    if (h->hist_pending) {
        const hipError_t q = hipEventQuery(h->ev_hist);
        if (q == hipSuccess) {
            std::memcpy(h->hist_landed, h->h_hist, sizeof h->hist_landed);
            h->hist_landed_max_iter = h->hist_max_iter;
            h->hist_landed_valid = true;
            h->hist_pending = false;
        } else {
            (void)hipGetLastError();  // hipErrorNotReady is not an error
        }
    }
    if (!h->hist_landed_valid || h->hist_landed_max_iter != h->max_iter) return 0;
    const int full = h->max_iter, top = full < 255 ? full : 255;
    double total = 0;
    for (int j = 0; j < 256; ++j) total += h->hist_landed[j];
    if (total <= 0) return 0;
    std::vector<double> F((size_t)top + 1, 0.0);  // F[j]: converged within j iterations
    double acc = 0;
    for (int j = 1; j <= top; ++j) { acc += h->hist_landed[j]; F[(size_t)j] = acc / total; }
    auto Fj = [&](int j) { return F[(size_t)(j < top ? j : top)]; };
    auto tile_runs = [&](int j) { return 1.0 - std::pow(Fj(j - 1), 64.0); };  // still going at iteration j
    double plain = 0;
    for (int j = 1; j <= full; ++j) plain += tile_runs(j);
    double best = plain, prefix = 0;
    int best_k = 0;
    for (int k = 1; k < full && k <= top; ++k) {
        prefix += tile_runs(k);
        const double live = 1.0 - Fj(k);
        if (k < 2 || live <= 0.0 || live > 0.6) continue;
        double rest = 0;
        for (int j = k + 1; j <= full; ++j) {
            const double g = (Fj(j - 1) - Fj(k)) / live;  // of the rows alive after k: done within j - 1
            const double r = 1.0 - std::pow(g < 0 ? 0 : g, 64.0);
            rest += r;
            if (r < 1e-9 && j > top) break;
        }
        const double cost = prefix + gather_cost * (1.0 + live) + 0.1 + live * rest;
        if (cost < best) { best = cost; best_k = k; *live_after = live; }
    }
    if (best_k > 0) {  // rows (of the histogram's batch) still running 8 iterations into the second pass: the stragglers its late rounds are for
        double late = h->hist_landed[0];
        for (int j = best_k + 9; j < 256; ++j) late += h->hist_landed[j];
        h->cont_late_rows = (int64_t)late;
        h->cont_alive_total = (int64_t)total;  // ... and those still running after best_k + R iterations, R = 0 .. 3 (where the lane kernel takes over)
        for (int R = 0; R <= 3; ++R) {
            double alive = h->hist_landed[0];
            for (int j = best_k + R + 1; j < 256; ++j) alive += h->hist_landed[j];
            h->cont_alive[R] = (int64_t)alive;
        }
    }
    return best < 0.97 * plain ? best_k : 0;
}

static int stream_leave_histogram(ldpc_hip_bp *h, const int32_t *iters, const uint8_t *conv, int64_t batch) {
    int rc;
    if ((rc = h->sp_hist.ensure(256 * sizeof(unsigned)))) return rc;
    if (!h->h_hist) HIPCHK(hipHostMalloc((void **)&h->h_hist, 256 * sizeof(unsigned), hipHostMallocDefault));
    HIPCHK(hipMemsetAsync(h->sp_hist.p, 0, 256 * sizeof(unsigned), h->stream));
    int64_t blocks = (batch + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(iteration_histogram_kernel, dim3((unsigned)blocks), dim3(256), 0, h->stream, iters, conv, batch, (unsigned *)h->sp_hist.p);
    HIPCHK(hipMemcpyAsync(h->h_hist, h->sp_hist.p, 256 * sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipEventRecord(h->ev_hist, h->stream));
    h->hist_pending = true;
    h->hist_max_iter = h->max_iter;
    return LDPC_HIP_OK;
}

// rows_dev[0] = rows listed by osd_collect_kernel, rows_dev[1] = their 64-row tiles (BpArgs::rows_dev)
__global__ void repack_rows_kernel(const unsigned *__restrict__ counters, unsigned *__restrict__ rows_dev) {
    const unsigned c = counters[0];
    rows_dev[0] = c;
    rows_dev[1] = (c + LDPC_WAVE - 1) / LDPC_WAVE;
}

// pos[list[r]] = r for the listed rows: where a row of the caller's arrays sits in the second pass's compacted tiles
__global__ void __launch_bounds__(256) row_positions_kernel(const int32_t *__restrict__ list, const unsigned *__restrict__ count_dev, int32_t *__restrict__ pos) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < (int64_t)count_dev[0]) pos[list[r]] = (int32_t)r;
}

// out[i] = pos[list[i]] for the listed rows
__global__ void __launch_bounds__(256) compose_positions_kernel(const int32_t *__restrict__ list, const unsigned *__restrict__ count_dev, const int32_t *__restrict__ pos,
                                                                int32_t *__restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < (int64_t)count_dev[0]) out[r] = pos[list[r]];
}

// Nothing here waits for the device: the second pass is queued at once, sized for the most rows there can be (all of them), and
// finds out on the device how many rows the first pass left -- osd_collect_kernel lists them, repack_rows_kernel turns the count
// into rows / tiles, every kernel of the second pass reads those (BpArgs::rows_dev) and reaches the caller's arrays through the
// list (BpArgs::row_map), so no row is copied out and back.  No extra message memory either: the compacted bit_to_check state is
// gathered into the first pass's check_to_bit array (dead by then -- every iteration starts by rewriting it) and the second
// pass uses the first pass's bit_to_check array as ITS check_to_bit array.
static int decode_stream_repacked(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                                  double *llr, int32_t *iters, uint8_t *conv) {
    const int full = h->max_iter;
    const size_t B = (size_t)batch;
    int rc;
    if (!conv) { if ((rc = h->osd_conv.ensure(B))) return rc; conv = (uint8_t *)h->osd_conv.p; }
    if (!iters) { if ((rc = h->sp_iters.ensure(B * 4))) return rc; iters = (int32_t *)h->sp_iters.p; }
    double live = 0.5;
    int k1 = stream_first_pass_length(h, &live, 0.25);
    const int64_t tiles1 = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    if (k1 >= 2 && k1 < full) {
        // the compaction needs the whole batch's message state resident (one chunk); else decode plainly
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t per_tile = 2 * sizeof(double) * (size_t)h->nnz * LDPC_WAVE + (llr ? sizeof(double) * (size_t)h->n * LDPC_WAVE : 0) + 16 * (size_t)(h->m + h->n + 1);
        const size_t have = h->msgA.cap + h->msgC.cap + h->llr_t.cap;
        if ((double)per_tile * (double)tiles1 > (double)(free_b + have) * 0.85 || tiles1 > 32768 || (h->max_chunk_tiles > 0 && tiles1 > h->max_chunk_tiles) || h->nnz == 0) k1 = 0;
    }
    if (k1 < 2 || k1 >= full) {
        if ((rc = decode_device(h, synd, batch, decoding, llr, iters, conv, false))) return rc;
        return stream_leave_histogram(h, iters, conv, batch);
    }
    h->max_iter = k1;
    h->keep_state = true;
    rc = decode_device(h, synd, batch, decoding, llr, iters, conv, false);
    h->keep_state = false;
    h->max_iter = full;
    if (rc) return rc;
    if (h->last_chunk_tiles < tiles1) {
        // the first pass was cut into chunks after all (free memory moved between the estimate above and decode_device's own): its message
        // state is not resident at once, so there is nothing to compact -- decode the batch plainly (same results; the first pass's work is lost)
        if ((rc = decode_device(h, synd, batch, decoding, llr, iters, conv, false))) return rc;
        return stream_leave_histogram(h, iters, conv, batch);
    }
    if ((rc = h->osd_list.ensure(B * sizeof(int32_t)))) return rc;
    if ((rc = h->osd_counters.ensure(8 * sizeof(unsigned)))) return rc;
    HIPCHK(hipMemsetAsync(h->osd_counters.p, 0, 8 * sizeof(unsigned), h->stream));
    hipLaunchKernelGGL(osd_collect_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, h->stream, conv, batch,
                       (int32_t *)h->osd_list.p, (unsigned *)h->osd_counters.p);
    hipLaunchKernelGGL(repack_rows_kernel, dim3(1), dim3(1), 0, h->stream, (const unsigned *)h->osd_counters.p, (unsigned *)h->osd_counters.p + 2);
    HIPCHK(hipGetLastError());
    // the first pass's events stay readable while the second pass records its own (ldpc_hip_bp_last_kernel_ms adds both; nobody waits here)
    std::swap(h->ev0, h->evp0);
    std::swap(h->ev1, h->evp1);
    std::swap(h->ev_mid, h->evp_mid);
    h->timed_prev = h->timed;
    h->timed_prev_mid = h->timed_mid;
    // What is left after the first pass -- or after a few more rounds in compacted tiles -- CAN finish a workgroup per syndrome
    // (bp_flood_lane_kernel.h) instead of in tiles that keep moving 64 lanes for the one or two still alive in each: lane_after = the rounds
    // in tiles before that (0: none; -1: tiles to the end).  Measured on the headline code at p = 0.05 (round 5, same box, 64 .. 1024 workgroups,
    // after 0 .. 3 rounds): 108.8 - 111.2 ms against 109.4 ms for tiles to the end -- no gain: a row's two message arrays are 480 KB of 8-byte
    // gathers that only stay in L2 for a few dozen rows at a time, and 11 000 rows straight after the first pass cost 10 ms MORE than their
    // tiles.  So it is not chosen automatically; "FLOOD_LANES" 1 = straight after the first pass, 2 .. 4 = after 1 .. 3 rounds in tiles
    // (tests keep the path honest: it gives the tiles' bits).  The serial schedule's lane kernel (bp_serial_stream_kernel.h) is another
    // matter: there a tile-iteration is a chain of ~35 level barriers on one compute unit, here it is spread over the chip by the per-pass rounds.
    int lane_after = -1;
    {
        const int fl = h->sw("FLOOD_LANES");
        const bool fits = h->n <= 60000 && h->m <= 60000 && h->m > 0;  // (a byte per bit and per check in LDS; nodes heavier than the register bounds stream through memory)
        if (fits && fl != 0) {
            if (fl >= 1) lane_after = fl - 1 < full - k1 ? fl - 1 : -1;
        }
    }
    auto launch_lanes = [&](int it_start, const double *tiles, const int32_t *pos, const int32_t *rows, const unsigned *count_dev, bool own_interval) -> int {
        const int64_t groups = h->sw("FLOOD_LANE_GROUPS") > 0 ? h->sw("FLOOD_LANE_GROUPS") : 1024;
        if ((rc = h->flood_lane_scratch.ensure(2 * sizeof(double) * (size_t)h->nnz * (size_t)groups))) return rc;
        FloodLaneArgs fa = {};
        fa.m = h->m; fa.n = h->n; fa.nnz = h->nnz; fa.max_iter = full; fa.it_start = it_start;
        fa.ms_scaling_factor = h->ms_scaling_factor;
        fa.row_ptr = h->d_row_ptr; fa.col_idx = h->d_col_idx; fa.col_ptr = h->d_col_ptr; fa.csc_edge = h->d_csc_edge;
        fa.llr0 = h->d_llr0;
        fa.A_tiles = tiles;
        fa.pos = pos;
        fa.rows = rows;
        fa.count_dev = count_dev;
        fa.A = (double *)h->flood_lane_scratch.p;
        fa.C = fa.A + (size_t)h->nnz * (size_t)groups;
        fa.synd = synd; fa.decoding = decoding; fa.llr = llr; fa.iters = iters; fa.conv = conv;
        void (*kern)(const FloodLaneArgs);
        const bool wide = h->max_row_deg > 8 || h->max_col_deg > 4;
#define LDPC_PICK_FLOOD_LANE(M, F) (wide ? bp_flood_lane_kernel<M, F, 16, 8> : bp_flood_lane_kernel<M, F, 8, 4>)
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = LDPC_PICK_FLOOD_LANE(LDPC_HIP_MINIMUM_SUM, 0);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = LDPC_PICK_FLOOD_LANE(LDPC_HIP_PRODUCT_SUM, 1);
        else kern = LDPC_PICK_FLOOD_LANE(LDPC_HIP_PRODUCT_SUM, 0);
#undef LDPC_PICK_FLOOD_LANE
        const size_t dyn = (((size_t)h->n + 15) & ~(size_t)15) + (((size_t)h->m + 15) & ~(size_t)15);
        if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        if (own_interval) {
            h->accumulated_ms = 0.f;
            h->accumulated_persistent_ms = 0.f;
            h->timed_mid = false;
            HIPCHK(hipEventRecord(h->ev0, h->stream));
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3(512), (unsigned)dyn, h->stream, fa);
        HIPCHK(hipEventRecord(h->ev1, h->stream));  // (behind a second pass: its interval now ends here)
        h->timed = true;
        HIPCHK(hipGetLastError());
        return LDPC_HIP_OK;
    };
    if (lane_after == 0) {
        if ((rc = launch_lanes(k1, (const double *)h->msgA.p, nullptr, (const int32_t *)h->osd_list.p, (const unsigned *)h->osd_counters.p, true))) return rc;
        return stream_leave_histogram(h, iters, conv, batch);
    }
    // A SECOND COMPACTION (round 6).  Where the histogram says that a few iterations into the second pass most of ITS rows are done too -- the
    // headline code at p = 0.05: 17 % of the batch go on after iteration 7, 1.9 % after iteration 8 -- the rows that are left sit in every one of
    // the second pass's tiles, which keep moving 64 lanes' messages for them.  So the second pass stops after k2 iterations (keeping its
    // messages), the rows still decoding are listed again, their lane state is gathered once more -- out of the second pass's tiles, into the
    // array that was its check_to_bit scratch -- and a third pass finishes them in a tenth of the tiles.  Same machinery, same bits, still
    // nothing waits for the device.  MEASURED (profiles/r6_second_compaction_ab.txt, same box, interleaved): 111.5 - 113.4 ms with it against
    // 112.0 - 112.5 ms without on the headline code at p = 0.05, 146.5 - 148.0 against 146.1 - 146.7 on the irregular code at p = 0.06 -- the gather
    // and the extra launches cost what the thinner rounds save -- so it is NOT chosen by default: "REPACK2" 1 .. 3 = always, after that many
    // iterations of the second pass; 4 = where the histogram expects at most a fifth of the second pass's rows (but more than the late rounds'
    // list compaction is for) to be left after 1 .. 3 iterations.  The tests keep the path honest: it gives the plain decode's bits.
    int k2 = 0;
    if (lane_after < 0 && h->sw("REPACK2") > 0) {
        if (h->sw("REPACK2") <= 3) k2 = h->sw("REPACK2");
        else if (h->cont_alive[0] > 0)
            for (int R = 1; R <= 3 && !k2; ++R)
                if (h->cont_alive[R] >= 0 && h->cont_alive[R] * 5 <= h->cont_alive[0] && h->cont_alive[R] * (double)batch / (double)(h->cont_alive_total > 0 ? h->cont_alive_total : 1) > 256.0) k2 = R;
        if (k1 + k2 + 2 > full) k2 = 0;  // (nothing worth a gather is left to do)
    }
    const int64_t alive2 = k2 > 0 && h->cont_alive[k2] >= 0 && h->cont_alive_total > 0 ? (int64_t)((double)h->cont_alive[k2] * (double)batch / (double)h->cont_alive_total) : -1;
    h->cont_A = (double *)h->msgC.p;   // compacted bit_to_check state (gathered inside decode_device)
    h->cont_C = (double *)h->msgA.p;   // the gather's source, then the second pass's check_to_bit array
    h->cont_it_start = k1;
    h->cont_row_map = (const int32_t *)h->osd_list.p;
    h->cont_rows_dev = (const unsigned *)h->osd_counters.p + 2;
    // grids of the tile-looping kernels: the rows the histogram expects + a margin (they loop, so any count is handled)
    h->cont_grid_tiles = (int64_t)(live * 1.25 * (double)tiles1) + 8;
    if (lane_after > 0) { h->max_iter = k1 + lane_after; h->keep_state = true; }  // (lanes follow: these rounds leave their messages behind)
    else if (k2 > 0) { h->max_iter = k1 + k2; h->keep_state = true; }             // (a third pass follows)
    rc = decode_device(h, synd, batch, decoding, llr, iters, conv, false);
    h->keep_state = false;
    h->max_iter = full;
    h->cont_A = h->cont_C = nullptr;
    h->cont_it_start = 0;
    h->cont_row_map = nullptr;
    h->cont_rows_dev = nullptr;
    if (rc) return rc;
    if (k2 > 0 && lane_after <= 0) {
        // what the second pass left: the rows of the whole batch whose flag is still down (only its rows can be), where each sat in its tiles,
        // {rows, tiles} of the third pass on the device
        if ((rc = h->flood_list2.ensure(B * sizeof(int32_t))) || (rc = h->flood_pos.ensure(B * sizeof(int32_t))) || (rc = h->flood_src.ensure(B * sizeof(int32_t)))) return rc;
        unsigned *count2 = (unsigned *)h->osd_counters.p + 4;
        HIPCHK(hipMemsetAsync(count2, 0, 4 * sizeof(unsigned), h->stream));
        const dim3 gb((unsigned)((batch + 255) / 256));
        hipLaunchKernelGGL(osd_collect_kernel, gb, dim3(256), 0, h->stream, conv, batch, (int32_t *)h->flood_list2.p, count2);
        hipLaunchKernelGGL(row_positions_kernel, gb, dim3(256), 0, h->stream, (const int32_t *)h->osd_list.p, (const unsigned *)h->osd_counters.p, (int32_t *)h->flood_pos.p);
        hipLaunchKernelGGL(compose_positions_kernel, gb, dim3(256), 0, h->stream, (const int32_t *)h->flood_list2.p, (const unsigned *)count2, (const int32_t *)h->flood_pos.p,
                           (int32_t *)h->flood_src.p);
        hipLaunchKernelGGL(repack_rows_kernel, dim3(1), dim3(1), 0, h->stream, (const unsigned *)count2, count2 + 2);
        HIPCHK(hipGetLastError());
        h->cont_A = (double *)h->msgA.p;   // gathered out of the second pass's bit_to_check array ...
        h->cont_C = (double *)h->msgC.p;   // ... which then serves as the third pass's check_to_bit array
        h->cont_it_start = k1 + k2;
        h->cont_row_map = (const int32_t *)h->flood_list2.p;
        h->cont_src_map = (const int32_t *)h->flood_src.p;
        h->cont_rows_dev = (const unsigned *)count2 + 2;
        h->cont_grid_tiles = (alive2 >= 0 ? (int64_t)((double)alive2 * 1.5 / LDPC_WAVE) : tiles1 / 8) + 8;
        h->cont_late_rows = -1;
        h->cont_extend = true;
        rc = decode_device(h, synd, batch, decoding, llr, iters, conv, false);
        h->cont_extend = false;
        h->cont_A = h->cont_C = nullptr;
        h->cont_it_start = 0;
        h->cont_row_map = h->cont_src_map = nullptr;
        h->cont_rows_dev = nullptr;
        if (rc) return rc;
    }
    if (lane_after > 0) {
        // what those rounds left: the rows of the whole batch whose flag is still down (only rows of the second pass can be), where each sat in
        // the second pass's tiles (the inverse of its row list), and the lane kernel on them from the second pass's bit->check array
        if ((rc = h->flood_list2.ensure(B * sizeof(int32_t))) || (rc = h->flood_pos.ensure(B * sizeof(int32_t)))) return rc;
        unsigned *count2 = (unsigned *)h->osd_counters.p + 4;
        HIPCHK(hipMemsetAsync(count2, 0, 2 * sizeof(unsigned), h->stream));
        hipLaunchKernelGGL(osd_collect_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, h->stream, conv, batch, (int32_t *)h->flood_list2.p, count2);
        hipLaunchKernelGGL(row_positions_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, h->stream, (const int32_t *)h->osd_list.p,
                           (const unsigned *)h->osd_counters.p, (int32_t *)h->flood_pos.p);
        HIPCHK(hipGetLastError());
        if ((rc = launch_lanes(k1 + lane_after, (const double *)h->msgC.p, (const int32_t *)h->flood_pos.p, (const int32_t *)h->flood_list2.p, count2, false))) return rc;
    }
    return stream_leave_histogram(h, iters, conv, batch);
}
