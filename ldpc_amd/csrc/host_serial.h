// host_serial.h -- host side of the serial schedules (fixed order, random order, serial_relative) and of soft-syndrome decoding
// Part of libldpc_hip.so: included by bp_hip.hip (one translation unit), in the order given there.
#pragma once

// levels of the serial schedule: see bp_serial_level_kernel
static int ensure_serial_levels(ldpc_hip_bp *h) {
    if (h->levels_valid) return LDPC_HIP_OK;
    const int m = h->m, n = h->n;
    std::vector<int32_t> order((size_t)n);
    if (h->custom_order) HIPCHK(hipMemcpy(order.data(), h->d_order, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost));
    else for (int j = 0; j < n; ++j) order[(size_t)j] = j;
    std::vector<std::vector<int32_t>> checks_of((size_t)n);
    for (int i = 0; i < m; ++i)
        for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) checks_of[(size_t)h->h_col_idx[(size_t)e]].push_back(i);
    // (the order need not be a permutation -- the reference accepts any n bit numbers -- so levels belong to POSITIONS)
    std::vector<int32_t> check_level((size_t)(m ? m : 1), 0), level((size_t)(n ? n : 1), 1);
    int32_t n_levels = n ? 1 : 0;
    for (int t = 0; t < n; ++t) {
        const int j = order[(size_t)t];
        int32_t l = 1;
        for (int i : checks_of[(size_t)j]) l = std::max(l, check_level[(size_t)i] + 1);
        for (int i : checks_of[(size_t)j]) check_level[(size_t)i] = l;
        level[(size_t)t] = l;
        n_levels = std::max(n_levels, l);
    }
    std::vector<int32_t> ptr((size_t)n_levels + 1, 0), bits((size_t)(n ? n : 1));
    for (int t = 0; t < n; ++t) ptr[(size_t)level[(size_t)t]]++;
    for (int l = 0; l < n_levels; ++l) ptr[(size_t)l + 1] += ptr[(size_t)l];
    {
        std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
        for (int t = 0; t < n; ++t) bits[(size_t)fill[(size_t)level[(size_t)t] - 1]++] = order[(size_t)t];  // schedule order inside a level
    }
    int rc;
    if ((rc = h->lvl_ptr.ensure(sizeof(int32_t) * ((size_t)n_levels + 1))) || (rc = h->lvl_bits.ensure(sizeof(int32_t) * (size_t)(n ? n : 1)))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->lvl_ptr.p, ptr.data(), sizeof(int32_t) * ((size_t)n_levels + 1), hipMemcpyHostToDevice));
    if (n) HIPCHK(hipMemcpy(h->lvl_bits.p, bits.data(), sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice));
    h->n_levels = n_levels;
    h->levels_valid = true;
    h->h_lvl_ptr = std::move(ptr);
    h->h_lvl_bits = std::move(bits);
    h->ser_pos_valid = false;
    h->ser_var_valid = false;
    return LDPC_HIP_OK;
}

// bp_serial_stream_kernel: which form (if any) takes the fixed-order serial schedule of this matrix.  Single row weight, single column
// weight (the records and the ring slots are sized by them), levels wide enough to keep a workgroup's wavefronts busy.
struct SerialStreamPlan {
    int dr = 0, dc = 0;  // 0: not applicable
    bool var = false;    // the item form (bp_serial_var_kernel.h): dr, dc are the register bounds of the instantiation (8 | 16, 4 | 8)
};
static SerialStreamPlan plan_serial_stream(const ldpc_hip_bp *h) {
    SerialStreamPlan p;
    if (h->m <= 0 || h->n <= 0 || h->nnz >= (1 << 22) || h->n_levels <= 0) return p;
    if (h->serial_kernel != 2 && (double)h->n / (double)h->n_levels < 32.0) return p;  // (narrow levels: the wavefronts would idle at the barriers)
    if (h->regular && h->max_row_deg == 6 && h->max_col_deg == 3 && h->sw("SER_VAR") <= 0) {  // the form built around the (6,3) record
        p.dr = h->max_row_deg;
        p.dc = h->max_col_deg;
        return p;
    }
    // any other degree profile with rows of <= 16 and columns of 1 .. 8 entries: items ("SER_VAR" 0: never -- bp_serial_level_kernel as before round 6)
    if (h->sw("SER_VAR") == 0 || h->max_row_deg > 16 || h->max_col_deg > 8 || h->min_col_deg < 1) return p;
    p.var = true;
    p.dr = h->max_row_deg <= 8 ? 8 : 16;
    p.dc = h->max_col_deg <= 4 ? 4 : 8;
    return p;
}

// wavefronts per tile and units of 1 KiB per wavefront of bp_serial_stream_var_kernel
static void serial_var_geometry(const ldpc_hip_bp *h, int waves_cap, int &waves, int &units) {
    units = h->sw("SER_VAR_UNITS") > 0 ? h->sw("SER_VAR_UNITS") : 8;
    if (units < h->max_row_deg / 2) units = h->max_row_deg / 2;  // (an item must fit the queue)
    if (units < 1) units = 1;
    if (units > 32) units = 32;
    waves = h->sw("SER_WAVES") > 0 ? h->sw("SER_WAVES") : 16;
    if (waves > waves_cap) waves = waves_cap;
    if (waves > 16) waves = 16;
    while (waves > 1 && (size_t)waves * (size_t)units * 1024u > 150u * 1024u) --waves;
}

// item tables of bp_serial_var_kernel.h from the host copies of the levels: the streams of `waves` wavefronts per tile, and the lane
// kernel's level-major list (a position's items never straddle a wavefront)
static int ensure_serial_var_tables(ldpc_hip_bp *h, int waves) {
    if (h->ser_var_valid && h->ser_var_waves == waves) return LDPC_HIP_OK;
    const int n = h->n, m = h->m, L = h->n_levels;
    std::vector<int32_t> col_ptr((size_t)n + 1, 0), col_edge((size_t)(h->nnz ? h->nnz : 1)), row_of((size_t)(h->nnz ? h->nnz : 1));
    for (int e = 0; e < h->nnz; ++e) col_ptr[(size_t)h->h_col_idx[(size_t)e] + 1]++;
    for (int j = 0; j < n; ++j) col_ptr[(size_t)j + 1] += col_ptr[(size_t)j];
    {
        std::vector<int32_t> fill(col_ptr.begin(), col_ptr.end() - 1);
        for (int i = 0; i < m; ++i)
            for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) {  // (CSR order: a column's edges come out rows ascending, the order of its linked list)
                col_edge[(size_t)fill[(size_t)h->h_col_idx[(size_t)e]]++] = e;
                row_of[(size_t)e] = i;
            }
    }
    // which entries an earlier position of the schedule has written, per item of every position (first iteration; the positions of one level
    // share no row, so "earlier" is "in an earlier level" and the dealing below changes nothing): mask over the row's OTHER entries
    std::vector<int32_t> pos_item0((size_t)n + 1, 0);
    for (int p = 0; p < n; ++p) { const int j = h->h_lvl_bits[(size_t)p]; pos_item0[(size_t)p + 1] = pos_item0[(size_t)p] + (col_ptr[(size_t)j + 1] - col_ptr[(size_t)j]); }
    std::vector<int32_t> item_mask((size_t)pos_item0[(size_t)n] + 1, 0);
    {
        std::vector<char> written((size_t)(h->nnz ? h->nnz : 1), 0);
        for (int p = 0; p < n; ++p) {
            const int j = h->h_lvl_bits[(size_t)p], dj = col_ptr[(size_t)j + 1] - col_ptr[(size_t)j];
            for (int k = 0; k < dj; ++k) {
                const int e = col_edge[(size_t)col_ptr[(size_t)j] + k], i = row_of[(size_t)e], rs = h->h_row_ptr[(size_t)i], d = h->h_row_ptr[(size_t)i + 1] - rs;
                int32_t mask = 0;
                for (int t = 0; t < d - 1; ++t) if (written[(size_t)(rs + t + (t >= e - rs ? 1 : 0))]) mask |= 1 << t;
                item_mask[(size_t)pos_item0[(size_t)p] + k] = mask;
            }
            for (int k = 0; k < dj; ++k) written[(size_t)col_edge[(size_t)col_ptr[(size_t)j] + k]] = 1;
        }
    }
    auto record = [&](int32_t *r, int p, int k, bool lane_form) {
        const int j = h->h_lvl_bits[(size_t)p];
        const int e = col_edge[(size_t)col_ptr[(size_t)j] + k], i = row_of[(size_t)e], rs = h->h_row_ptr[(size_t)i], d = h->h_row_ptr[(size_t)i + 1] - rs;
        const int dj = col_ptr[(size_t)j + 1] - col_ptr[(size_t)j];
        r[0] = e; r[1] = rs; r[2] = d | ((e - rs) << 8) | (k << 16) | (dj << 24); r[3] = j; r[4] = i; r[5] = lane_form ? 1 : 0;
        r[6] = item_mask[(size_t)pos_item0[(size_t)p] + k]; r[7] = 0;
    };
    std::vector<int32_t> items, wq((size_t)L * (size_t)waves + 1, 0), lane_items, lane_lvl((size_t)L + 1, 0);
    items.reserve((size_t)h->nnz * SERIAL_VAR_REC);
    for (int l = 0; l < L; ++l) {
        const int p0 = h->h_lvl_ptr[(size_t)l], p1 = h->h_lvl_ptr[(size_t)l + 1];
        // The level's positions dealt to the wavefronts so that every wavefront gets about the same work (the level ends with a barrier:
        // it takes as long as its slowest wavefront): heaviest position first, each to the wavefront with the least so far.  Work of a
        // position ~ segments it fetches + a constant per message (the `log`) and per own entry (the `tanh`, the store).  The positions of a
        // level share no check, so their order changes no result.
        {
            std::vector<std::pair<int, int>> cost;  // (work, position)
            cost.reserve((size_t)(p1 - p0));
            for (int p = p0; p < p1; ++p) {
                const int j = h->h_lvl_bits[(size_t)p];
                int c = 0;
                for (int q = col_ptr[(size_t)j]; q < col_ptr[(size_t)j + 1]; ++q) {
                    const int i = row_of[(size_t)col_edge[(size_t)q]];
                    c += (h->h_row_ptr[(size_t)i + 1] - h->h_row_ptr[(size_t)i] - 1) + 8;
                }
                cost.emplace_back(c, p);
            }
            std::stable_sort(cost.begin(), cost.end(), [](const std::pair<int, int> &x, const std::pair<int, int> &y) { return x.first > y.first; });
            std::vector<std::vector<int>> deal((size_t)waves);
            std::vector<long> load((size_t)waves, 0);
            for (const auto &cp : cost) {
                int best = 0;
                for (int w = 1; w < waves; ++w) if (load[(size_t)w] < load[(size_t)best]) best = w;
                deal[(size_t)best].push_back(cp.second);
                load[(size_t)best] += cp.first;
            }
            for (int w = 0; w < waves; ++w) {
                wq[(size_t)l * waves + w] = (int32_t)(items.size() / SERIAL_VAR_REC);
                for (int p : deal[(size_t)w]) {
                    const int j = h->h_lvl_bits[(size_t)p], dj = col_ptr[(size_t)j + 1] - col_ptr[(size_t)j];
                    for (int k = 0; k < dj; ++k) {
                        items.resize(items.size() + SERIAL_VAR_REC);
                        record(items.data() + items.size() - SERIAL_VAR_REC, p, k, false);
                    }
                }
            }
        }
        lane_lvl[(size_t)l] = (int32_t)(lane_items.size() / SERIAL_VAR_REC);
        for (int p = p0; p < p1; ++p) {
            const int j = h->h_lvl_bits[(size_t)p], dj = col_ptr[(size_t)j + 1] - col_ptr[(size_t)j];
            const size_t at = lane_items.size() / SERIAL_VAR_REC;
            if (at % 64 + (size_t)dj > 64) lane_items.resize((at + 63) / 64 * 64 * SERIAL_VAR_REC, 0);  // padding: the position starts a new wavefront
            for (int k = 0; k < dj; ++k) {
                lane_items.resize(lane_items.size() + SERIAL_VAR_REC);
                record(lane_items.data() + lane_items.size() - SERIAL_VAR_REC, p, k, true);
            }
        }
        lane_items.resize((lane_items.size() / SERIAL_VAR_REC + 63) / 64 * 64 * SERIAL_VAR_REC, 0);  // a level ends on a wavefront boundary
    }
    wq[(size_t)L * waves] = (int32_t)(items.size() / SERIAL_VAR_REC);
    lane_lvl[(size_t)L] = (int32_t)(lane_items.size() / SERIAL_VAR_REC);
    int rc;
    if ((rc = h->ser_var_items.ensure((items.size() + SERIAL_VAR_REC) * sizeof(int32_t))) || (rc = h->ser_var_wq.ensure(wq.size() * sizeof(int32_t))) ||
        (rc = h->ser_var_lane_items.ensure((lane_items.size() + SERIAL_VAR_REC) * sizeof(int32_t))) || (rc = h->ser_var_lane_lvl.ensure(lane_lvl.size() * sizeof(int32_t)))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!items.empty()) HIPCHK(hipMemcpy(h->ser_var_items.p, items.data(), items.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->ser_var_wq.p, wq.data(), wq.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!lane_items.empty()) HIPCHK(hipMemcpy(h->ser_var_lane_items.p, lane_items.data(), lane_items.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->ser_var_lane_lvl.p, lane_lvl.data(), lane_lvl.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->ser_var_valid = true;
    h->ser_var_waves = waves;
    return LDPC_HIP_OK;
}

// the array of initial segments of the item form (bp_serial_var_kernel.h), from the current priors; nullptr in *out where the first iteration
// must find its messages written out (an order that skips bits, or EXPLICIT_INIT)
static int serial_var_init_segments(ldpc_hip_bp *h, const double **out) {
    *out = nullptr;
    if (!h->order_visits_all || h->on("EXPLICIT_INIT") || h->nnz <= 0) return LDPC_HIP_OK;
    int rc;
    if ((rc = h->ser_var_init.ensure(sizeof(double) * (size_t)h->nnz * LDPC_WAVE))) return rc;
    const dim3 g((unsigned)((h->nnz + 3) / 4));
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) hipLaunchKernelGGL((serial_var_init_kernel<LDPC_HIP_MINIMUM_SUM, 0>), g, dim3(256), 0, h->stream, h->d_llr0, h->d_col_idx, h->nnz, (double *)h->ser_var_init.p);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) hipLaunchKernelGGL((serial_var_init_kernel<LDPC_HIP_PRODUCT_SUM, 1>), g, dim3(256), 0, h->stream, h->d_llr0, h->d_col_idx, h->nnz, (double *)h->ser_var_init.p);
    else hipLaunchKernelGGL((serial_var_init_kernel<LDPC_HIP_PRODUCT_SUM, 0>), g, dim3(256), 0, h->stream, h->d_llr0, h->d_col_idx, h->nnz, (double *)h->ser_var_init.p);
    HIPCHK(hipGetLastError());
    *out = (const double *)h->ser_var_init.p;
    return LDPC_HIP_OK;
}

template <int METHOD, int MATH>
static void (*pick_serial_var(const SerialStreamPlan &sp))(const SerialArgs) {
    return sp.dr <= 8 && sp.dc <= 4 ? bp_serial_stream_var_kernel<METHOD, MATH, 8, 4> : bp_serial_stream_var_kernel<METHOD, MATH, 16, 8>;
}
template <int METHOD, int MATH>
static void (*pick_serial_lane_var(const SerialStreamPlan &sp))(const SerialLaneVarArgs) {
    return sp.dr <= 8 && sp.dc <= 4 ? bp_serial_lane_var_kernel<METHOD, MATH, 8, 4> : bp_serial_lane_var_kernel<METHOD, MATH, 16, 8>;
}

// the records of the positions (layout: bp_serial_stream_kernel.h), from the host copies of the levels
static int ensure_serial_stream_table(ldpc_hip_bp *h, const SerialStreamPlan &sp) {
    if (h->ser_pos_valid) return LDPC_HIP_OK;
    const int n = h->n, dr = sp.dr, dc = sp.dc;
    const int rec = SERIAL_STREAM_REC;
    std::vector<int32_t> tab((size_t)n * (size_t)rec, 0);
    std::vector<int32_t> col_fill((size_t)n, 0), col_edges((size_t)n * (size_t)dc, 0);
    for (int e = 0; e < h->nnz; ++e) {  // (CSR order: a column's edges come out rows ascending, the order of its linked list)
        const int j = h->h_col_idx[(size_t)e];
        col_edges[(size_t)j * dc + col_fill[(size_t)j]++] = e;
    }
    std::vector<char> written((size_t)h->nnz, 0);
    for (int p = 0; p < n; ++p) {
        const int j = h->h_lvl_bits[(size_t)p];
        int32_t *r = tab.data() + (size_t)p * rec;
        int t = 0;
        uint32_t mask = 0;
        for (int k = 0; k < dc; ++k) {
            const int e = col_edges[(size_t)j * dc + k], rs = e / dr * dr;
            for (int q = 0; q < dr; ++q)
                if (rs + q != e) {
                    if (written[(size_t)(rs + q)]) mask |= 1u << t;
                    r[t++] = rs + q;
                }
            r[16 + k] = e;
        }
        r[15] = (int32_t)mask;
        r[16 + dc] = j;
        r[16 + dc + 1] = (int32_t)mask;
        // (positions of one level share no row, so marking at once is marking at the end of the level)
        for (int k = 0; k < dc; ++k) written[(size_t)col_edges[(size_t)j * dc + k]] = 1;
    }
    int rc;
    if ((rc = h->ser_pos_tab.ensure(tab.size() * sizeof(int32_t)))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->ser_pos_tab.p, tab.data(), tab.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    h->ser_pos_valid = true;
    return LDPC_HIP_OK;
}

template <int METHOD, int MATH>
static void (*pick_serial_stream(int ring))(const SerialArgs) {
    return ring >= 2 ? bp_serial_stream_kernel<METHOD, MATH, 6, 3, 2> : bp_serial_stream_kernel<METHOD, MATH, 6, 3, 1>;
}

template <int METHOD, int MATH>
static void (*pick_serial_level(int max_row, int max_col))(const SerialArgs) {
    if (max_row <= 4 && max_col <= 2) return bp_serial_level_kernel<METHOD, MATH, 2, 4>;
    if (max_row <= 6 && max_col <= 3) return bp_serial_level_kernel<METHOD, MATH, 3, 6>;
    return bp_serial_level_kernel<METHOD, MATH, 4, 8>;
}

template <int METHOD, int MATH>
static void (*pick_serial(int max_row, int max_col))(const SerialArgs) {
    if (max_row <= 4 && max_col <= 2) return bp_serial_kernel<METHOD, MATH, 2, 4>;
    if (max_row <= 6 && max_col <= 3) return bp_serial_kernel<METHOD, MATH, 3, 6>;
    return bp_serial_kernel<METHOD, MATH, 4, 8>;  // also the variant that streams heavier nodes (SerialArgs::fast == 0)
}

static int decode_serial_pass(ldpc_hip_bp *h, int max_iter, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                              int32_t *iters, uint8_t *conv, const int32_t *orders = nullptr, int n_orders = 0, int orders_first = 0) {
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
    // level-parallel variant when the schedule has at least two bits per level on average (or when asked for)
    int level_waves = 0;
    const bool orders_levels = orders && h->rnd.valid && h->sched_lvl_bits.p && h->sched_lvl_ptr.p;  // (the ring carries levels: random_orders_append)
    if (h->serial_kernel != 0 && h->n > 0 && (!orders || orders_levels)) {
        double per_level;
        if (orders) {  // a schedule that changes per iteration: the levels of every row of the ring
            double sum = 0.0;
            int cnt = 0;
            for (int32_t nl : h->rnd.n_levels) if (nl > 0) { sum += (double)nl; ++cnt; }
            per_level = cnt ? (double)h->n / (sum / cnt) : 0.0;
        } else {
            if ((rc = ensure_serial_levels(h))) return rc;
            per_level = (double)h->n / (double)(h->n_levels ? h->n_levels : 1);
        }
        if (h->serial_kernel == 1 || per_level >= 2.0) {
            level_waves = (int)(per_level + 0.999);
            if (level_waves > 8) level_waves = 8;
            if (level_waves < 1) level_waves = 1;
        }
    }
    // the streamed form (bp_serial_stream_kernel.h) where the matrix and the schedule allow it: automatic, or asked for (mode 2).  (Here: batches
    // whose state is not resident at once, chunk by chunk, one pass each; resident batches go through decode_serial_streamed.)
    SerialStreamPlan sp;
    int ser_ring = 1, ser_waves = 16;
    if (level_waves && !orders && (h->serial_kernel == -1 || h->serial_kernel == 2)) sp = plan_serial_stream(h);
    int var_units = 0;
    if (sp.var) {
        serial_var_geometry(h, 16, ser_waves, var_units);
        if ((rc = ensure_serial_var_tables(h, ser_waves))) return rc;
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_serial_var<LDPC_HIP_MINIMUM_SUM, 0>(sp);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_serial_var<LDPC_HIP_PRODUCT_SUM, 1>(sp);
        else kern = pick_serial_var<LDPC_HIP_PRODUCT_SUM, 0>(sp);
    } else if (sp.dr) {
        if ((rc = ensure_serial_stream_table(h, sp))) return rc;
        if (h->sw("SER_RING") > 0) ser_ring = h->sw("SER_RING") >= 2 ? 2 : 1;
        if (h->sw("SER_WAVES") > 0) ser_waves = h->sw("SER_WAVES");
        const int slot = serial_stream_slot_bytes(sp.dr, sp.dc);
        while (ser_waves > 1 && (size_t)ser_waves * (size_t)(ser_ring * slot + LDPC_NEAR_BYTES) > 150u * 1024u) --ser_waves;
        if (ser_waves > 16) ser_waves = 16;
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_serial_stream<LDPC_HIP_MINIMUM_SUM, 0>(ser_ring);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_serial_stream<LDPC_HIP_PRODUCT_SUM, 1>(ser_ring);
        else kern = pick_serial_stream<LDPC_HIP_PRODUCT_SUM, 0>(ser_ring);
    } else if (level_waves) {
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_serial_level<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_serial_level<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg);
        else kern = pick_serial_level<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg);
    } else if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_serial<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_serial<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg);
    else kern = pick_serial<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg);
    h->accumulated_ms = 0.f;
    h->accumulated_persistent_ms = 0.f;
    h->timed = false;
    h->timed_mid = false;
    hipStream_t st = h->stream;
    for (int64_t t0 = 0; t0 < tiles_total; t0 += chunk) {
        const int64_t tiles = (tiles_total - t0 < chunk) ? tiles_total - t0 : chunk;
        const int64_t b0 = t0 * LDPC_WAVE;
        const int64_t nb = (batch - b0 < tiles * LDPC_WAVE) ? batch - b0 : tiles * LDPC_WAVE;
        HIPCHK(hipMemsetAsync(h->invalid.p, 0, sizeof(uint64_t) * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dec.p, 0, sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dcur.p, 0, sizeof(uint64_t) * (size_t)(h->n ? h->n : 1) * (size_t)tiles, st));
        if (llr && !h->order_visits_all)  // bits the order never visits report 0 (the reference leaves them stale)
            HIPCHK(hipMemsetAsync(h->llr_t.p, 0, per_tile_llr * (size_t)tiles, st));
        if (h->m > 0) {
            dim3 g((unsigned)((h->m + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(pack_syndromes_kernel, g, dim3(256), 0, st, synd + b0 * h->m, nb, h->m,
                               (uint64_t *)h->par.p, (uint64_t *)h->nzm.p, (uint64_t *)h->invalid.p);
        }
        SerialArgs a = {};
        a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = max_iter; a.fast = fast ? 1 : 0;
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
        a.lvl_ptr = (const int32_t *)h->lvl_ptr.p; a.lvl_bits = (const int32_t *)h->lvl_bits.p; a.n_levels = h->n_levels;
        a.orders = orders; a.n_orders = n_orders; a.orders_first = orders_first;
        if (orders && level_waves) { a.orders_lvl = (const int32_t *)h->sched_lvl_bits.p; a.orders_lvl_ptr = (const int32_t *)h->sched_lvl_ptr.p; }
        if (sp.var) {
            a.var_items = (const int32_t *)h->ser_var_items.p; a.var_wq = (const int32_t *)h->ser_var_wq.p; a.var_units = var_units;
            a.clk = h->d_clk;
            if ((rc = serial_var_init_segments(h, &a.var_init))) return rc;
            const size_t dyn = (size_t)ser_waves * (size_t)var_units * 1024u;
            if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3((unsigned)(64 * ser_waves)), (unsigned)dyn, st, a);
        } else if (sp.dr) {
            a.pos_tab = (const int32_t *)h->ser_pos_tab.p;
            a.clk = h->d_clk;
            if (h->order_visits_all && !h->on("EXPLICIT_INIT")) {  // the first iteration reads this table instead of initial messages
                if ((rc = h->d_edge0.ensure(sizeof(double) * (size_t)h->n))) return rc;
                const dim3 ge((unsigned)((h->n + 255) / 256));
                if (h->bp_method == LDPC_HIP_MINIMUM_SUM) hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_MINIMUM_SUM, 0>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
                else if (h->math_mode == LDPC_HIP_MATH_FAST) hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 1>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
                else hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 0>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
                a.edge0 = (const double *)h->d_edge0.p;
                if ((rc = h->ser_pos_e0.ensure(sizeof(double) * 16 * (size_t)h->n))) return rc;
                hipLaunchKernelGGL(serial_pos_e0_kernel, dim3((unsigned)((h->n * 16 + 255) / 256)), dim3(256), 0, st, (const int32_t *)h->ser_pos_tab.p, h->d_col_idx,
                                   (const double *)h->d_edge0.p, h->n, sp.dc * (sp.dr - 1), (double *)h->ser_pos_e0.p);
                a.pos_e0 = (const double *)h->ser_pos_e0.p;
            }
            const size_t dyn = (size_t)ser_waves * (size_t)(ser_ring * serial_stream_slot_bytes(sp.dr, sp.dc) + LDPC_NEAR_BYTES);
            if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
            hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3((unsigned)(64 * ser_waves)), (unsigned)dyn, st, a);
        } else
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3((unsigned)(64 * (level_waves ? level_waves : 1))), 0, st, a);
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


// The serial kernel decodes a 64-syndrome tile with one wavefront, which runs until its slowest lane is done: one
// syndrome that never converges keeps 63 finished ones waiting for max_iter iterations.  Repacking: a first pass with
// few iterations over everything, then the rows it left unconverged -- packed densely into new tiles -- are decoded
// again from the start with the full iteration budget (BP is deterministic: restarting gives what continuing would),
// and their results replace the first pass's.  Work ~ k1 + f * max_iter instead of max_iter (f = unconverged fraction).
// ---- schedules whose order lives in the decoder object and changes while decoding (bp.hpp:467-483) ------------------------
// The reference decodes one syndrome at a time and carries serial_schedule_order (and the shuffle generator) from decode to
// decode.  A batch cannot do that across its rows (where row b starts would depend on how many iterations rows 0 .. b-1
// took), so: EVERY ROW OF A CALL STARTS FROM THE HANDLE'S CURRENT STATE -- what the reference gives with a new decoder
// object per syndrome when the state is the initial one -- and the call leaves the state its LAST row produced.  A batch
// of one row is therefore exactly one BpDecoder::decode, and a sequence of one-row calls is exactly a sequence of decodes
// on one reference object.  Both wait for the device at the end (the state comes back to the host).
// ---- the random schedule's table of per-iteration orders ---------------------------------------------------------------------
// Iteration it of EVERY row of a call walks the handle's order after it rearrangements (std::shuffle on the object's generator,
// rng.hpp:128-130; the soft-syndrome routine applies one fixed rearrangement again and again, bp.hpp:573-577), and the call leaves
// the order and generator of its LAST row.  The table [max_iter][n] lives on the device as a ring: the next call's table is this
// one minus the rows the last row consumed plus as many new ones behind its end, so a call costs as many shuffles as iterations
// actually ran (as in the reference) instead of max_iter of them -- at the reference's default max_iter = n that was n^2 draws and
// a 4 n^2-byte upload per decode.  Any change of the order, generator, seed, max_iter or n from outside rebuilds it.
static void random_orders_shuffle(ldpc_hip_bp *h, int kind, std::vector<int> &v, std::mt19937 &g) {
    if (kind == 0) std::shuffle(v.begin(), v.end(), g);
    else std::shuffle(v.begin(), v.end(), std::default_random_engine(h->sched_seed_raw));
}

// One order cut into levels of mutually check-disjoint POSITIONS (as ensure_serial_levels does for the fixed schedule): `bits` gets the
// order level-major (schedule order inside a level), ptr[0] the number of levels, ptr[1 + l] where level l starts.  Levels belong to
// positions, so an order with repeated bits (a caller's serial_schedule_order being shuffled) is handled like any other.
static void random_orders_levels(ldpc_hip_bp *h, const std::vector<int> &order, int32_t *bits, int32_t *ptr, std::vector<int32_t> &check_level,
                                 std::vector<int32_t> &level) {
    auto &r = h->rnd;
    const int n = r.n;
    std::fill(check_level.begin(), check_level.end(), 0);
    int32_t n_levels = n ? 1 : 0;
    for (int t = 0; t < n; ++t) {
        const int j = order[(size_t)t];
        int32_t l = 1;
        for (int q = r.csc_ptr[(size_t)j]; q < r.csc_ptr[(size_t)j + 1]; ++q) l = std::max(l, check_level[(size_t)r.csc_row[(size_t)q]] + 1);
        for (int q = r.csc_ptr[(size_t)j]; q < r.csc_ptr[(size_t)j + 1]; ++q) check_level[(size_t)r.csc_row[(size_t)q]] = l;
        level[(size_t)t] = l;
        n_levels = std::max(n_levels, l);
    }
    std::fill(ptr, ptr + n + 2, 0);
    ptr[0] = n_levels;
    for (int t = 0; t < n; ++t) ptr[1 + level[(size_t)t]]++;        // (count of level l at ptr[1 + l], shifted into starts below)
    for (int l = 0; l < n_levels; ++l) ptr[2 + l] += ptr[1 + l];
    std::vector<int32_t> fill(ptr + 1, ptr + 1 + n_levels);
    for (int t = 0; t < n; ++t) bits[(size_t)fill[(size_t)level[(size_t)t] - 1]++] = order[(size_t)t];
}

// rows [pos, pos + count) of the ring <- `count` further rearrangements of r.row_end (host staging in blocks); every row also goes up
// level-major with its level bounds, for bp_serial_level_kernel / bp_softinfo_level_kernel
static int random_orders_append(ldpc_hip_bp *h, int pos, int count) {
    auto &r = h->rnd;
    const size_t n = (size_t)r.n;
    const bool with_levels = h->sched_lvl_bits.p && h->sched_lvl_ptr.p;
    const int block = (int)std::max<size_t>(1, std::min<size_t>((size_t)count, ((size_t)1 << 22) / (n ? n : 1)));
    std::vector<int32_t> stage((size_t)block * n), stage_bits(with_levels ? (size_t)block * n : 0), stage_ptr(with_levels ? (size_t)block * (n + 2) : 0);
    std::vector<int32_t> check_level(with_levels ? (size_t)(h->m ? h->m : 1) : 0), level(with_levels ? (n ? n : 1) : 0);
    for (int done = 0; done < count;) {
        const int now = std::min(block, count - done);
        for (int q = 0; q < now; ++q) {
            random_orders_shuffle(h, r.kind, r.row_end, r.rng_end);
            std::copy(r.row_end.begin(), r.row_end.end(), stage.begin() + (size_t)q * n);
            if (with_levels) {
                random_orders_levels(h, r.row_end, stage_bits.data() + (size_t)q * n, stage_ptr.data() + (size_t)q * (n + 2), check_level, level);
                r.n_levels[(size_t)((pos + done + q) % r.rows)] = stage_ptr[(size_t)q * (n + 2)];
            }
        }
        for (int q = 0; q < now;) {  // (the ring may wrap inside a block)
            const int at = (pos + done + q) % r.rows;
            const int run = std::min(now - q, r.rows - at);
            HIPCHK(hipMemcpy((int32_t *)h->sched_orders.p + (size_t)at * n, stage.data() + (size_t)q * n, (size_t)run * n * sizeof(int32_t), hipMemcpyHostToDevice));
            if (with_levels) {
                HIPCHK(hipMemcpy((int32_t *)h->sched_lvl_bits.p + (size_t)at * n, stage_bits.data() + (size_t)q * n, (size_t)run * n * sizeof(int32_t), hipMemcpyHostToDevice));
                HIPCHK(hipMemcpy((int32_t *)h->sched_lvl_ptr.p + (size_t)at * (n + 2), stage_ptr.data() + (size_t)q * (n + 2), (size_t)run * (n + 2) * sizeof(int32_t), hipMemcpyHostToDevice));
            }
            q += run;
        }
        done += now;
    }
    return LDPC_HIP_OK;
}

// the table of this call, current on the device (h->stream is idle afterwards)
static int random_orders_prepare(ldpc_hip_bp *h, int kind) {
    auto &r = h->rnd;
    const int n = h->n, rows = h->max_iter;
    if ((size_t)rows * (size_t)(n ? n : 1) > ((size_t)1 << 28))
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "random serial schedule: max_iter x n = %d x %d orders exceed the 1 GiB table of per-iteration orders; lower max_iter", rows, n);
    int rc;
    if ((rc = h->sched_orders.ensure((size_t)rows * (size_t)n * sizeof(int32_t) + 16))) return rc;  // (+16: max_iter = 0 leaves the table empty)
    if ((rc = h->sched_lvl_bits.ensure((size_t)rows * (size_t)n * sizeof(int32_t) + 16)) ||
        (rc = h->sched_lvl_ptr.ensure((size_t)rows * ((size_t)n + 2) * sizeof(int32_t) + 16))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));  // a previous launch may still read the table
    const bool current = r.valid && r.kind == kind && r.rows == rows && r.n == n && r.expect_state == h->sched_state &&
                         (kind == 0 ? r.expect_rng == h->sched_rng : r.seed_raw == h->sched_seed_raw);
    if (current || rows == 0 || n == 0) return LDPC_HIP_OK;
    r.valid = false;
    r.kind = kind; r.rows = rows; r.n = n; r.first = 0; r.seed_raw = h->sched_seed_raw;
    r.n_levels.assign((size_t)rows, 0);
    {   // the checks of every bit, for the levels
        r.csc_ptr.assign((size_t)n + 1, 0);
        for (int e = 0; e < h->nnz; ++e) r.csc_ptr[(size_t)h->h_col_idx[(size_t)e] + 1]++;
        for (int j = 0; j < n; ++j) r.csc_ptr[(size_t)j + 1] += r.csc_ptr[(size_t)j];
        r.csc_row.assign((size_t)(h->nnz ? h->nnz : 1), 0);
        std::vector<int32_t> at(r.csc_ptr.begin(), r.csc_ptr.end() - 1);
        for (int i = 0; i < h->m; ++i)
            for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) r.csc_row[(size_t)at[(size_t)h->h_col_idx[(size_t)e]]++] = i;
    }
    r.row_end.assign(h->sched_state.begin(), h->sched_state.end());
    r.rng_end = h->sched_rng;
    if ((rc = random_orders_append(h, 0, rows))) return rc;
    r.expect_state = h->sched_state;
    r.expect_rng = h->sched_rng;
    r.valid = true;
    return LDPC_HIP_OK;
}

// after the call: its last row ran `last` iterations -- the handle's order and generator move on by as many rearrangements, the
// ring drops those rows and grows as many behind its end
static int random_orders_consume(ldpc_hip_bp *h, int kind, int last) {
    auto &r = h->rnd;
    if (last > h->max_iter) last = h->max_iter;
    if (last <= 0 || h->n == 0) return LDPC_HIP_OK;
    std::vector<int> v(h->sched_state.begin(), h->sched_state.end());
    for (int it = 0; it < last; ++it) random_orders_shuffle(h, kind, v, h->sched_rng);
    std::copy(v.begin(), v.end(), h->sched_state.begin());
    if (!r.valid) return LDPC_HIP_OK;
    r.valid = false;
    int rc;
    if ((rc = random_orders_append(h, r.first, last))) return rc;
    r.first = (r.first + last) % r.rows;
    r.expect_state = h->sched_state;
    r.expect_rng = h->sched_rng;
    r.valid = true;
    return LDPC_HIP_OK;
}

static int decode_serial_random(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                                int32_t *iters, uint8_t *conv) {
    int rc;
    if ((rc = random_orders_prepare(h, 0))) return rc;
    if (!iters) { if ((rc = h->sp_iters.ensure((size_t)batch * 4))) return rc; iters = (int32_t *)h->sp_iters.p; }
    if ((rc = decode_serial_pass(h, h->max_iter, synd, batch, decoding, llr, iters, conv, (const int32_t *)h->sched_orders.p, h->max_iter, h->rnd.first))) return rc;
    int32_t last = 0;
    HIPCHK(hipMemcpyAsync(&last, iters + (batch - 1), sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return random_orders_consume(h, 0, last);  // the last row consumed `last` shuffles
}

// serial_relative with a syndrome's whole state in LDS, one wavefront per syndrome (bp_relative_lds_kernel.h): codes with columns of
// <= 8 and rows of <= 16 entries whose state leaves room for at least four wavefronts per compute unit; LDPC_HIP_REL_LDS=0 keeps the
// per-lane kernel (A/B, tests).  Returns 1 if the batch was decoded here, 0 if the caller should carry on, < 0 on error.
// the record of an entry beyond its column's weight, as bp_relative_lds_kernel builds it from the zero-filled tables: edge 0 of check 0
static unsigned long long m_pad_word(const ldpc_hip_bp *h) {
    const unsigned long long rs = (unsigned long long)h->h_row_ptr[0], rd = (unsigned long long)(h->h_row_ptr[1] - h->h_row_ptr[0]);
    return 0ull | (rs << 16) | (rd << 32);
}

static int decode_serial_relative_lds(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                                      int32_t *iters, uint8_t *conv) {
    if (h->sw("REL_LDS") == 0 || h->m <= 0 || h->n <= 0 || h->nnz <= 0) return 0;
    if (h->max_col_deg > 8 || h->max_row_deg > 16 || h->n >= 65535 || h->nnz >= 65535 || h->m >= 65535) return 0;
    const bool ps = h->bp_method == LDPC_HIP_PRODUCT_SUM;
    const int dc = h->max_col_deg;
    size_t shared = rel_lds_shared(h->m, h->n, h->nnz, dc, ps), per_syn = rel_lds_per_syndrome(h->m, h->n, h->nnz, dc);
    size_t scratch = rel_lds_scratch(h->n, dc, false);
    const size_t lds = 160u * 1024u - 64u;
    // State beyond LDS (round 6): where everything in LDS leaves fewer than four wavefronts per compute unit, the messages (the bulk of a
    // syndrome's state) and the per-entry records (the bulk of the shared tables) move to global memory -- bp_relative_lds_kernel<..., EXT = 1>.
    bool ext = h->sw("REL_EXT") > 0 || (h->sw("REL_EXT") != 0 && shared + 4 * (per_syn + scratch) > lds);
    if (h->max_row_deg <= 4 || dc <= 2) ext = false;  // (instantiated for rows of 5 .. 16 and columns of 3 .. 8 entries)
    if (ext) {
        shared = rel_lds_shared(h->m, h->n, h->nnz, dc, ps, true);
        per_syn = rel_lds_per_syndrome(h->m, h->n, h->nnz, dc, true);
    }
    // The sweep goes level by level (bp_relative_lds_kernel.h) when "the position of bit b" is one place: the order must be a permutation
    // of the bits (it is, unless the caller gave a serial_schedule_order with repeats); LDPC_HIP_REL_LEVELS=0 walks bit by bit (A/B, tests).
    bool levels = h->sw("REL_LEVELS") != 0 && (int)h->sched_state.size() == h->n;
    if (levels) {
        std::vector<char> seen((size_t)h->n, 0);
        for (int t = 0; levels && t < h->n; ++t) {
            const int b = h->sched_state[(size_t)t];
            if (b < 0 || b >= h->n || seen[(size_t)b]) levels = false; else seen[(size_t)b] = 1;
        }
    }
    // lanes per syndrome: 64 with levels (a level's bits fill the lanes).  Bit by bit, product-sum is bound by the instruction stream of
    // one log + one tanh per bit, which four syndromes can share (GS = 16) where their state leaves room for a few wavefronts per compute
    // unit; min-sum waits on LDS rather than on the vector unit: one syndrome per wavefront.  LDPC_HIP_REL_LDS = 64 / 16 forces a form
    // (16 implies bit by bit; 32 lanes per syndrome measured in between -- profiles/r4_stateful_schedules.jsonl -- and is not built).
    int gs = 64;
    if (!ext && !levels && ps && shared + 4 * (4 * per_syn + scratch) <= lds) gs = 16;
    if (!ext && (h->sw("REL_LDS") == 16 || h->sw("REL_LDS") == 64)) gs = h->sw("REL_LDS");
    if (gs == 16) levels = false;
    // level by level every bit's posterior is rewritten by the sweep: the array is free from the sort's ranks to the sweep and houses most of the
    // scratch (rel_lds_scratch) -- more wavefronts per compute unit.  LDPC_HIP_REL_SCRATCH_IN_L=0: separate scratch (A/B, tests).
    const bool in_l = !ext && levels && gs == 64 && h->n >= 16 && h->n <= 512 && h->sw("REL_SCRATCH_IN_L") != 0;
    if (in_l) scratch = rel_lds_scratch(h->n, dc, true);
    const int G = 64 / gs;
    const size_t per_wave = (size_t)G * per_syn + scratch;
    // (round 6: ONE wavefront per compute unit is enough -- measured on the [[1600,64]] hypergraph-product code, whose 61 KB of state per
    // syndrome + 93 KB of tables leave exactly that: 60.7 k syndromes/s against the per-lane kernel's 7.0 k, profiles/r6_serial_relative_hgp1600.txt)
    if (shared + (gs == 64 ? 1 : 4) * per_wave > lds) {
        if (gs != 64 && shared + (per_syn + scratch) <= lds) gs = 64; else return 0;
    }
    const int Gf = 64 / gs;
    const size_t per_wave_f = (size_t)Gf * per_syn + scratch;
    int waves = (int)((lds - shared) / per_wave_f);
    if (waves > 16) waves = 16;
    if (ext && waves > 8) waves = 8;  // (the EXT instantiations are compiled for workgroups of at most 8 wavefronts: 256 VGPRs, no spills)
    // several workgroups per compute unit where the state is small: each pays for its own copy of the shared tables
    int groups_per_cu = 1;
    while (waves * (groups_per_cu + 1) <= 24 && (size_t)(groups_per_cu + 1) * (shared + (size_t)waves * per_wave_f) <= lds) ++groups_per_cu;
    int rc;
    if (h->rl_dc != dc) {
        std::vector<uint16_t> te((size_t)h->n * dc, 0), tc((size_t)h->n * dc, 0);
        std::vector<uint8_t> cd((size_t)h->n, 0);
        for (int i = 0; i < h->m; ++i)
            for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
                const int j = h->h_col_idx[(size_t)e], k = cd[(size_t)j]++;
                te[(size_t)j * dc + k] = (uint16_t)e;   // (rows ascending: the order of the column's linked list, sparse_matrix_base.hpp:423-482)
                tc[(size_t)j * dc + k] = (uint16_t)i;
            }
        if ((rc = h->rl_edge.ensure(te.size() * 2)) || (rc = h->rl_chk.ensure(tc.size() * 2)) || (rc = h->rl_cdeg.ensure(cd.size())) ||
            (rc = h->rl_last.ensure((size_t)h->n * sizeof(int32_t)))) return rc;
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->rl_edge.p, te.data(), te.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->rl_chk.p, tc.data(), tc.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(h->rl_cdeg.p, cd.data(), cd.size(), hipMemcpyHostToDevice));
        h->rl_dc = dc;
    }
    if ((rc = h->sched_order0.ensure((size_t)h->n * sizeof(int32_t))) || (rc = h->counter.ensure(16 + 14 * 8))) return rc;
    hipStream_t st = h->stream;
    HIPCHK(hipStreamSynchronize(st));
    if (ext && !h->rl_rec_valid) {  // the records the kernel otherwise builds in LDS: CSR edge | row start << 16 | row weight << 32 | check << 48 per (bit, entry of its column)
        std::vector<unsigned long long> rec((size_t)h->n * dc, 0ull);
        std::vector<uint8_t> cd((size_t)h->n, 0);
        for (int i = 0; i < h->m; ++i) {
            const unsigned long long rs = (unsigned long long)h->h_row_ptr[(size_t)i], rd = (unsigned long long)(h->h_row_ptr[(size_t)i + 1] - h->h_row_ptr[(size_t)i]);
            for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
                const int j = h->h_col_idx[(size_t)e], k = cd[(size_t)j]++;
                rec[(size_t)j * dc + k] = (unsigned long long)e | (rs << 16) | (rd << 32) | ((unsigned long long)i << 48);
            }
        }
        // (entries beyond a column's weight: the kernel's words for t_chk = t_edge = 0 -- check 0's row)
        const unsigned long long pad = m_pad_word(h);
        for (int j = 0; j < h->n; ++j)
            for (int k = cd[(size_t)j]; k < dc; ++k) rec[(size_t)j * dc + k] = pad;
        if ((rc = h->rl_rec.ensure(rec.size() * 8))) return rc;
        HIPCHK(hipMemcpy(h->rl_rec.p, rec.data(), rec.size() * 8, hipMemcpyHostToDevice));
        h->rl_rec_valid = true;
    }
    HIPCHK(hipMemcpy(h->sched_order0.p, h->sched_state.data(), (size_t)h->n * sizeof(int32_t), hipMemcpyHostToDevice));
    HIPCHK(hipMemsetAsync(h->counter.p, 0, 16 + 14 * 8, st));
    RelLdsArgs a = {};
    a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter; a.dc = dc;
    a.ms_scaling_factor = h->ms_scaling_factor;
    a.batch = batch;
    a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx;
    a.t_edge = (const uint16_t *)h->rl_edge.p; a.t_chk = (const uint16_t *)h->rl_chk.p; a.t_cdeg = (const uint8_t *)h->rl_cdeg.p;
    a.order0 = (const int32_t *)h->sched_order0.p;
    a.llr0 = h->d_llr0;
    a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    a.last_order = (int32_t *)h->rl_last.p;
    a.next = (unsigned long long *)h->counter.p;
    a.lds_shared = (int32_t)shared; a.lds_per_syn = (int32_t)per_syn; a.lds_scratch = (int32_t)scratch;
    a.levels = levels ? 1 : 0;
    a.scratch_in_l = in_l ? 1 : 0;
    a.clk = h->d_clk;
    a.prof = h->sw("REL_PROF") > 0 ? (unsigned long long *)((char *)h->counter.p + 16) : nullptr;
    void (*kern)(const RelLdsArgs);
#define LDPC_PICK_REL_G(M, F, GSZ, DCT) (h->max_row_deg <= 4 ? bp_relative_lds_kernel<M, F, 4, GSZ, DCT> : h->max_row_deg <= 8 ? bp_relative_lds_kernel<M, F, 8, GSZ, DCT> : bp_relative_lds_kernel<M, F, 16, GSZ, DCT>)
#define LDPC_PICK_REL_X(M, F) (h->max_row_deg <= 8 ? (dc <= 4 ? bp_relative_lds_kernel<M, F, 8, 64, 4, 1> : bp_relative_lds_kernel<M, F, 8, 64, 8, 1>) \
                                                   : (dc <= 4 ? bp_relative_lds_kernel<M, F, 16, 64, 4, 1> : bp_relative_lds_kernel<M, F, 16, 64, 8, 1>))
#define LDPC_PICK_REL(M, F) (ext ? LDPC_PICK_REL_X(M, F) : gs == 16 ? LDPC_PICK_REL_G(M, F, 16, 8) : dc <= 2 ? LDPC_PICK_REL_G(M, F, 64, 2) : dc <= 4 ? LDPC_PICK_REL_G(M, F, 64, 4) : LDPC_PICK_REL_G(M, F, 64, 8))
    if (!ps) kern = LDPC_PICK_REL(LDPC_HIP_MINIMUM_SUM, 0);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = LDPC_PICK_REL(LDPC_HIP_PRODUCT_SUM, 1);
    else kern = LDPC_PICK_REL(LDPC_HIP_PRODUCT_SUM, 0);
#undef LDPC_PICK_REL
#undef LDPC_PICK_REL_X
#undef LDPC_PICK_REL_G
    const size_t dyn = shared + (size_t)waves * per_wave_f;
    if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    {   // the first iteration's sort, once per call (every row starts from the same order with the same keys): bp_relative_lds_kernel.h
        const size_t n1 = (size_t)h->n;
        const size_t pre = n1 * 8 + ((n1 * 2 + 15) & ~(size_t)15) + rel_lds_scratch(h->n, dc, false);
        if (h->sw("REL_FIRST_ONCE") != 0 && batch > 1 && h->max_iter > 0 && pre <= 150u * 1024u && (int)h->sched_state.size() == h->n) {
            if ((rc = h->rl_first.ensure(n1 * sizeof(int32_t)))) return rc;
            if (pre > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)rel_first_order_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pre));
            hipLaunchKernelGGL(rel_first_order_kernel, dim3(1), dim3(64), (unsigned)pre, st, h->d_llr0, (const int32_t *)h->sched_order0.p, h->n, (int32_t *)h->rl_first.p);
            HIPCHK(hipGetLastError());
            a.first_order = (const int32_t *)h->rl_first.p;
        }
    }
    int64_t groups = (batch + (int64_t)waves * Gf - 1) / ((int64_t)waves * Gf);
    if (groups > 256 * (int64_t)groups_per_cu) groups = 256 * (int64_t)groups_per_cu;
    if (ext) {
        if ((rc = h->rl_ext_A.ensure((size_t)groups * (size_t)waves * ((size_t)h->nnz + (size_t)h->n) * sizeof(double)))) return rc;
        a.A_g = (double *)h->rl_ext_A.p;
        a.rec_g = (const unsigned long long *)h->rl_rec.p;
        if (ps) {  // the edge form of the priors (what the messages start from): a table in global memory here
            if ((rc = h->d_edge0.ensure(sizeof(double) * (size_t)h->n))) return rc;
            const dim3 ge((unsigned)((h->n + 255) / 256));
            if (h->math_mode == LDPC_HIP_MATH_FAST) hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 1>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
            else hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 0>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
            HIPCHK(hipGetLastError());
            a.pform_g = (const double *)h->d_edge0.p;
        }
    }
    h->accumulated_ms = 0.f;
    h->accumulated_persistent_ms = 0.f;
    h->timed_mid = false;
    h->timed_prev = h->timed_prev_mid = false;
    HIPCHK(hipEventRecord(h->ev0, st));
    hipLaunchKernelGGL(kern, dim3((unsigned)groups), dim3((unsigned)(waves * 64)), (unsigned)dyn, st, a);
    HIPCHK(hipEventRecord(h->ev1, st));
    h->timed = true;
    HIPCHK(hipGetLastError());
    // the order the LAST row ended with becomes the object's serial_schedule_order
    HIPCHK(hipMemcpyAsync(h->sched_state.data(), h->rl_last.p, (size_t)h->n * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (a.prof) {  // LDPC_HIP_REL_PROF=1: where the wavefronts' cycles went (tools/profile_serial_relative.sh)
        unsigned long long v[14];
        HIPCHK(hipMemcpy(v, a.prof, sizeof v, hipMemcpyDeviceToHost));
        const double tot = v[8] ? (double)v[8] : 1.0, its = v[6] ? (double)v[6] : 1.0;
        std::fprintf(stderr, "[rel_lds gs=%d levels=%d waves/group=%d groups=%lld] cycles per wavefront-iteration %.0f: refill %.1f%% sort %.1f%% levels %.1f%% sweep %.1f%% test %.1f%% out %.1f%%; "
                     "levels per iteration %.1f; wavefront-iterations %llu; of the sort: ranks %.1f%% partitions %.1f%% final pass %.1f%%, partitions per sort %.1f (of more than 64 places: %.1f)\n", gs, a.levels, waves, (long long)groups, tot / its, 100.0 * v[0] / tot, 100.0 * v[1] / tot, 100.0 * v[2] / tot,
                     100.0 * v[3] / tot, 100.0 * v[4] / tot, 100.0 * v[5] / tot, (double)v[7] / its, v[6], 100.0 * v[10] / tot, 100.0 * v[11] / tot, 100.0 * v[12] / tot, (double)(v[13] & 0xffffffffull) / its, (double)(v[13] >> 32) / its);
    }
    return 1;
}

static int decode_serial_relative(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                                  int32_t *iters, uint8_t *conv) {
    {
        const int took = decode_serial_relative_lds(h, synd, batch, decoding, llr, iters, conv);
        if (took < 0) return took;
        if (took > 0) return LDPC_HIP_OK;
    }
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    const size_t n1 = (size_t)(h->n ? h->n : 1), m1 = (size_t)(h->m ? h->m : 1);
    const size_t per_tile_msg = sizeof(double) * (size_t)(h->nnz ? h->nnz : 1) * LDPC_WAVE;
    int64_t chunk = tiles_total;
    if (h->max_chunk_tiles > 0 && chunk > h->max_chunk_tiles) chunk = h->max_chunk_tiles;
    if (chunk > 32768) chunk = 32768;
    {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t have = h->msgA.cap + h->msgC.cap + h->llr_t.cap + h->rel_ord.cap + h->rel_dbit.cap;
        const size_t per_tile = 2 * per_tile_msg + n1 * LDPC_WAVE * (8 + 4 + 1) + 24 * (m1 + n1);
        int64_t fit = (int64_t)((double)(free_b + have) * 0.85 / (double)per_tile);
        if (fit < 1) return fail(LDPC_HIP_ERR_NOMEM, "not enough device memory for one 64-syndrome tile");
        if (chunk > fit) chunk = fit;
    }
    int rc;
    if ((rc = h->msgA.ensure(per_tile_msg * (size_t)chunk)) || (rc = h->msgC.ensure(per_tile_msg * (size_t)chunk)) ||
        (rc = h->llr_t.ensure(n1 * LDPC_WAVE * 8 * (size_t)chunk)) || (rc = h->rel_ord.ensure(n1 * LDPC_WAVE * 4 * (size_t)chunk)) ||
        (rc = h->rel_dbit.ensure(n1 * LDPC_WAVE * (size_t)chunk)) || (rc = h->par.ensure(sizeof(uint64_t) * m1 * (size_t)chunk)) ||
        (rc = h->nzm.ensure(sizeof(uint64_t) * m1 * (size_t)chunk)) || (rc = h->invalid.ensure(sizeof(uint64_t) * (size_t)chunk)) ||
        (rc = h->sched_order0.ensure(n1 * sizeof(int32_t)))) return rc;
    hipStream_t st = h->stream;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipMemcpy(h->sched_order0.p, h->sched_state.data(), (size_t)h->n * sizeof(int32_t), hipMemcpyHostToDevice));
    void (*kern)(const RelArgs);
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = bp_serial_relative_kernel<LDPC_HIP_MINIMUM_SUM, 0>;
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = bp_serial_relative_kernel<LDPC_HIP_PRODUCT_SUM, 1>;
    else kern = bp_serial_relative_kernel<LDPC_HIP_PRODUCT_SUM, 0>;
    h->accumulated_ms = 0.f;
    h->accumulated_persistent_ms = 0.f;
    h->timed = false;
    h->timed_mid = false;
    int64_t last_tiles = 0;
    for (int64_t t0 = 0; t0 < tiles_total; t0 += chunk) {
        const int64_t tiles = (tiles_total - t0 < chunk) ? tiles_total - t0 : chunk;
        const int64_t b0 = t0 * LDPC_WAVE;
        const int64_t nb = (batch - b0 < tiles * LDPC_WAVE) ? batch - b0 : tiles * LDPC_WAVE;
        HIPCHK(hipMemsetAsync(h->invalid.p, 0, sizeof(uint64_t) * (size_t)tiles, st));
        if (h->m > 0) {
            dim3 g((unsigned)((h->m + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(pack_syndromes_kernel, g, dim3(256), 0, st, synd + b0 * h->m, nb, h->m,
                               (uint64_t *)h->par.p, (uint64_t *)h->nzm.p, (uint64_t *)h->invalid.p);
        }
        RelArgs a = {};
        a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter;
        a.ms_scaling_factor = h->ms_scaling_factor;
        a.batch = nb;
        a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx; a.col_ptr = h->d_col_ptr; a.csc_edge = h->d_csc_edge; a.csc_row = h->d_csc_row;
        a.order0 = (const int32_t *)h->sched_order0.p;
        a.llr0 = h->d_llr0;
        a.A = (double *)h->msgA.p; a.C = (double *)h->msgC.p; a.llr_t = (double *)h->llr_t.p;
        a.ord = (int32_t *)h->rel_ord.p; a.dbit = (uint8_t *)h->rel_dbit.p;
        a.par = (const uint64_t *)h->par.p; a.invalid = (const uint64_t *)h->invalid.p;
        a.decoding = decoding + b0 * h->n;
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
        if (llr && h->n > 0) {
            dim3 gt((unsigned)((h->n + LDPC_WAVE - 1) / LDPC_WAVE), (unsigned)tiles);
            hipLaunchKernelGGL(transpose_llr_kernel, gt, dim3(256), 0, st, (const double *)h->llr_t.p, nb, h->n, llr + (size_t)b0 * h->n);
        }
        HIPCHK(hipGetLastError());
        last_tiles = tiles;
    }
    // the order the LAST row ended with becomes the object's serial_schedule_order: column (last lane) of the last tile's ord
    if (h->n > 0) {
        const int64_t lane = (batch - 1) % LDPC_WAVE;
        const int32_t *src = (const int32_t *)h->rel_ord.p + (size_t)(last_tiles - 1) * n1 * LDPC_WAVE + (size_t)lane;
        HIPCHK(hipMemcpy2DAsync(h->sched_state.data(), sizeof(int32_t), src, sizeof(int32_t) * LDPC_WAVE, sizeof(int32_t), (size_t)h->n,
                                hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipStreamSynchronize(st));
    return LDPC_HIP_OK;
}

// ---- the streamed serial schedule (bp_serial_stream_kernel.h) over a batch whose message state is resident ----------------------------
// One launch of bp_serial_stream_kernel: iterations it_start + 1 .. it_end over `rows` rows whose state is (it_start > 0) or will be in
// `state`.  resume: the same tiles as the launch before (their packed syndromes, frozen decisions and posteriors are still in the
// workspace; lanes it finished stay finished).
// row_map (or nullptr): row r of the launch is row row_map[r] of `decoding` / `llr`; `iters` / `conv` are indexed by the launch's own rows.
static int serial_stream_launch(ldpc_hip_bp *h, const SerialStreamPlan &sp, int it_start, int it_end, bool resume, double *state, const uint8_t *synd,
                                int64_t rows, const int32_t *row_map, uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv, int waves_cap = 16) {
    const int64_t tiles = (rows + LDPC_WAVE - 1) / LDPC_WAVE;
    const size_t m1 = (size_t)h->m, n1 = (size_t)h->n;
    hipStream_t st = h->stream;
    int rc;
    if ((rc = h->par.ensure(sizeof(uint64_t) * m1 * (size_t)tiles)) || (rc = h->nzm.ensure(sizeof(uint64_t) * m1 * (size_t)tiles)) ||
        (rc = h->invalid.ensure(sizeof(uint64_t) * (size_t)tiles)) || (rc = h->dec.ensure(sizeof(uint64_t) * n1 * (size_t)tiles)) ||
        (rc = h->dcur.ensure(sizeof(uint64_t) * n1 * (size_t)tiles)) || (llr && (rc = h->llr_t.ensure(sizeof(double) * n1 * LDPC_WAVE * (size_t)tiles)))) return rc;
    if (!resume) {
        HIPCHK(hipMemsetAsync(h->invalid.p, 0, sizeof(uint64_t) * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dec.p, 0, sizeof(uint64_t) * n1 * (size_t)tiles, st));
        HIPCHK(hipMemsetAsync(h->dcur.p, 0, sizeof(uint64_t) * n1 * (size_t)tiles, st));
        if (llr && !h->order_visits_all) HIPCHK(hipMemsetAsync(h->llr_t.p, 0, sizeof(double) * n1 * LDPC_WAVE * (size_t)tiles, st));  // bits the order never visits report 0
        dim3 g((unsigned)((h->m + 255) / 256), (unsigned)(tiles < 32768 ? tiles : 32768));
        hipLaunchKernelGGL(pack_syndromes_kernel, g, dim3(256), 0, st, synd, rows, h->m, (uint64_t *)h->par.p, (uint64_t *)h->nzm.p, (uint64_t *)h->invalid.p,
                           (const int32_t *)nullptr, (const unsigned *)nullptr);
    }
    int ser_ring = 1, ser_waves = 16, var_units = 0;
    void (*kern)(const SerialArgs);
    int slot = 0;
    if (sp.var) {
        // (one table for every pass of a decode: the streams are cut for a number of wavefronts, so a pass asked to use fewer keeps the table's)
        serial_var_geometry(h, 16, ser_waves, var_units);
        (void)waves_cap;
        if ((rc = ensure_serial_var_tables(h, ser_waves))) return rc;
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_serial_var<LDPC_HIP_MINIMUM_SUM, 0>(sp);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_serial_var<LDPC_HIP_PRODUCT_SUM, 1>(sp);
        else kern = pick_serial_var<LDPC_HIP_PRODUCT_SUM, 0>(sp);
    } else {
        if (h->sw("SER_RING") > 0) ser_ring = h->sw("SER_RING") >= 2 ? 2 : 1;
        if (h->sw("SER_WAVES") > 0) ser_waves = h->sw("SER_WAVES");
        if (ser_waves > waves_cap) ser_waves = waves_cap;
        slot = serial_stream_slot_bytes(sp.dr, sp.dc);
        while (ser_waves > 1 && (size_t)ser_waves * (size_t)(ser_ring * slot + LDPC_NEAR_BYTES) > 150u * 1024u) --ser_waves;
        if (ser_waves > 16) ser_waves = 16;
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_serial_stream<LDPC_HIP_MINIMUM_SUM, 0>(ser_ring);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_serial_stream<LDPC_HIP_PRODUCT_SUM, 1>(ser_ring);
        else kern = pick_serial_stream<LDPC_HIP_PRODUCT_SUM, 0>(ser_ring);
    }
    SerialArgs a = {};
    a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = it_end; a.fast = 1;
    a.ms_scaling_factor = h->ms_scaling_factor;
    a.batch = rows;
    a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx; a.col_ptr = h->d_col_ptr; a.csc_edge = h->d_csc_edge; a.csc_row = h->d_csc_row;
    a.llr0 = h->d_llr0;
    a.A = state;
    a.par = (const uint64_t *)h->par.p; a.invalid = (const uint64_t *)h->invalid.p;
    a.dec = (uint64_t *)h->dec.p; a.dcur = (uint64_t *)h->dcur.p;
    a.llr_t = llr ? (double *)h->llr_t.p : nullptr;
    a.iters = iters; a.conv = conv;
    a.lvl_ptr = (const int32_t *)h->lvl_ptr.p; a.lvl_bits = (const int32_t *)h->lvl_bits.p; a.n_levels = h->n_levels;
    a.pos_tab = (const int32_t *)h->ser_pos_tab.p;
    a.clk = h->d_clk;
    a.it_start = it_start;
    a.resume = resume ? 1 : 0;
    if (sp.var) {
        a.var_items = (const int32_t *)h->ser_var_items.p; a.var_wq = (const int32_t *)h->ser_var_wq.p; a.var_units = var_units;
        if (it_start == 0 && (rc = serial_var_init_segments(h, &a.var_init))) return rc;
    } else if (it_start == 0 && h->order_visits_all && !h->on("EXPLICIT_INIT")) {  // the first iteration reads these tables instead of initial messages
        a.edge0 = (const double *)h->d_edge0.p;
        a.pos_e0 = (const double *)h->ser_pos_e0.p;
    }
    const size_t dyn = sp.var ? (size_t)ser_waves * (size_t)var_units * 1024u : (size_t)ser_waves * (size_t)(ser_ring * slot + LDPC_NEAR_BYTES);
    if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3((unsigned)(64 * ser_waves)), (unsigned)dyn, st, a);
    HIPCHK(hipGetLastError());
    {   // (these kernels loop over the tiles beyond their grid)
        const unsigned gy = (unsigned)(tiles < 32768 ? tiles : 32768);
        hipLaunchKernelGGL(unpack_decoding_kernel, dim3((unsigned)((h->n + 255) / 256), gy), dim3(256), 0, st, (const uint64_t *)h->dec.p, rows, h->n, decoding, row_map,
                           (const unsigned *)nullptr);
        if (llr)
            hipLaunchKernelGGL(transpose_llr_kernel, dim3((unsigned)((h->n + LDPC_WAVE - 1) / LDPC_WAVE), gy), dim3(256), 0, st, (const double *)h->llr_t.p, rows, h->n, llr, row_map,
                               (const unsigned *)nullptr);
    }
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// A handful of rows on bp_serial_lane_kernel (one workgroup per syndrome): iterations it_start + 1 .. max_iter; `state_rows`: [rows][nnz]
static int serial_lane_launch(ldpc_hip_bp *h, const SerialStreamPlan &sp, int it_start, double *state_rows, const uint8_t *synd, int64_t rows,
                              uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (sp.var) {
        int waves = 0, units = 0, rc;
        serial_var_geometry(h, 16, waves, units);
        if ((rc = ensure_serial_var_tables(h, waves))) return rc;
        SerialLaneVarArgs v = {};
        v.m = h->m; v.n = h->n; v.nnz = h->nnz; v.max_iter = h->max_iter; v.it_start = it_start; v.n_levels = h->n_levels;
        v.ms_scaling_factor = h->ms_scaling_factor;
        v.rows = rows;
        v.row_ptr = h->d_row_ptr; v.col_idx = h->d_col_idx;
        v.lane_lvl = (const int32_t *)h->ser_var_lane_lvl.p; v.lane_items = (const int32_t *)h->ser_var_lane_items.p;
        v.llr0 = h->d_llr0;
        v.A = state_rows; v.synd = synd; v.decoding = decoding; v.llr = llr; v.iters = iters; v.conv = conv;
        void (*kv)(const SerialLaneVarArgs);
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kv = pick_serial_lane_var<LDPC_HIP_MINIMUM_SUM, 0>(sp);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) kv = pick_serial_lane_var<LDPC_HIP_PRODUCT_SUM, 1>(sp);
        else kv = pick_serial_lane_var<LDPC_HIP_PRODUCT_SUM, 0>(sp);
        if (llr && !h->order_visits_all) HIPCHK(hipMemsetAsync(llr, 0, sizeof(double) * (size_t)h->n * (size_t)rows, h->stream));  // bits the order never visits report 0
        const size_t dynv = ((size_t)h->n + 15) & ~(size_t)15;
        if (dynv > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)kv, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynv));
        hipLaunchKernelGGL(kv, dim3((unsigned)rows), dim3((unsigned)(h->sw("SER_LANE_THREADS") > 0 ? h->sw("SER_LANE_THREADS") : rows <= 2048 ? 1024 : 512)), (unsigned)dynv, h->stream, v);
        HIPCHK(hipGetLastError());
        return LDPC_HIP_OK;
    }
    SerialLaneArgs a = {};
    a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter; a.it_start = it_start; a.n_levels = h->n_levels;
    a.ms_scaling_factor = h->ms_scaling_factor;
    a.rows = rows;
    a.col_idx = h->d_col_idx; a.lvl_ptr = (const int32_t *)h->lvl_ptr.p; a.pos_tab = (const int32_t *)h->ser_pos_tab.p;
    a.llr0 = h->d_llr0;
    a.A = state_rows; a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    void (*kern)(const SerialLaneArgs);
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = bp_serial_lane_kernel<LDPC_HIP_MINIMUM_SUM, 0, 6, 3>;
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = bp_serial_lane_kernel<LDPC_HIP_PRODUCT_SUM, 1, 6, 3>;
    else kern = bp_serial_lane_kernel<LDPC_HIP_PRODUCT_SUM, 0, 6, 3>;
    (void)sp;
    if (llr && !h->order_visits_all) HIPCHK(hipMemsetAsync(llr, 0, sizeof(double) * (size_t)h->n * (size_t)rows, h->stream));  // bits the order never visits report 0
    const size_t dyn = ((size_t)h->n + 15) & ~(size_t)15;
    if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3((unsigned)(h->sw("SER_LANE_THREADS") > 0 ? h->sw("SER_LANE_THREADS") : rows <= 2048 ? 1024 : 512)), (unsigned)dyn, h->stream, a);
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// new_list[i] = list[sub[i]]
__global__ void __launch_bounds__(256) compose_lists_kernel(const int32_t *__restrict__ list, const int32_t *__restrict__ sub, int64_t count, int32_t *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = list[sub[i]];
}

// The whole decode.  A tile runs until the slowest of its 64 lanes is done, so the batch is decoded in PASSES: the first ends after 4
// iterations (ldpc_hip_bp_set_repack: where; 0 = one pass), the next after twice as many -- or, when a pass left less than 40 % of its
// rows (they are converging now), after ONE more iteration: a second pass of two iterations where half of the tiles need one keeps a
// compute unit on a two-tile queue for four tile-iterations while the others idle.  After a pass the rows still decoding are counted (the
// one place the host waits) and
//   * a handful of them (<= SER_LANE_MAX, default 2048) finish on bp_serial_lane_kernel, a workgroup per syndrome -- the hopeless
//     syndrome that would keep a tile on one compute unit for max_iter iterations costs a few milliseconds instead;
//   * most of the rows (> 60 %): the same tiles carry on where they stopped;
//   * else their message state is compacted, lane by lane, into dense tiles (the other message array) and the next pass runs on those.
// Every row keeps decoding from the state it had -- nothing restarts -- so the results are those of one uninterrupted decode.
// Returns 1 if the batch was decoded here, 0 if the caller should take the chunked path (the state of the whole batch must be resident).
static int decode_serial_streamed(ldpc_hip_bp *h, const SerialStreamPlan &sp, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                                  int32_t *iters, uint8_t *conv) {
    const int full = h->max_iter;
    const size_t B = (size_t)batch, m1 = (size_t)h->m, n1 = (size_t)h->n;
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    const size_t per_tile_msg = sizeof(double) * (size_t)h->nnz * LDPC_WAVE;
    int lane_max = h->sw("SER_LANE_MAX") >= 0 ? h->sw("SER_LANE_MAX") : 2048;
    if (h->n > 60000) lane_max = 0;  // (the lane kernel keeps a byte per bit in LDS)
    const bool lane_only = lane_max > 0 && batch <= (lane_max < 256 ? lane_max : 256) && h->repack_iters != 0;  // a small batch: a workgroup per syndrome from the start
    int rc;
    if (full <= 0) return 0;
    if (!lane_only) {
        size_t free_b = 0, total_b = 0;
        HIPCHK(hipMemGetInfo(&free_b, &total_b));
        const size_t have = h->msgA.cap + h->msgC.cap + h->llr_t.cap;
        const size_t per_tile = per_tile_msg + (llr ? sizeof(double) * n1 * LDPC_WAVE : 0) + 24 * (m1 + n1 + 1);
        if ((double)per_tile * (double)tiles_total > (double)(free_b + have) * 0.8 || (h->max_chunk_tiles > 0 && tiles_total > h->max_chunk_tiles)) return 0;
    }
    if (!sp.var && (rc = ensure_serial_stream_table(h, sp))) return rc;
    if (!conv) { if ((rc = h->osd_conv.ensure(B))) return rc; conv = (uint8_t *)h->osd_conv.p; }
    if (!iters) { if ((rc = h->sp_iters.ensure(B * 4))) return rc; iters = (int32_t *)h->sp_iters.p; }
    if (!h->h_counters) HIPCHK(hipHostMalloc((void **)&h->h_counters, 16, hipHostMallocDefault));
    hipStream_t st = h->stream;
    h->accumulated_ms = 0.f;
    h->accumulated_persistent_ms = 0.f;
    h->timed = h->timed_mid = false;
    h->timed_prev = h->timed_prev_mid = false;
    HIPCHK(hipEventRecord(h->ev0, st));
    auto finish = [&]() -> int {
        HIPCHK(hipEventRecord(h->ev1, st));
        h->timed = true;
        return 1;
    };
    if (lane_only) {
        if ((rc = h->msgC.ensure(sizeof(double) * (size_t)h->nnz * B))) return rc;
        if ((rc = serial_lane_launch(h, sp, 0, (double *)h->msgC.p, synd, batch, decoding, llr, iters, conv))) return rc;
        return finish();
    }
    if (!sp.var && h->order_visits_all && !h->on("EXPLICIT_INIT")) {  // the tables that stand in for the initial messages (bp_serial_stream_kernel.h)
        if ((rc = h->d_edge0.ensure(sizeof(double) * n1)) || (rc = h->ser_pos_e0.ensure(sizeof(double) * 16 * n1))) return rc;
        const dim3 ge((unsigned)((h->n + 255) / 256));
        if (h->bp_method == LDPC_HIP_MINIMUM_SUM) hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_MINIMUM_SUM, 0>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
        else if (h->math_mode == LDPC_HIP_MATH_FAST) hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 1>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
        else hipLaunchKernelGGL((serial_edge0_kernel<LDPC_HIP_PRODUCT_SUM, 0>), ge, dim3(256), 0, st, h->d_llr0, h->n, (double *)h->d_edge0.p);
        hipLaunchKernelGGL(serial_pos_e0_kernel, dim3((unsigned)((h->n * 16 + 255) / 256)), dim3(256), 0, st, (const int32_t *)h->ser_pos_tab.p, h->d_col_idx,
                           (const double *)h->d_edge0.p, h->n, sp.dc * (sp.dr - 1), (double *)h->ser_pos_e0.p);
        HIPCHK(hipGetLastError());
    }
    if ((rc = h->msgA.ensure(per_tile_msg * (size_t)tiles_total))) return rc;
    DeviceBuf *state = &h->msgA, *other = &h->msgC;
    // the rows of the running pass: the caller's (identity) or a compacted subset -- their syndromes, their numbers in the caller's arrays
    bool identity = true;
    int64_t R = batch;
    const uint8_t *cur_synd = synd;
    int cur = 0;  // which of the two row-list / syndrome buffers describes the running rows
    DeviceBuf *lists[2] = {&h->ser_rows[0], &h->ser_rows[1]}, *synds[2] = {&h->rp_synd, &h->ser_synd2};
    auto grid = [](size_t items) { return flat_grid(items); };
    auto scatter_out = [&](const int32_t *rows_list, int64_t cnt, bool big) -> int {  // rp_* -> the caller's arrays
        const size_t C = (size_t)cnt;
        if (big) {
            hipLaunchKernelGGL(scatter_rows_kernel<uint8_t>, grid(C * n1), dim3(256), 0, st, (const uint8_t *)h->rp_dec.p, rows_list, cnt, h->n, decoding);
            if (llr) hipLaunchKernelGGL(scatter_rows_kernel<double>, grid(C * n1), dim3(256), 0, st, (const double *)h->rp_llr.p, rows_list, cnt, h->n, llr);
        }
        hipLaunchKernelGGL(scatter_rows_kernel<int32_t>, grid(C), dim3(256), 0, st, (const int32_t *)h->rp_iters.p, rows_list, cnt, 1, iters);
        hipLaunchKernelGGL(scatter_rows_kernel<uint8_t>, grid(C), dim3(256), 0, st, (const uint8_t *)h->rp_conv.p, rows_list, cnt, 1, conv);
        HIPCHK(hipGetLastError());
        return LDPC_HIP_OK;
    };
    const int first = h->repack_iters > 0 ? h->repack_iters : 4;
    int it = 0;
    bool resume = false;
    bool thinning = false;  // the last pass left a minority of its rows: they are converging now, look again after one more iteration
    for (;;) {
        // (resume: most rows of the last pass are still decoding and carry on in their tiles -- a look after every iteration costs a launch
        // and a count, while a pass that runs on to twice the iterations keeps every tile going to its slowest lane)
        int next = h->repack_iters == 0 ? full : (it == 0 ? first : (thinning || resume) ? it + 1 : it * 2);
        if (next > full || next <= it) next = full;
        // (a compacted pass reaches the caller's decoding / llr rows through its row list; its iteration counts and flags go by the pass's own rows)
        int32_t *o_it = identity ? iters : (int32_t *)h->rp_iters.p;
        uint8_t *o_cv = identity ? conv : (uint8_t *)h->rp_conv.p;
        const int waves_cap = !identity && h->sw("SER_WAVES2") > 0 ? h->sw("SER_WAVES2") : 16;  // (workgroups of 8 for a second pass of 257 .. 512 tiles measured slower than a second round of 16)
        if ((rc = serial_stream_launch(h, sp, it, next, resume, (double *)state->p, cur_synd, R, identity ? nullptr : (const int32_t *)lists[cur]->p, decoding, llr, o_it, o_cv, waves_cap))) return rc;
        if (!identity && (rc = scatter_out((const int32_t *)lists[cur]->p, R, false))) return rc;
        if (next >= full) break;
        // the rows of this pass that are still decoding: listed (numbers within the pass) and counted
        if ((rc = h->osd_list.ensure((size_t)R * sizeof(int32_t))) || (rc = h->osd_counters.ensure(2 * sizeof(unsigned)))) return rc;
        HIPCHK(hipMemsetAsync(h->osd_counters.p, 0, 2 * sizeof(unsigned), st));
        hipLaunchKernelGGL(osd_collect_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, o_cv, R, (int32_t *)h->osd_list.p, (unsigned *)h->osd_counters.p);
        HIPCHK(hipMemcpyAsync(&h->h_counters[2], h->osd_counters.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));  // the size of what is left is needed on the host
        const int64_t cnt = (int64_t)h->h_counters[2];
        if (cnt == 0) break;
        it = next;
        const int32_t *sub = (const int32_t *)h->osd_list.p;
        const size_t C = (size_t)cnt;
        const bool to_lanes = cnt <= lane_max;
        thinning = cnt * 10 <= R * 4;
        if (!to_lanes && cnt * 10 > R * 6) { resume = true; continue; }  // most rows are still decoding: the same tiles carry on
        // the rows that go on: their numbers in the caller's arrays, their syndromes
        const int nxt = cur ^ 1;
        if ((rc = lists[nxt]->ensure(C * sizeof(int32_t))) || (rc = synds[nxt]->ensure(C * m1))) return rc;
        if (identity) HIPCHK(hipMemcpyAsync(lists[nxt]->p, sub, C * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        else hipLaunchKernelGGL(compose_lists_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, st, (const int32_t *)lists[cur]->p, sub, cnt, (int32_t *)lists[nxt]->p);
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, grid(C * m1), dim3(256), 0, st, cur_synd, sub, cnt, h->m, (uint8_t *)synds[nxt]->p);
        HIPCHK(hipGetLastError());
        if ((rc = h->rp_iters.ensure(C * 4)) || (rc = h->rp_conv.ensure(C))) return rc;
        // A compute unit holds one 16-wavefront tile: a pass of 267 tiles is a round of 256 and a round of 11 that takes as long.  The rows
        // beyond the last whole round of tiles -- if they are few enough for the lane kernel -- finish there instead, at once.
        int64_t keep = cnt;  // rows that go on in tiles
        if (!to_lanes && lane_max > 0) {
            const int64_t round_rows = (h->sw("SER_ROUND_TILES") > 0 ? h->sw("SER_ROUND_TILES") : 256) * (int64_t)LDPC_WAVE, over = cnt % round_rows;
            if (cnt > round_rows && over > 0 && over <= lane_max && h->sw("SER_NO_REMAINDER") <= 0) keep = cnt - over;
        }
        const int64_t tiles2 = to_lanes ? 0 : (keep + LDPC_WAVE - 1) / LDPC_WAVE, n_lane = to_lanes ? cnt : cnt - keep;
        // (one allocation for both: the compacted tiles first, the lane kernel's row-major arrays behind them)
        if ((rc = other->ensure(per_tile_msg * (size_t)tiles2 + sizeof(double) * (size_t)h->nnz * (size_t)n_lane))) return rc;
        if (n_lane > 0) {
            const int64_t at = cnt - n_lane;  // the last n_lane rows of the list
            double *rows_state = (double *)((char *)other->p + per_tile_msg * (size_t)tiles2);
            if ((rc = h->rp_dec.ensure((size_t)n_lane * n1)) || (llr && (rc = h->rp_llr.ensure((size_t)n_lane * n1 * 8)))) return rc;
            hipLaunchKernelGGL(serial_rows_from_tiles_kernel, dim3((unsigned)((h->nnz + 1023) / 1024), (unsigned)n_lane), dim3(256), 0, st, (const double *)state->p, sub + at, n_lane,
                               h->nnz, rows_state);
            HIPCHK(hipGetLastError());
            if ((rc = serial_lane_launch(h, sp, it, rows_state, (const uint8_t *)synds[nxt]->p + (size_t)at * m1, n_lane, (uint8_t *)h->rp_dec.p, llr ? (double *)h->rp_llr.p : nullptr,
                                         (int32_t *)h->rp_iters.p, (uint8_t *)h->rp_conv.p))) return rc;
            if ((rc = scatter_out((const int32_t *)lists[nxt]->p + at, n_lane, true))) return rc;
            if (to_lanes) break;
        }
        // the others' message state, lane by lane, into dense tiles of the other array
        const int epw = 16;
        const dim3 gg((unsigned)((h->nnz + 4 * epw - 1) / (4 * epw)), (unsigned)(tiles2 < 32768 ? tiles2 : 32768));
        hipLaunchKernelGGL(gather_lane_state_kernel, gg, dim3(256), 0, st, (const double *)state->p, sub, keep, h->nnz, epw, (double *)other->p, (const unsigned *)nullptr);
        HIPCHK(hipGetLastError());
        std::swap(state, other);
        cur = nxt;
        cur_synd = (const uint8_t *)synds[cur]->p;
        identity = false;
        resume = false;
        R = keep;
    }
    return finish();
}

int decode_serial(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                  int32_t *iters, uint8_t *conv) {
    if (h->random_serial) return decode_serial_random(h, synd, batch, decoding, llr, iters, conv);  // (takes precedence, bp.hpp:467-469)
    if (h->schedule == 2) return decode_serial_relative(h, synd, batch, decoding, llr, iters, conv);
    const size_t B = (size_t)batch, m1 = (size_t)(h->m ? h->m : 1), n1 = (size_t)(h->n ? h->n : 1);
    int rc;
    // the streamed kernels (bp_serial_stream_kernel.h) where the matrix and the schedule allow them: automatic, or asked for (mode 2)
    if ((h->serial_kernel == -1 || h->serial_kernel == 2) && h->n > 0 && h->m > 0 && batch > 0) {
        if ((rc = ensure_serial_levels(h))) return rc;
        const SerialStreamPlan sp = plan_serial_stream(h);
        if (sp.dr) {
            int took = decode_serial_streamed(h, sp, synd, batch, decoding, llr, iters, conv);
            if (took < 0) return took;
            if (took > 0) return LDPC_HIP_OK;
            // The batch's message state is not resident at once (e.g. 1 048 576 rows of the n = 10 000 code: 252 GB): in pieces that are, each
            // decoded in passes by itself -- a piece's stragglers finish a workgroup per syndrome instead of holding their tiles to max_iter.
            size_t free_b = 0, total_b = 0;
            HIPCHK(hipMemGetInfo(&free_b, &total_b));
            const size_t have = h->msgA.cap + h->msgC.cap + h->llr_t.cap;
            const size_t per_tile = sizeof(double) * (size_t)h->nnz * LDPC_WAVE + (llr ? sizeof(double) * n1 * LDPC_WAVE : 0) + 24 * (m1 + n1 + 1);
            int64_t fit = (int64_t)((double)(free_b + have) * 0.7 / (double)per_tile);
            if (h->max_chunk_tiles > 0 && fit > h->max_chunk_tiles) fit = h->max_chunk_tiles;
            if (fit >= 64) {
                float ms_sum = 0.f;
                for (int64_t b0 = 0; b0 < batch; b0 += fit * LDPC_WAVE) {
                    const int64_t nb = batch - b0 < fit * LDPC_WAVE ? batch - b0 : fit * LDPC_WAVE;
                    took = decode_serial_streamed(h, sp, synd + b0 * h->m, nb, decoding + b0 * h->n, llr ? llr + (size_t)b0 * h->n : nullptr, iters ? iters + b0 : nullptr,
                                                  conv ? conv + b0 : nullptr);
                    if (took < 0) return took;
                    if (took == 0) return fail(LDPC_HIP_ERR_NOMEM, "serial schedule: a piece of %lld rows sized from free device memory did not fit after all", (long long)nb);
                    if (b0 + nb < batch) { float ms = 0.f; (void)ldpc_hip_bp_last_kernel_ms(h, &ms); ms_sum += ms; }
                }
                h->accumulated_ms += ms_sum;  // (the last piece's events are still the handle's)
                return LDPC_HIP_OK;
            }
        }
    }
    int k1 = h->repack_iters < 0 ? h->max_iter / 8 : h->repack_iters;
    if (h->repack_iters < 0 && k1 < 2) k1 = 2;
    if (k1 <= 0 || k1 >= h->max_iter || batch <= 4 * LDPC_WAVE)
        return decode_serial_pass(h, h->max_iter, synd, batch, decoding, llr, iters, conv);
    if (!conv) { if ((rc = h->osd_conv.ensure(B))) return rc; conv = (uint8_t *)h->osd_conv.p; }
    if (!h->h_counters) HIPCHK(hipHostMalloc((void **)&h->h_counters, 16, hipHostMallocDefault));
    if ((rc = decode_serial_pass(h, k1, synd, batch, decoding, llr, iters, conv))) return rc;
    if ((rc = h->osd_list.ensure(B * sizeof(int32_t)))) return rc;
    if ((rc = h->osd_counters.ensure(2 * sizeof(unsigned)))) return rc;
    HIPCHK(hipMemsetAsync(h->osd_counters.p, 0, 2 * sizeof(unsigned), h->stream));
    hipLaunchKernelGGL(osd_collect_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, h->stream, conv, batch,
                       (int32_t *)h->osd_list.p, (unsigned *)h->osd_counters.p);
    HIPCHK(hipMemcpyAsync(&h->h_counters[2], h->osd_counters.p, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // the size of the second pass is needed on the host
    const int64_t cnt = (int64_t)h->h_counters[2];
    if (cnt == 0) return LDPC_HIP_OK;
    float ms1 = 0.f;
    (void)ldpc_hip_bp_last_kernel_ms(h, &ms1);
    const size_t C = (size_t)cnt;
    if ((rc = h->rp_synd.ensure(C * m1)) || (rc = h->rp_dec.ensure(C * n1)) || (rc = h->rp_iters.ensure(C * 4)) ||
        (rc = h->rp_conv.ensure(C)) || (llr && (rc = h->rp_llr.ensure(C * n1 * 8)))) return rc;
    const int32_t *list = (const int32_t *)h->osd_list.p;
    auto grid = [](size_t items) { return flat_grid(items); };
    if (h->m > 0)
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, grid(C * h->m), dim3(256), 0, h->stream, synd, list, cnt, h->m, (uint8_t *)h->rp_synd.p);
    HIPCHK(hipGetLastError());
    if ((rc = decode_serial_pass(h, h->max_iter, (const uint8_t *)h->rp_synd.p, cnt, (uint8_t *)h->rp_dec.p,
                                 llr ? (double *)h->rp_llr.p : nullptr, (int32_t *)h->rp_iters.p, (uint8_t *)h->rp_conv.p))) return rc;
    h->accumulated_ms += ms1;  // both passes count as this decode's kernel time
    if (h->n > 0) {
        hipLaunchKernelGGL(scatter_rows_kernel<uint8_t>, grid(C * h->n), dim3(256), 0, h->stream, (const uint8_t *)h->rp_dec.p, list, cnt, h->n, decoding);
        if (llr) hipLaunchKernelGGL(scatter_rows_kernel<double>, grid(C * h->n), dim3(256), 0, h->stream, (const double *)h->rp_llr.p, list, cnt, h->n, llr);
    }
    if (iters) hipLaunchKernelGGL(scatter_rows_kernel<int32_t>, grid(C), dim3(256), 0, h->stream, (const int32_t *)h->rp_iters.p, list, cnt, 1, iters);
    hipLaunchKernelGGL(scatter_rows_kernel<uint8_t>, grid(C), dim3(256), 0, h->stream, (const uint8_t *)h->rp_conv.p, list, cnt, 1, conv);
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// soft_info_decode_serial over a batch (bp_softinfo_kernel).  Device pointers, on h->stream.
int soft_info_device(ldpc_hip_bp *h, const double *soft, int64_t batch, double cutoff, double sigma, uint8_t *decoding,
                            double *llr, int32_t *iters, uint8_t *conv, double *soft_out) {
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    if (tiles_total == 0) return LDPC_HIP_OK;
    const size_t m1 = (size_t)(h->m ? h->m : 1), n1 = (size_t)(h->n ? h->n : 1);
    const size_t per_tile_msg = sizeof(double) * (size_t)(h->nnz ? h->nnz : 1) * LDPC_WAVE;
    const size_t per_tile_llr = llr ? sizeof(double) * n1 * LDPC_WAVE : 0;
    const size_t per_tile_soft = sizeof(double) * m1 * LDPC_WAVE;
    const size_t lds = sizeof(uint64_t) * (m1 + 32);  // hard-syndrome words + the level kernel's reduction slots
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
    int level_waves = 0;  // level-parallel variant: as for the serial schedule
    if (h->serial_kernel != 0 && h->n > 0) {
        if ((rc = ensure_serial_levels(h))) return rc;
        const double per_level = (double)h->n / (double)(h->n_levels ? h->n_levels : 1);
        if (h->serial_kernel == 1 || per_level >= 2.0) {
            level_waves = (int)(per_level + 0.999);
            if (level_waves > 8) level_waves = 8;
            if (level_waves < 1) level_waves = 1;
        }
    }
    // random_serial_schedule in this routine (bp.hpp:573-577): at the top of every iteration that still runs the order the
    // object carries is rearranged by std::shuffle with a NEW std::default_random_engine(random_schedule_seed) -- one fixed
    // rearrangement applied again and again.  Every row of the batch starts from the handle's order; the call leaves the order
    // of its last row (its iteration count many rearrangements on).
    const bool shuffled = h->random_serial && h->n > 0;
    int32_t *d_iters_last = nullptr;
    if (shuffled) {
        if ((rc = random_orders_prepare(h, 1))) return rc;  // (random_orders_*, above)
        level_waves = 0;  // the fixed order's levels do not apply: those of the ring's rows do (random_orders_append)
        if (h->serial_kernel != 0 && h->rnd.valid) {
            double sum = 0.0;
            int cnt = 0;
            for (int32_t nl : h->rnd.n_levels) if (nl > 0) { sum += (double)nl; ++cnt; }
            const double per_level = cnt ? (double)h->n / (sum / cnt) : 0.0;
            if (h->serial_kernel == 1 || per_level >= 2.0) {
                level_waves = (int)(per_level + 0.999);
                if (level_waves > 8) level_waves = 8;
                if (level_waves < 1) level_waves = 1;
            }
        }
        if (!iters) { if ((rc = h->sp_iters.ensure((size_t)batch * 4))) return rc; iters = (int32_t *)h->sp_iters.p; }
        d_iters_last = iters + (batch - 1);
    }
    void (*soft_kern)(const SoftArgs);
    if (h->max_row_deg <= 4 && h->max_col_deg <= 2) soft_kern = level_waves ? bp_softinfo_level_kernel<2, 4> : bp_softinfo_kernel<2, 4>;
    else if (h->max_row_deg <= 6 && h->max_col_deg <= 3) soft_kern = level_waves ? bp_softinfo_level_kernel<3, 6> : bp_softinfo_kernel<3, 6>;
    else if (h->max_row_deg <= 8 && h->max_col_deg <= 4) soft_kern = level_waves ? bp_softinfo_level_kernel<4, 8> : bp_softinfo_kernel<4, 8>;
    else soft_kern = level_waves ? bp_softinfo_level_kernel<0, 0> : bp_softinfo_kernel<0, 0>;
    if (lds > 48u * 1024u)
        HIPCHK(hipFuncSetAttribute((const void *)soft_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    h->timed_prev = h->timed_prev_mid = false;
    h->accumulated_ms = 0.f;
    h->accumulated_persistent_ms = 0.f;
    h->timed = false;
    h->timed_mid = false;
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
        if (shuffled && h->max_iter > 0) {
            a.orders = (const int32_t *)h->sched_orders.p; a.n_orders = h->max_iter; a.orders_first = h->rnd.first;
            if (level_waves) { a.orders_lvl = (const int32_t *)h->sched_lvl_bits.p; a.orders_lvl_ptr = (const int32_t *)h->sched_lvl_ptr.p; }
        }
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
        a.lvl_ptr = (const int32_t *)h->lvl_ptr.p; a.lvl_bits = (const int32_t *)h->lvl_bits.p; a.n_levels = h->n_levels;
        hipLaunchKernelGGL(soft_kern, dim3((unsigned)tiles), dim3((unsigned)(64 * (level_waves ? level_waves : 1))), (unsigned)lds, st, a);
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
    if (shuffled && batch > 0) {  // the order the last row leaves behind
        int32_t last = 0;
        HIPCHK(hipMemcpyAsync(&last, d_iters_last, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        if ((rc = random_orders_consume(h, 1, last))) return rc;
    }
    return LDPC_HIP_OK;
}
