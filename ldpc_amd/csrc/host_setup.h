// host_setup.h -- C ABI: create / destroy / setters / timings (include/ldpc_hip.h)
// Part of libldpc_hip.so: included by bp_hip.hip (one translation unit), in the order given there.
#pragma once

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
    for (int i = 0; i < k_n_switches; ++i) {
        const std::string var = std::string("LDPC_HIP_") + k_switch_names[i];
        const char *e = getenv(var.c_str());
        h->switches[i] = e ? (*e ? atoi(e) : 1) : -1;
    }
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
    h->min_col_deg = min_col;
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
    h->sched_state.resize((size_t)d->n);
    for (int j = 0; j < d->n; ++j) h->sched_state[(size_t)j] = j;  // bp.hpp:120-124
    h->h_row_ptr.assign(d->csr_row_ptr, d->csr_row_ptr + d->m + 1);  // kept for tables that are built on first use
    h->h_col_idx.assign(d->csr_col_idx, d->csr_col_idx + d->nnz);
    hipError_t e = hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&h->ev0);
    if (e == hipSuccess) e = hipEventCreate(&h->ev1);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_mid);
    if (e == hipSuccess) e = hipEventCreate(&h->evp0);
    if (e == hipSuccess) e = hipEventCreate(&h->evp1);
    if (e == hipSuccess) e = hipEventCreate(&h->evp_mid);
    if (e == hipSuccess) e = hipEventCreate(&h->ev_hist);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) { std::memset(h->h_flag, 0, 64); e = hipHostGetDevicePointer((void **)&h->d_flag, h->h_flag, 0); }
    if (e == hipSuccess) e = hipMalloc((void **)&h->d_clk, 16);
    if (e == hipSuccess) e = hipMemset(h->d_clk, 0, 16);
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
    resident_retire(h);  // (the resident decode workgroup, if one lingers: it reads the host-mapped block freed below)
    if (h->res.ended) (void)hipEventDestroy(h->res.ended);
    if (h->res.stream) (void)hipStreamDestroy(h->res.stream);
    (void)hipStreamSynchronize(h->stream);
    for (DeviceBuf *b : {&h->flood_lane_scratch, &h->flood_list2, &h->flood_pos, &h->ser_pos_tab, &h->ser_pos_e0, &h->ser_rows[0], &h->ser_rows[1], &h->ser_synd2}) b->release();
    for (DeviceBuf *b : {&h->msgA, &h->msgC, &h->par, &h->nzm, &h->invalid, &h->dec, &h->dcur, &h->llr_t,
                         &h->st_synd, &h->st_dec, &h->st_llr, &h->st_iters, &h->st_conv, &h->st_misc, &h->osd_llr, &h->osd_conv, &h->osd_packed, &h->osd_ell, &h->osd_scratch, &h->sp_hist, &h->sp_iters, &h->osd_list, &h->osd_counters, &h->osd_status, &h->osd_fix_synd, &h->osd_fix_list, &h->osd_fix_counters, &h->osd_fix_scratch, &h->rel_ord, &h->rel_dbit, &h->rl_edge, &h->rl_chk, &h->rl_cdeg, &h->rl_last, &h->sched_orders, &h->sched_order0, &h->sched_lvl_bits, &h->sched_lvl_ptr, &h->lvl_ptr, &h->lvl_bits, &h->rp_synd, &h->rp_dec, &h->rp_llr, &h->rp_iters, &h->rp_conv, &h->counter, &h->w_rdeg, &h->w_cdeg, &h->w_col, &h->w_apos, &h->w_prior, &h->d_edge0, &h->var_row_items, &h->var_pair_items, &h->e_partner, &h->e_kind, &h->e_scol, &h->e_prior, &h->wp_rdeg, &h->wp_col, &h->wp_epos,
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
    if (h->ev_mid) (void)hipEventDestroy(h->ev_mid);
    if (h->evp0) (void)hipEventDestroy(h->evp0);
    if (h->evp1) (void)hipEventDestroy(h->evp1);
    if (h->evp_mid) (void)hipEventDestroy(h->evp_mid);
    if (h->ev_hist) (void)hipEventDestroy(h->ev_hist);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    if (h->d_clk) (void)hipFree(h->d_clk);
    if (h->pin_host) (void)hipHostFree(h->pin_host);
    for (int q = 0; q < ldpc_hip_bp::HostPipe::NB; ++q) {
        if (h->pipe.pin_in[q]) (void)hipHostFree(h->pipe.pin_in[q]);
        if (h->pipe.pin_out[q]) (void)hipHostFree(h->pipe.pin_out[q]);
        if (h->pipe.ev_in[q]) (void)hipEventDestroy(h->pipe.ev_in[q]);
        if (h->pipe.ev_cmp[q]) (void)hipEventDestroy(h->pipe.ev_cmp[q]);
        if (h->pipe.ev_out[q]) (void)hipEventDestroy(h->pipe.ev_out[q]);
        for (DeviceBuf *b : {&h->pipe.d_in[q], &h->pipe.d_dec[q], &h->pipe.d_llr[q], &h->pipe.d_it[q], &h->pipe.d_cv[q]}) b->release();
    }
    if (h->pipe.s_in) (void)hipStreamDestroy(h->pipe.s_in);
    if (h->pipe.s_out) (void)hipStreamDestroy(h->pipe.s_out);
    if (h->h_hist) (void)hipHostFree(h->h_hist);
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
    hipStream_t ns;
    if (s == LDPC_HIP_STREAM_LEGACY_DEFAULT) ns = nullptr;  // hipStream_t 0: the device's legacy default stream
    else ns = s ? (hipStream_t)s : h->own_stream;
    if (ns != h->stream && h->work_queued) {
        // The handle has ONE workspace: work queued on the old stream (an *_async decode) may still be using it, so
        // everything queued on the new stream from now on is ordered after it.
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamWaitEvent(ns, h->ev_done, 0));
    }
    h->stream = ns;
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
    if (schedule != 0 && schedule != 1 && schedule != 2) return fail(LDPC_HIP_ERR_INVALID, "Invalid BP schedule");  // bp.hpp:188
    HIPCHK(hipSetDevice(h->device));
    for (int j = 0; j < h->n; ++j) {  // the object's serial_schedule_order: the given order, else 0 .. n-1 (bp.hpp:110-124)
        if (serial_schedule_order && (serial_schedule_order[j] < 0 || serial_schedule_order[j] >= h->n))
            return fail(LDPC_HIP_ERR_INVALID, "serial_schedule_order[%d] is out of range", j);
        h->sched_state[(size_t)j] = serial_schedule_order ? serial_schedule_order[j] : j;
    }
    if (schedule == 0 && serial_schedule_order) {
        for (int j = 0; j < h->n; ++j)
            if (serial_schedule_order[j] < 0 || serial_schedule_order[j] >= h->n)
                return fail(LDPC_HIP_ERR_INVALID, "serial_schedule_order[%d] is out of range", j);
        if (!h->d_order) HIPCHK(hipMalloc((void **)&h->d_order, sizeof(int32_t) * (size_t)(h->n ? h->n : 1)));
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(h->d_order, serial_schedule_order, sizeof(int32_t) * (size_t)h->n, hipMemcpyHostToDevice));
        h->custom_order = true;
        std::vector<char> seen((size_t)(h->n ? h->n : 1), 0);
        for (int j = 0; j < h->n; ++j) seen[(size_t)serial_schedule_order[j]] = 1;
        h->order_visits_all = true;
        for (int j = 0; j < h->n; ++j) h->order_visits_all = h->order_visits_all && seen[(size_t)j];
    } else {
        h->custom_order = false;
        h->order_visits_all = true;
    }
    h->schedule = schedule;
    h->levels_valid = false;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_random_serial(ldpc_hip_bp *h, int32_t enable, uint32_t seed) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    h->random_serial = enable != 0;
    h->sched_seed_raw = (int32_t)seed;  // soft_info_decode_serial seeds a std::default_random_engine with the member as it is (bp.hpp:576)
    if (seed == 0)  // rng.hpp:117-123: seed 0 = take the system clock
        seed = (unsigned)std::chrono::system_clock::now().time_since_epoch().count();
    h->sched_rng.seed(seed);  // BpDecoder::set_random_schedule_seed (bp.hpp:142-145)
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_get_schedule_order(ldpc_hip_bp *h, int32_t *order) {
    if (!h || !order) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    for (int j = 0; j < h->n; ++j) order[j] = h->sched_state[(size_t)j];
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_serial_kernel(ldpc_hip_bp *h, int32_t mode) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (mode < -1 || mode > 2) return fail(LDPC_HIP_ERR_INVALID, "mode must be -1 (automatic), 0 (one wavefront per tile), 1 (level-parallel) or 2 (level-parallel, streamed through LDS rings where the matrix allows it)");
    h->serial_kernel = mode;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_handoff(ldpc_hip_bp *h, int32_t threshold_tiles) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (threshold_tiles < -1) return fail(LDPC_HIP_ERR_INVALID, "threshold must be -1 (auto), 0 (off) or a tile count");
    h->handoff = threshold_tiles > 32768 ? 32768 : threshold_tiles;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_debug_switch(ldpc_hip_bp *h, const char *name, int32_t value) {
    if (!h || !name) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    for (int i = 0; i < k_n_switches; ++i)
        if (!std::strcmp(name, k_switch_names[i])) { h->switches[i] = value < 0 ? -1 : value; return LDPC_HIP_OK; }
    return fail(LDPC_HIP_ERR_INVALID, "unknown switch '%s'", name);
}

int ldpc_hip_bp_set_small_code_kernel(ldpc_hip_bp *h, int32_t mode) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (mode < -1 || mode > 6)
        return fail(LDPC_HIP_ERR_INVALID, "mode must be -1 (auto), 0 (off), 1 (whenever one fits), 2 (slot kernel only), 3 (lane = node wavefront kernel), "
                                          "4 (that kernel, one wavefront per syndrome), 5 (that kernel, a workgroup per syndrome) or 6 (lane = edge kernel where it applies)");
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
    if (h->timed_prev) {  // the first pass of a two-pass decode (recorded earlier on the same stream: complete by now)
        HIPCHK(hipEventElapsedTime(&last, h->evp0, h->evp1));
        *ms += last;
    }
    return LDPC_HIP_OK;
}

void *ldpc_hip_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}

void ldpc_hip_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

int ldpc_hip_bp_clock_probe(ldpc_hip_bp *h, uint64_t *cycles, uint64_t *ticks, double *tick_hz) {
    if (!h || !cycles || !ticks || !tick_hz) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    unsigned long long v[2] = {0, 0};
    HIPCHK(hipMemcpy(v, h->d_clk, 16, hipMemcpyDeviceToHost));
    int khz = 0;
    HIPCHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->device));
    *cycles = v[0];
    *ticks = v[1];
    *tick_hz = (double)khz * 1e3;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_copy_probe(ldpc_hip_bp *h, int64_t tiles, int32_t segments_per_tile, int32_t passes, float *ms, double *gbytes_per_s) {
    if (!h || !ms || !gbytes_per_s) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    if (tiles <= 0 || segments_per_tile <= 0 || passes <= 0 || passes > 64) return fail(LDPC_HIP_ERR_INVALID, "copy probe: tiles, segments_per_tile > 0, passes in 1 .. 64");
    HIPCHK(hipSetDevice(h->device));
    const size_t bytes = (size_t)tiles * (size_t)segments_per_tile * 512u;
    int rc;
    // (the message arrays of the handle: a decode of the same geometry has them already, and the next decode initialises what it reads)
    if ((rc = h->msgA.ensure(bytes)) || (rc = h->msgC.ensure(bytes))) return rc;
    hipStream_t st = h->stream;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0));
    HIPCHK(hipEventCreate(&e1));
    double *a = (double *)h->msgA.p, *c = (double *)h->msgC.p;
    hipLaunchKernelGGL(segcopy_probe_kernel, dim3((unsigned)tiles), dim3(768), 0, st, a, c, (int)segments_per_tile);  // untimed: page tables, clocks
    HIPCHK(hipEventRecord(e0, st));
    for (int p = 0; p < passes; ++p) {
        hipLaunchKernelGGL(segcopy_probe_kernel, dim3((unsigned)tiles), dim3(768), 0, st, (p & 1) ? c : a, (p & 1) ? a : c, (int)segments_per_tile);
    }
    HIPCHK(hipEventRecord(e1, st));
    HIPCHK(hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(hipEventElapsedTime(&t, e0, e1));
    HIPCHK(hipEventDestroy(e0));
    HIPCHK(hipEventDestroy(e1));
    *ms = t;
    *gbytes_per_s = t > 0.f ? 2.0 * (double)bytes * (double)passes / ((double)t * 1e-3) / 1e9 : 0.0;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_last_phase_ms(ldpc_hip_bp *h, float *persistent_ms, float *per_pass_ms) {
    if (!h || !persistent_ms || !per_pass_ms) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    *persistent_ms = *per_pass_ms = 0.f;
    float total = 0.f;
    int rc = ldpc_hip_bp_last_kernel_ms(h, &total);
    if (rc) return rc;
    float pers = h->accumulated_persistent_ms;
    if (h->timed && h->timed_mid) {
        float last = 0.f;
        HIPCHK(hipEventElapsedTime(&last, h->ev0, h->ev_mid));
        pers += last;
    }
    if (h->timed_prev && h->timed_prev_mid) {
        float last = 0.f;
        HIPCHK(hipEventElapsedTime(&last, h->evp0, h->evp_mid));
        pers += last;
    }
    *persistent_ms = pers;
    *per_pass_ms = total - pers;
    return LDPC_HIP_OK;
}

}  // extern "C"
