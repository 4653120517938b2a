// host_decode_abi.h -- C ABI: decode entry points (host / device buffers), H v, shot generation, b8 I/O, soft syndromes
// Part of libldpc_hip.so: included by bp_hip.hip (one translation unit), in the order given there.
#pragma once

extern "C" {

int ldpc_hip_bposd0_decode_batch_async(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch,
                                       uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    HIPCHK(hipSetDevice(h->device));
    return mark_queued(h, bposd_device(h, 1, 0, synd, batch, decoding, llr, iters, conv));
}

int ldpc_hip_bp_set_osd(ldpc_hip_bp *h, int32_t osd_method, int32_t osd_order) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (osd_method < 0 || osd_method > 3) return fail(LDPC_HIP_ERR_INVALID, "osd_method must be 0 (off), 1 (OSD_0), 2 (OSD_E) or 3 (OSD_CS)");
    if (osd_order < 0) return fail(LDPC_HIP_ERR_INVALID, "osd_order must not be negative");  // _bposd_decoder.pyx:222-223
    if (osd_method == 1 && osd_order != 0) return fail(LDPC_HIP_ERR_INVALID, "osd_method OSD_0 requires osd_order 0");  // pyx:225-226
    if (osd_method == 2 && osd_order > 24)
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "OSD_E with osd_order > 24 (more than 16 million candidates per syndrome) is not available");
    h->osd_method = osd_method;
    h->osd_order = osd_order;
    return LDPC_HIP_OK;
}

int ldpc_hip_bposd_get_status(ldpc_hip_bp *h, uint8_t *status, int64_t batch) {
    if (!h || !status) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    if (batch != h->osd_status_rows) return fail(LDPC_HIP_ERR_INVALID, "the last BP + OSD decode on this handle had %lld rows, not %lld", (long long)h->osd_status_rows, (long long)batch);
    if (batch == 0) return LDPC_HIP_OK;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemcpyAsync(status, h->osd_status.p, (size_t)batch, is_device_ptr(status) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_repack(ldpc_hip_bp *h, int32_t first_pass_iters) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (first_pass_iters < -1) return fail(LDPC_HIP_ERR_INVALID, "first_pass_iters must be -1 (automatic), 0 (off) or an iteration count");
    h->repack_iters = first_pass_iters;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_set_osd_kernel(ldpc_hip_bp *h, int32_t mode) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (mode < -1 || mode > 2 || mode == 1)
        return fail(LDPC_HIP_ERR_INVALID, "mode must be -1 (automatic), 0 (matrix in LDS) or 2 (OSD-0: matrix in HBM)");
    h->osd_reg = mode != 0;
    h->osd_big = mode == 2;
    return LDPC_HIP_OK;
}

int ldpc_hip_bposd_decode_batch_async(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch,
                                      uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    HIPCHK(hipSetDevice(h->device));
    return mark_queued(h, bposd_device(h, h->osd_method, h->osd_order, synd, batch, decoding, llr, iters, conv));
}

int ldpc_hip_bp_decode_batch_async(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch,
                                   uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    if (batch > (1ll << 40)) return fail(LDPC_HIP_ERR_INVALID, "batch too large");
    HIPCHK(hipSetDevice(h->device));
    return mark_queued(h, decode_device(h, synd, batch, decoding, llr, iters, conv));
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

// The caller's result arrays are usually fresh allocations (np.empty) whose pages do not exist yet: writing 5 GB of log-ratios
// into them is 1.3 million first-touch faults.  Where the kernel offers transparent huge pages on request (THP mode "madvise"),
// ask for them on the 2 MiB-aligned inside of a large array: 512 times fewer faults.  Advice only -- no effect where THP is off.
static void advise_huge_pages(void *p, size_t bytes) {
    if (!p || bytes < ((size_t)64 << 20)) return;
    const uintptr_t two_mb = (uintptr_t)2 << 20;
    const uintptr_t lo = ((uintptr_t)p + two_mb - 1) & ~(two_mb - 1), hi = ((uintptr_t)p + bytes) & ~(two_mb - 1);
    if (hi > lo) (void)madvise((void *)lo, (size_t)(hi - lo), MADV_HUGEPAGE);
}

// Copy between pinned staging and the caller's pageable arrays, split over a few threads: the caller's pages are usually untouched
// (np.empty), and first-touch faults -- not memory bandwidth -- bound a single thread at 2 - 4 GB/s.
static void host_copy_parallel(void *dst, const void *src, size_t bytes) {
    const size_t slice_min = (size_t)4 << 20;
    int nt = (int)(bytes / slice_min);
    static const int nt_max = [] { const unsigned hc = std::thread::hardware_concurrency(); return hc >= 32 ? 16 : hc >= 8 ? (int)(hc / 2) : 2; }();
    if (nt > nt_max) nt = nt_max;
    if (nt < 2) { std::memcpy(dst, src, bytes); return; }
    const size_t per = ((bytes / (size_t)nt) + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (int q = 1; q < nt; ++q) {
        const size_t off = per * (size_t)q;
        if (off >= bytes) break;
        const size_t len = bytes - off < per ? bytes - off : per;
        th.emplace_back([=]() { std::memcpy((char *)dst + off, (const char *)src + off, len); });
    }
    std::memcpy(dst, src, per < bytes ? per : bytes);
    for (auto &t : th) t.join();
}

// A large batch whose buffers all live in (pageable) host memory -- the reference API's only mode: NumPy in, NumPy out
// (_bp_decoder.pyx:642-695).  One H2D copy, the kernels, four D2H copies in sequence leave the GPU idle while 0.3 - 6 GB cross PCIe
// through the runtime's own staging.  Instead the batch is cut into chunks of whole tiles that move through pinned buffers, three in
// flight: while the kernels decode chunk c, chunk c + 1's syndromes are on their way in (copy stream), chunk c - 1's results on their
// way out (another copy stream) and a helper thread copies chunk c - 2's results from its pinned buffer into the caller's arrays
// (with two buffers and the calling thread doing that copy, "out over PCIe" and "into the caller's array" of one chunk followed one
// another inside every period: 42 ms against 29 ms of kernels per 2 816-row chunk with log-ratios).  Rows are independent under the
// parallel and the fixed-order serial schedule, so chunking changes no result (the schedules that carry state from row to row are
// not chunked).  ldpc_hip_bp_last_kernel_ms then describes the LAST chunk only.  Returns 1 if the staging could not be set up.
static int decode_batch_pipelined(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                                  double *llr, int32_t *iters, uint8_t *conv, int64_t rows) {
    auto &P = h->pipe;
    constexpr int NB = ldpc_hip_bp::HostPipe::NB;
    const size_t m = (size_t)h->m, n = (size_t)h->n;
    auto up = [](size_t v) { return (v + 4095) & ~(size_t)4095; };
    const size_t R = (size_t)rows;
    // log-ratios into page-locked memory of the caller's (ldpc_hip_host_alloc, hipHostRegister): the copy engine writes them where they
    // belong -- no staging buffer, no host-side copy for the one array that is eight ninths of the results
    const bool llr_direct = llr && is_pinned_host_ptr(llr) && is_pinned_host_ptr(llr + (size_t)batch * n - 1) && !h->on("NO_DIRECT_LLR");
    const size_t o_llr = up(R * n), o_it = o_llr + (llr && !llr_direct ? up(R * n * 8) : 0), o_cv = o_it + (iters ? up(R * 4) : 0), out_bytes = o_cv + (conv ? up(R) : 0);
    const size_t in_bytes = up(R * m ? R * m : 1);
    int rc;
    // streams, events, pinned and device staging: if any of it cannot be had (pinned memory is a limited resource) the caller takes
    // the one-shot path instead -- nothing has been queued yet
    auto setup = [&]() -> bool {
        if (!P.s_in) {
            if (hipStreamCreateWithFlags(&P.s_in, hipStreamNonBlocking) != hipSuccess) { P.s_in = nullptr; return false; }
            if (hipStreamCreateWithFlags(&P.s_out, hipStreamNonBlocking) != hipSuccess) return false;
            for (int q = 0; q < NB; ++q)
                if (hipEventCreateWithFlags(&P.ev_in[q], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&P.ev_cmp[q], hipEventDisableTiming) != hipSuccess ||
                    hipEventCreateWithFlags(&P.ev_out[q], hipEventDisableTiming) != hipSuccess) return false;
        }
        if (!P.s_out || !P.ev_out[NB - 1]) return false;  // (an earlier attempt got stuck half way)
        for (int q = 0; q < NB; ++q) {
            if (P.pin_in_cap < in_bytes || !P.pin_in[q]) {
                if (P.pin_in[q]) { (void)hipHostFree(P.pin_in[q]); P.pin_in[q] = nullptr; }
                if (hipHostMalloc((void **)&P.pin_in[q], in_bytes, hipHostMallocDefault) != hipSuccess) { P.pin_in[q] = nullptr; P.pin_in_cap = 0; return false; }
            }
            if (P.pin_out_cap < out_bytes || !P.pin_out[q]) {
                if (P.pin_out[q]) { (void)hipHostFree(P.pin_out[q]); P.pin_out[q] = nullptr; }
                if (hipHostMalloc((void **)&P.pin_out[q], out_bytes, hipHostMallocDefault) != hipSuccess) { P.pin_out[q] = nullptr; P.pin_out_cap = 0; return false; }
            }
            if (P.d_in[q].ensure(in_bytes) || P.d_dec[q].ensure(R * n ? R * n : 1) || (llr && P.d_llr[q].ensure(R * n * 8 ? R * n * 8 : 1)) ||
                P.d_it[q].ensure(R * 4) || P.d_cv[q].ensure(R)) return false;
        }
        if (P.pin_in_cap < in_bytes) P.pin_in_cap = in_bytes;
        if (P.pin_out_cap < out_bytes) P.pin_out_cap = out_bytes;
        return true;
    };
    HIPCHK(hipStreamSynchronize(h->stream));  // (an earlier asynchronous call may still use the workspace)
    if (!setup()) {
        (void)hipGetLastError();
        g_last_error.clear();
        return 1;  // "not here": decode_batch_staged carries on with the one-shot path
    }
    advise_huge_pages(decoding, (size_t)batch * n);
    if (llr && !llr_direct) advise_huge_pages(llr, (size_t)batch * n * 8);
    // The chunks: `rows` each -- but with log-ratios the LAST chunk's results (90 KB a row on the n = 10 000 code: 1.5 GB for 16 384 rows, ~30 ms
    // of PCIe) cross after its kernels with nothing left to overlap them, so the last quarter of the batch goes in chunks that halve down
    // to 4 096 rows: what is left exposed is a quarter of that.  (LDPC_HIP_HOST_TAPER=0: uniform chunks; measurement.)
    std::vector<int64_t> chunk_start, chunk_rows;
    {
        int64_t at = 0;
        const bool taper = llr && h->sw("HOST_TAPER") != 0 && rows >= 8192 && batch >= 3 * rows;
        while (at < batch) {
            int64_t r = rows;
            const int64_t left = batch - at;
            if (taper && left <= rows) {
                r = left / 2 / LDPC_WAVE * LDPC_WAVE;
                if (r < 4096 || left < 8192) r = left;
            }
            if (r > left) r = left;
            chunk_start.push_back(at);
            chunk_rows.push_back(r);
            at += r;
        }
    }
    const int64_t chunks = (int64_t)chunk_rows.size();
    auto rows_of = [&](int64_t c) { return chunk_rows[(size_t)c]; };
    auto start_of = [&](int64_t c) { return chunk_start[(size_t)c]; };
    // the helper: chunk after chunk, wait for its results to have landed in the pinned buffer, copy them into the caller's arrays
    std::atomic<int64_t> queued{0}, drained{0};
    std::atomic<int> drain_err{0};
    std::atomic<bool> stop{false};
    const int device = h->device;
    const bool clocked = h->on("HOST_PIPE_TIMING");  // (measurement: where the chunks' time goes, printed at the end)
    double t_wait_queue = 0, t_wait_event = 0, t_copy_out = 0, t_main_in = 0, t_main_wait = 0;
    auto now = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    std::thread drainer([&]() {
        if (hipSetDevice(device) != hipSuccess) { drain_err = 1; return; }
        for (int64_t c = 0; c < chunks; ++c) {
            const double t0 = clocked ? now() : 0;
            while (queued.load(std::memory_order_acquire) <= c) {
                if (stop.load(std::memory_order_acquire)) return;
                std::this_thread::yield();
            }
            const int q = (int)(c % NB);
            const size_t r = (size_t)rows_of(c), b0 = (size_t)start_of(c);
            const double t1 = clocked ? now() : 0;
            if (hipEventSynchronize(P.ev_out[q]) != hipSuccess) { drain_err = 1; return; }
            const double t2 = clocked ? now() : 0;
            host_copy_parallel(decoding + b0 * n, P.pin_out[q], r * n);
            if (llr && !llr_direct) host_copy_parallel(llr + b0 * n, P.pin_out[q] + o_llr, r * n * 8);
            if (iters) std::memcpy(iters + b0, P.pin_out[q] + o_it, r * 4);
            if (conv) std::memcpy(conv + b0, P.pin_out[q] + o_cv, r);
            if (clocked) { const double t3 = now(); t_wait_queue += t1 - t0; t_wait_event += t2 - t1; t_copy_out += t3 - t2; }
            drained.store(c + 1, std::memory_order_release);
        }
    });
    auto finish = [&](int code) {  // (every exit: the helper must be gone before its captures are)
        if (code) stop.store(true, std::memory_order_release);
        drainer.join();
        if (code) {  // a failed call leaves no copy in flight: the next call reuses the slots, their staging buffers and their events
            (void)hipStreamSynchronize(P.s_in);
            (void)hipStreamSynchronize(P.s_out);
            (void)hipGetLastError();
        }
        return code;
    };
#define PIPECHK(expr)                                                                                                            \
    do {                                                                                                                         \
        hipError_t _e = (expr);                                                                                                  \
        if (_e != hipSuccess) return finish(fail(LDPC_HIP_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__)); \
    } while (0)
    for (int64_t c = 0; c < chunks; ++c) {
        const int q = (int)(c % NB);
        const size_t r = (size_t)rows_of(c), b0 = (size_t)start_of(c);
        const double tm0 = clocked ? now() : 0;
        if (c >= NB) {
            // slot q carried chunk c - NB: its upload has completed (pin_in free), and the helper has emptied its pin_out
            PIPECHK(hipEventSynchronize(P.ev_in[q]));
            while (drained.load(std::memory_order_acquire) <= c - NB) {
                if (drain_err.load()) return finish(fail(LDPC_HIP_ERR_DEVICE, "the copy-out thread of the pipelined host path failed"));
                std::this_thread::yield();
            }
        }
        const double tm1 = clocked ? now() : 0;
        host_copy_parallel(P.pin_in[q], synd + b0 * m, r * m);
        if (clocked) { t_main_wait += tm1 - tm0; t_main_in += now() - tm1; }
        if (c >= NB) PIPECHK(hipStreamWaitEvent(P.s_in, P.ev_cmp[q], 0));  // d_in[q] was chunk c - NB's input
        if (r * m) PIPECHK(hipMemcpyAsync(P.d_in[q].p, P.pin_in[q], r * m, hipMemcpyHostToDevice, P.s_in));
        PIPECHK(hipEventRecord(P.ev_in[q], P.s_in));
        // compute: after this chunk's upload (the results of chunk c - NB have left d_dec[q]: the helper waited for that)
        PIPECHK(hipStreamWaitEvent(h->stream, P.ev_in[q], 0));
        if ((rc = decode_device(h, (const uint8_t *)P.d_in[q].p, (int64_t)r, (uint8_t *)P.d_dec[q].p, llr ? (double *)P.d_llr[q].p : nullptr,
                                (int32_t *)P.d_it[q].p, (uint8_t *)P.d_cv[q].p))) return finish(rc);
        PIPECHK(hipEventRecord(P.ev_cmp[q], h->stream));
        PIPECHK(hipStreamWaitEvent(P.s_out, P.ev_cmp[q], 0));
        if (r * n) PIPECHK(hipMemcpyAsync(P.pin_out[q], P.d_dec[q].p, r * n, hipMemcpyDeviceToHost, P.s_out));
        if (llr && r * n) PIPECHK(hipMemcpyAsync(llr_direct ? (void *)(llr + b0 * n) : (void *)(P.pin_out[q] + o_llr), P.d_llr[q].p, r * n * 8, hipMemcpyDeviceToHost, P.s_out));
        if (iters) PIPECHK(hipMemcpyAsync(P.pin_out[q] + o_it, P.d_it[q].p, r * 4, hipMemcpyDeviceToHost, P.s_out));
        if (conv) PIPECHK(hipMemcpyAsync(P.pin_out[q] + o_cv, P.d_cv[q].p, r, hipMemcpyDeviceToHost, P.s_out));
        PIPECHK(hipEventRecord(P.ev_out[q], P.s_out));
        queued.store(c + 1, std::memory_order_release);
    }
#undef PIPECHK
    drainer.join();
    if (drain_err.load()) return fail(LDPC_HIP_ERR_DEVICE, "the copy-out thread of the pipelined host path failed");
    HIPCHK(hipStreamSynchronize(h->stream));
    if (clocked)
        fprintf(stderr, "[ldpc_hip] host pipeline: %lld chunks of %lld rows; calling thread: waited for a free slot %.1f ms, copied syndromes in %.1f ms; "
                        "copy-out thread: waited for a queued chunk %.1f ms, for its results to land %.1f ms, copied out %.1f ms\n",
                (long long)chunks, (long long)rows, t_main_wait * 1e3, t_main_in * 1e3, t_wait_queue * 1e3, t_wait_event * 1e3, t_copy_out * 1e3);
    return LDPC_HIP_OK;
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
    // A small call whose buffers are all on the host (the reference's only mode: one syndrome per decode()): five copy commands
    // and their completion cost more than the kernels.  The kernels work in a host-mapped block instead.
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_dec = up16(B * m), o_llr = o_dec + up16(B * n), o_it = o_llr + up16(B * n * 8), o_cv = o_it + up16(B * 4), pin_need = o_cv + up16(B);
    if (h_synd && h_dec && (!llr || h_llr) && (!iters || h_it) && (!conv || h_cv) && pin_need <= ldpc_hip_bp::PIN_MAIL && !h->on("NO_PINNED_PATH")) {
        if (!h->pin_host) {
            if (hipHostMalloc((void **)&h->pin_host, ldpc_hip_bp::PIN_BYTES, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||
                hipHostGetDevicePointer((void **)&h->pin_dev, h->pin_host, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (h->pin_host) (void)hipHostFree(h->pin_host);
                h->pin_host = h->pin_dev = nullptr;
            }
        }
        if (h->pin_host) {
            HIPCHK(hipStreamSynchronize(h->stream));  // (a previous asynchronous call may still use the block's neighbours -- and its results)
            std::memcpy(h->pin_host, synd, B * m);
            unsigned char *dv = h->pin_dev;
            // BP + OSD needs log-ratios and flags whether the caller asks for them or not: the block has room for them
            double *p_llr = (llr || osd >= 0) ? (double *)(dv + o_llr) : nullptr;
            uint8_t *p_cv = (conv || osd >= 0) ? (uint8_t *)(dv + o_cv) : nullptr;
            int32_t *p_it = iters ? (int32_t *)(dv + o_it) : nullptr;
            if (batch == 1 && osd < 0 && !h->on("TIME_SMALL_CALLS")) {  // ONE syndrome: the resident workgroup, where one applies
                bool took = false;
                if ((rc = decode_onchip_resident(h, llr != nullptr, &took))) return rc;
                if (took) {
                    std::memcpy(decoding, h->pin_host + o_dec, B * n);
                    if (llr) std::memcpy(llr, h->pin_host + o_llr, B * n * 8);
                    if (iters) std::memcpy(iters, h->pin_host + o_it, B * 4);
                    if (conv) std::memcpy(conv, h->pin_host + o_cv, B);
                    return LDPC_HIP_OK;
                }
            }
            h->untimed_call = batch <= 4 && !h->on("TIME_SMALL_CALLS");
            rc = osd >= 0 ? bposd_device(h, osd ? h->osd_method : 1, osd ? h->osd_order : 0, dv, batch, dv + o_dec, p_llr, p_it, p_cv)
                          : decode_device(h, dv, batch, dv + o_dec, p_llr, p_it, p_cv);
            h->untimed_call = false;
            if (rc) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));
            std::memcpy(decoding, h->pin_host + o_dec, B * n);
            if (llr) std::memcpy(llr, h->pin_host + o_llr, B * n * 8);
            if (iters) std::memcpy(iters, h->pin_host + o_it, B * 4);
            if (conv) std::memcpy(conv, h->pin_host + o_cv, B);
            return LDPC_HIP_OK;
        }
    }
    // everything on the host, BP only, rows independent of one another, and enough of them for several chunks: pipelined
    if (h_synd && h_dec && (!llr || h_llr) && (!iters || h_it) && (!conv || h_cv) && osd < 0 && !h->random_serial && h->schedule != 2 &&
        !h->on("NO_HOST_PIPELINE")) {
        // chunk: ~256 MiB of staged results, between 1 024 and 16 384 rows (whole tiles), or what LDPC_HIP_HOST_CHUNK_ROWS says; log-ratios
        // that go straight into page-locked memory of the caller's (decode_batch_pipelined) are not staged and do not count
        const bool llr_staged = llr && !(is_pinned_host_ptr(llr) && !h->on("NO_DIRECT_LLR"));
        const size_t per_row = n * (llr_staged ? 9 : 1) + m + 5;
        int64_t rows = (int64_t)(((size_t)256 << 20) / (per_row ? per_row : 1));
        if (rows > 16384) rows = 16384;
        if (rows < 1024) rows = 1024;
        if (h->sw("HOST_CHUNK_ROWS") > 0) rows = h->sw("HOST_CHUNK_ROWS");
        rows = (rows + LDPC_WAVE - 1) / LDPC_WAVE * LDPC_WAVE;
        if (batch >= 3 * rows && (size_t)batch * per_row >= ((size_t)64 << 20)) {
            const int prc = decode_batch_pipelined(h, synd, batch, decoding, llr, iters, conv, rows);
            if (prc <= 0) return prc;  // (1: its staging could not be set up -- the one-shot path below)
        }
    }
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
    hipLaunchKernelGGL(gf2_mulvec_kernel, flat_grid((size_t)(total)), dim3(256), 0, h->stream,
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

// device-side conversion between one byte per bit and b8 rows; both buffers are device pointers, work is queued on the
// handle's stream (no synchronisation): meant for packing results before they cross a link (PCIe, xGMI)
int ldpc_hip_pack_b8(ldpc_hip_bp *h, const uint8_t *bytes, int64_t batch, int32_t bits, uint8_t *packed) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0 || bits < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch or bits");
    if (batch == 0 || bits == 0) return LDPC_HIP_OK;
    if (!bytes || !packed) return fail(LDPC_HIP_ERR_INVALID, "null buffer");
    if (!is_device_ptr(bytes) || !is_device_ptr(packed)) return fail(LDPC_HIP_ERR_INVALID, "ldpc_hip_pack_b8 takes device pointers");
    HIPCHK(hipSetDevice(h->device));
    const size_t total = (size_t)batch * (size_t)((bits + 7) / 8);
    hipLaunchKernelGGL(pack_b8_kernel, flat_grid((size_t)(total)), dim3(256), 0, h->stream, bytes, batch, bits, packed);
    HIPCHK(hipGetLastError());
    return mark_queued(h, LDPC_HIP_OK);
}

int ldpc_hip_unpack_b8(ldpc_hip_bp *h, const uint8_t *packed, int64_t batch, int32_t bits, uint8_t *bytes) {
    if (!h) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0 || bits < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch or bits");
    if (batch == 0 || bits == 0) return LDPC_HIP_OK;
    if (!bytes || !packed) return fail(LDPC_HIP_ERR_INVALID, "null buffer");
    if (!is_device_ptr(bytes) || !is_device_ptr(packed)) return fail(LDPC_HIP_ERR_INVALID, "ldpc_hip_unpack_b8 takes device pointers");
    HIPCHK(hipSetDevice(h->device));
    const size_t total = (size_t)batch * (size_t)bits;
    hipLaunchKernelGGL(unpack_b8_kernel, flat_grid((size_t)(total)), dim3(256), 0, h->stream, packed, batch, bits, bytes);
    HIPCHK(hipGetLastError());
    return mark_queued(h, LDPC_HIP_OK);
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
    if (m) hipLaunchKernelGGL(unpack_b8_kernel, flat_grid((size_t)(B * m)), dim3(256), 0, h->stream, d_in, batch, h->m, d_synd);
    HIPCHK(hipGetLastError());
    const bool h_it = iters && !is_device_ptr(iters), h_cv = conv && !is_device_ptr(conv);
    int32_t *d_it = iters;
    uint8_t *d_cv = conv;
    if (h_it) { if ((rc = h->st_iters.ensure(B * 4))) return rc; d_it = (int32_t *)h->st_iters.p; }
    if (h_cv) { if ((rc = h->st_conv.ensure(B))) return rc; d_cv = (uint8_t *)h->st_conv.p; }
    if ((rc = with_osd ? bposd_device(h, h->osd_method, h->osd_order, d_synd, batch, d_dec, nullptr, d_it, d_cv)
                       : decode_device(h, d_synd, batch, d_dec, nullptr, d_it, d_cv))) return rc;
    hipLaunchKernelGGL(zero_shot_shortcut_kernel, flat_grid((size_t)(B)), dim3(256), 0, h->stream, d_in, batch, h->m, h->n,
                       d_dec, d_it, d_cv);
    size_t off = 0;
    if ((rc = h->b8_out.ensure(B * (kb + nb) ? B * (kb + nb) : 1))) return rc;
    uint8_t *d_obs = obs_b8, *d_dec8 = decoding_b8;
    const bool h_obs = obs_b8 && !is_device_ptr(obs_b8), h_dec8 = decoding_b8 && !is_device_ptr(decoding_b8);
    if (h_obs) { d_obs = (uint8_t *)h->b8_out.p; off = B * kb; }
    if (h_dec8) d_dec8 = (uint8_t *)h->b8_out.p + off;
    if (obs_b8 && kb)
        hipLaunchKernelGGL(observables_b8_kernel, flat_grid((size_t)(B * kb)), dim3(256), 0, h->stream,
                           (const int32_t *)h->obs_row_ptr.p, (const int32_t *)h->obs_col_idx.p, h->obs_k, h->n, d_dec, batch, d_obs);
    if (decoding_b8 && nb)
        hipLaunchKernelGGL(pack_b8_kernel, flat_grid((size_t)(B * nb)), dim3(256), 0, h->stream, d_dec, batch, h->n, d_dec8);
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
        hipLaunchKernelGGL(gen_bsc_syndromes_kernel, flat_grid((size_t)(total)), dim3(256), 0,
                           h->stream, h->d_row_ptr, h->d_col_idx, h->m, h->n, seed, threshold, shot0,
                           batch, d_s);
    }
    if (errors && h->n > 0) {
        const int64_t total = batch * h->n;
        hipLaunchKernelGGL(gen_bsc_errors_kernel, flat_grid((size_t)(total)), dim3(256), 0,
                           h->stream, h->n, seed, threshold, shot0, batch, d_e);
    }
    HIPCHK(hipGetLastError());
    if (h_s) HIPCHK(hipMemcpyAsync(syndromes, d_s, B * m, hipMemcpyDeviceToHost, h->stream));
    if (h_e) HIPCHK(hipMemcpyAsync(errors, d_e, B * n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return LDPC_HIP_OK;
}

}  // extern "C"
