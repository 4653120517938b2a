// multi_device.h -- ldpc_hip_bp_multi: one decoder object over several GPUs of a node, inside ONE process (host code only)
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes this file last).
//
// Syndromes are independent, so a batch is cut into contiguous row ranges (boundaries on whole
// 64-syndrome tiles), one per entry of `device_ids`; every range is decoded by an ordinary single-device handle, driven
// by its own host thread for the duration of the call -- no data-path collective, nothing shared but the read-only H.
// What remains is moving rows to and from the caller's buffers:
//   * host buffers: each device stages ITS rows over its own PCIe link, straight from / into the caller's arrays;
//   * device buffers (all on one GPU, the "root"): a device other than the root receives its rows by peer copy
//     (hipMemcpyPeerAsync: xGMI), returns log-ratios / iteration counts / flags the same way, and its hard decisions
//     BIT-PACKED (1/8 of the bytes on the link, ldpc_hip_pack_b8) into a scratch buffer on the root, where the root's
//     handle unpacks them into the caller's array.  The root must be one of `device_ids` for the packed route; otherwise
//     the decisions cross as bytes.
// The reference has no counterpart: ldpc::bp::BpDecoder is single-threaded (bp.hpp:129-140, OpenMP is a stub).
#pragma once

#include <thread>

struct MultiDev {
    ldpc_hip_bp *h = nullptr;
    int device = 0;
    DeviceBuf in, dec, llr, it, cv, b8;  // staging on this device for rows that live on another GPU
    DeviceBuf root_b8;                   // on the ROOT device: this shard's packed decisions, waiting to be unpacked there
    int root_b8_device = -1;
    float kernel_ms = 0.f;
    int rc = 0;
    std::string err;
};

struct ldpc_hip_bp_multi {
    std::vector<MultiDev> devs;
    int32_t m = 0, n = 0;
    bool force_staging = false;  // testing: treat device buffers as foreign even on the GPU they live on
};

static int pointer_device(const void *p) {  // -1: host memory
    if (!p) return -1;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    if (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) return attr.device;
    return -1;
}

// [lo, hi) of shard d: equal numbers of whole 64-syndrome tiles, the remainder tiles go to the first shards
static void multi_range(int64_t batch, int ndev, int d, int64_t &lo, int64_t &hi) {
    const int64_t tiles = (batch + LDPC_WAVE - 1) / LDPC_WAVE, base = tiles / ndev, rem = tiles % ndev;
    const int64_t t0 = d * base + (d < rem ? d : rem), t1 = t0 + base + (d < rem ? 1 : 0);
    lo = t0 * LDPC_WAVE < batch ? t0 * LDPC_WAVE : batch;
    hi = t1 * LDPC_WAVE < batch ? t1 * LDPC_WAVE : batch;
}

extern "C" {

int ldpc_hip_bp_multi_create(const ldpc_hip_bp_desc *desc, const int32_t *device_ids, int32_t ndev, ldpc_hip_bp_multi **out) {
    if (!desc || !out || !device_ids) return fail(LDPC_HIP_ERR_INVALID, "null descriptor, device list or output");
    *out = nullptr;
    if (ndev < 1 || ndev > 64) return fail(LDPC_HIP_ERR_INVALID, "ndev must be in [1, 64]");
    int count = 0;
    HIPCHK(hipGetDeviceCount(&count));
    for (int i = 0; i < ndev; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= count)
            return fail(LDPC_HIP_ERR_INVALID, "device_ids[%d] = %d, but %d device(s) are visible", i, device_ids[i], count);
    auto *mh = new ldpc_hip_bp_multi;
    mh->m = desc->m;
    mh->n = desc->n;
    mh->devs.resize((size_t)ndev);
    for (int i = 0; i < ndev; ++i) {
        ldpc_hip_bp_desc d = *desc;
        d.device = device_ids[i];
        mh->devs[(size_t)i].device = device_ids[i];
        const int rc = ldpc_hip_bp_create(&d, &mh->devs[(size_t)i].h);
        if (rc) {
            const std::string keep = g_last_error;
            for (auto &md : mh->devs) ldpc_hip_bp_destroy(md.h);
            delete mh;
            g_last_error = keep;
            return rc;
        }
    }
    // peer access lets a copy between two GPUs go directly over xGMI; without it the runtime stages through the host
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < ndev; ++j) {
            if (device_ids[i] == device_ids[j]) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, device_ids[i], device_ids[j]) == hipSuccess && can) {
                (void)hipSetDevice(device_ids[i]);
                (void)hipDeviceEnablePeerAccess(device_ids[j], 0);  // "already enabled" is fine
                (void)hipGetLastError();
            }
        }
    *out = mh;
    return LDPC_HIP_OK;
}

void ldpc_hip_bp_multi_destroy(ldpc_hip_bp_multi *mh) {
    if (!mh) return;
    for (auto &md : mh->devs) {
        (void)hipSetDevice(md.device);
        for (DeviceBuf *b : {&md.in, &md.dec, &md.llr, &md.it, &md.cv, &md.b8}) b->release();
        if (md.root_b8.p) { (void)hipSetDevice(md.root_b8_device); md.root_b8.release(); }
        ldpc_hip_bp_destroy(md.h);
    }
    delete mh;
}

int32_t ldpc_hip_bp_multi_devices(const ldpc_hip_bp_multi *mh) { return mh ? (int32_t)mh->devs.size() : 0; }

ldpc_hip_bp *ldpc_hip_bp_multi_handle(ldpc_hip_bp_multi *mh, int32_t i) {
    if (!mh || i < 0 || i >= (int32_t)mh->devs.size()) return nullptr;
    return mh->devs[(size_t)i].h;
}

int ldpc_hip_bp_multi_set_staging(ldpc_hip_bp_multi *mh, int32_t force) {
    if (!mh) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    mh->force_staging = force != 0;
    return LDPC_HIP_OK;
}

int ldpc_hip_bp_multi_last_kernel_ms(ldpc_hip_bp_multi *mh, float *ms_per_device) {
    if (!mh || !ms_per_device) return fail(LDPC_HIP_ERR_INVALID, "null argument");
    for (size_t i = 0; i < mh->devs.size(); ++i) ms_per_device[i] = mh->devs[i].kernel_ms;
    return LDPC_HIP_OK;
}

}  // extern "C"

// one shard on one device; runs on that device's own host thread
static int multi_decode_shard(ldpc_hip_bp_multi *mh, MultiDev &md, int osd, const uint8_t *synd, int64_t rows, uint8_t *decoding,
                              double *llr, int32_t *iters, uint8_t *conv, int ptr_dev, bool packed_route) {
    if (rows == 0) return LDPC_HIP_OK;
    ldpc_hip_bp *h = md.h;
    const size_t R = (size_t)rows, m = (size_t)mh->m, n = (size_t)mh->n;
    HIPCHK(hipSetDevice(md.device));
    int rc;
    const bool foreign = ptr_dev >= 0 && (ptr_dev != md.device || mh->force_staging);
    if (!foreign) {
        // host rows (the handle stages them over this GPU's PCIe link) or rows that already live on this GPU
        rc = osd < 0 ? ldpc_hip_bp_decode_batch(h, synd, rows, decoding, llr, iters, conv)
           : osd == 0 ? ldpc_hip_bposd0_decode_batch(h, synd, rows, decoding, llr, iters, conv)
                      : ldpc_hip_bposd_decode_batch(h, synd, rows, decoding, llr, iters, conv);
        if (rc) return rc;
        return ldpc_hip_bp_last_kernel_ms(h, &md.kernel_ms);
    }
    // rows on another GPU: peer copy in, decode here, peer copy out (hard decisions bit-packed when the root can unpack them)
    const size_t nb8 = (n + 7) / 8;
    if ((rc = md.in.ensure(R * m ? R * m : 1)) || (rc = md.dec.ensure(R * n ? R * n : 1)) || (llr && (rc = md.llr.ensure(R * n * 8 ? R * n * 8 : 1))) ||
        (iters && (rc = md.it.ensure(R * 4))) || (conv && (rc = md.cv.ensure(R))) || (packed_route && (rc = md.b8.ensure(R * nb8 ? R * nb8 : 1))))
        return rc;
    hipStream_t st = h->stream;
    if (m) HIPCHK(hipMemcpyPeerAsync(md.in.p, md.device, synd, ptr_dev, R * m, st));
    uint8_t *d_dec = (uint8_t *)md.dec.p;
    double *d_llr = llr ? (double *)md.llr.p : nullptr;
    int32_t *d_it = iters ? (int32_t *)md.it.p : nullptr;
    uint8_t *d_cv = conv ? (uint8_t *)md.cv.p : nullptr;
    rc = osd < 0 ? ldpc_hip_bp_decode_batch_async(h, (const uint8_t *)md.in.p, rows, d_dec, d_llr, d_it, d_cv)
       : osd == 0 ? ldpc_hip_bposd0_decode_batch_async(h, (const uint8_t *)md.in.p, rows, d_dec, d_llr, d_it, d_cv)
                  : ldpc_hip_bposd_decode_batch_async(h, (const uint8_t *)md.in.p, rows, d_dec, d_llr, d_it, d_cv);
    if (rc) return rc;
    if (n) {
        if (packed_route) {
            if ((rc = ldpc_hip_pack_b8(h, d_dec, rows, (int32_t)n, (uint8_t *)md.b8.p))) return rc;
            HIPCHK(hipMemcpyPeerAsync(md.root_b8.p, ptr_dev, md.b8.p, md.device, R * nb8, st));
        } else {
            HIPCHK(hipMemcpyPeerAsync(decoding, ptr_dev, d_dec, md.device, R * n, st));
        }
        if (llr) HIPCHK(hipMemcpyPeerAsync(llr, ptr_dev, d_llr, md.device, R * n * 8, st));
    }
    if (iters) HIPCHK(hipMemcpyPeerAsync(iters, ptr_dev, d_it, md.device, R * 4, st));
    if (conv) HIPCHK(hipMemcpyPeerAsync(conv, ptr_dev, d_cv, md.device, R, st));
    HIPCHK(hipStreamSynchronize(st));
    return ldpc_hip_bp_last_kernel_ms(h, &md.kernel_ms);
}

// The schedules that keep state in the decoder object (serial_relative's re-sorted order, the random schedule's order and
// generator: bp.hpp:467-483) promise "every row starts from the state at the time of the call, the call leaves the state
// of its LAST row" (decode_serial_random above).  With one handle per GPU that state has to be ONE state: handle 0 is the
// authority before a call (which also settles random_schedule_seed = 0, where every handle read the clock on its own),
// the handle that decoded the batch's last row is the authority after it.
static bool multi_schedule_has_state(const ldpc_hip_bp *h) { return h->random_serial || h->schedule == 2; }
static void multi_copy_schedule_state(ldpc_hip_bp_multi *mh, int from) {
    ldpc_hip_bp *src = mh->devs[(size_t)from].h;
    int entry_device = -1;
    (void)hipGetDevice(&entry_device);
    struct Restore { int d; ~Restore() { if (d >= 0) (void)hipSetDevice(d); } } restore{entry_device};
    for (size_t d = 0; d < mh->devs.size(); ++d) {
        ldpc_hip_bp *dst = mh->devs[d].h;
        if (dst == src) continue;
        const bool same = dst->sched_state == src->sched_state && dst->sched_rng == src->sched_rng && dst->sched_seed_raw == src->sched_seed_raw;
        dst->sched_state = src->sched_state;
        dst->sched_rng = src->sched_rng;
        dst->sched_seed_raw = src->sched_seed_raw;
        if (same) continue;
        // The random schedule's ring of per-iteration orders (host_serial.h: random_orders_*) belongs to that state: without it the
        // receiving handle finds its ring stale at the next call and regenerates max_iter x n orders (n^2 shuffles and a 4 n^2-byte
        // upload at the default max_iter = n).  The source's ring is current, so the receiver takes it over: bookkeeping + one
        // device-to-device copy of the table.  Any failure just leaves the receiver's ring invalid (it is then rebuilt as before).
        dst->rnd.valid = false;
        const auto &r = src->rnd;
        if (!r.valid || r.rows <= 0 || r.n <= 0 || !src->sched_orders.p) continue;
        const size_t bytes = (size_t)r.rows * (size_t)r.n * sizeof(int32_t);
        if (hipSetDevice(dst->device) != hipSuccess || hipStreamSynchronize(dst->stream) != hipSuccess || dst->sched_orders.ensure(bytes + 16) ||
            hipMemcpyPeer(dst->sched_orders.p, dst->device, src->sched_orders.p, src->device, bytes) != hipSuccess) {
            (void)hipGetLastError();
            continue;
        }
        {   // ... and the same rows level-major with their level bounds (the level kernels read those)
            const size_t ptr_bytes = (size_t)r.rows * ((size_t)r.n + 2) * sizeof(int32_t);
            if (!src->sched_lvl_bits.p || !src->sched_lvl_ptr.p || dst->sched_lvl_bits.ensure(bytes + 16) || dst->sched_lvl_ptr.ensure(ptr_bytes + 16) ||
                hipMemcpyPeer(dst->sched_lvl_bits.p, dst->device, src->sched_lvl_bits.p, src->device, bytes) != hipSuccess ||
                hipMemcpyPeer(dst->sched_lvl_ptr.p, dst->device, src->sched_lvl_ptr.p, src->device, ptr_bytes) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
        }
        dst->rnd = r;
    }
}

extern "C" int ldpc_hip_bp_multi_decode_batch(ldpc_hip_bp_multi *mh, int32_t with_osd, const uint8_t *synd, int64_t batch,
                                              uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv) {
    if (!mh) return fail(LDPC_HIP_ERR_INVALID, "null handle");
    if (batch < 0) return fail(LDPC_HIP_ERR_INVALID, "negative batch");
    if (with_osd < -1 || with_osd > 1) return fail(LDPC_HIP_ERR_INVALID, "with_osd must be -1 (BP only), 0 (BP + OSD-0) or 1 (BP + the handles' OSD method)");
    if (batch == 0) return LDPC_HIP_OK;
    if (!synd || !decoding) return fail(LDPC_HIP_ERR_INVALID, "syndromes and decoding must not be NULL");
    // where the caller's arrays live: all in host memory, or all on ONE GPU
    const int pd = pointer_device(synd);
    for (const void *p : {(const void *)decoding, (const void *)llr, (const void *)iters, (const void *)conv})
        if (p && pointer_device(p) != pd)
            return fail(LDPC_HIP_ERR_INVALID, "ldpc_hip_bp_multi_decode_batch: all buffers must be host memory, or all on the same GPU");
    const int ndev = (int)mh->devs.size();
    const size_t m = (size_t)mh->m, n = (size_t)mh->n, nb8 = (n + 7) / 8;
    int root = -1;  // the handle that lives where the caller's device buffers are
    if (pd >= 0)
        for (int d = 0; d < ndev && root < 0; ++d)
            if (mh->devs[(size_t)d].device == pd) root = d;
    // scratch on the root GPU for the packed decisions of every shard that is decoded elsewhere
    std::vector<char> packed((size_t)ndev, 0);
    for (int d = 0; d < ndev; ++d) {
        MultiDev &md = mh->devs[(size_t)d];
        int64_t lo, hi;
        multi_range(batch, ndev, d, lo, hi);
        md.rc = 0;
        md.kernel_ms = 0.f;
        const bool foreign = pd >= 0 && (pd != md.device || mh->force_staging);
        if (foreign && root >= 0 && n && hi > lo) {
            HIPCHK(hipSetDevice(pd));
            if (md.root_b8.p && md.root_b8_device != pd) { (void)hipSetDevice(md.root_b8_device); md.root_b8.release(); HIPCHK(hipSetDevice(pd)); }
            int rc = md.root_b8.ensure((size_t)(hi - lo) * nb8);
            if (rc) return rc;
            md.root_b8_device = pd;
            packed[(size_t)d] = 1;
        }
    }
    if (pd >= 0) {  // the caller's device buffers may still be written by work queued on the root's stream (pack / unpack of an earlier call)
        HIPCHK(hipSetDevice(pd));
        if (root >= 0) HIPCHK(hipStreamSynchronize(mh->devs[(size_t)root].h->stream));
    }
    const bool stateful = multi_schedule_has_state(mh->devs[0].h);
    if (stateful) multi_copy_schedule_state(mh, 0);
    std::vector<std::thread> threads;
    for (int d = 0; d < ndev; ++d) {
        threads.emplace_back([=, &packed]() {
            MultiDev &md = mh->devs[(size_t)d];
            int64_t lo, hi;
            multi_range(batch, ndev, d, lo, hi);
            md.rc = multi_decode_shard(mh, md, with_osd, synd + (size_t)lo * m, hi - lo, decoding + (size_t)lo * n,
                                       llr ? llr + (size_t)lo * n : nullptr, iters ? iters + lo : nullptr, conv ? conv + lo : nullptr,
                                       pd, packed[(size_t)d] != 0);
            if (md.rc) md.err = g_last_error;  // (thread-local: carry it over to the caller's thread)
        });
    }
    for (auto &t : threads) t.join();
    for (int d = 0; d < ndev; ++d)
        if (mh->devs[(size_t)d].rc) {
            g_last_error = "device " + std::to_string(mh->devs[(size_t)d].device) + ": " + mh->devs[(size_t)d].err;
            return mh->devs[(size_t)d].rc;
        }
    if (stateful)  // the state the batch's last row left (shards without rows decoded nothing and keep the old state until now)
        for (int d = ndev - 1; d >= 0; --d) {
            int64_t lo, hi;
            multi_range(batch, ndev, d, lo, hi);
            if (hi > lo) { multi_copy_schedule_state(mh, d); break; }
        }
    // packed decisions that arrived on the root: unpack them into the caller's array there
    bool any = false;
    for (int d = 0; d < ndev; ++d) {
        if (!packed[(size_t)d]) continue;
        int64_t lo, hi;
        multi_range(batch, ndev, d, lo, hi);
        int rc = ldpc_hip_unpack_b8(mh->devs[(size_t)root].h, (const uint8_t *)mh->devs[(size_t)d].root_b8.p, hi - lo, (int32_t)n, decoding + (size_t)lo * n);
        if (rc) return rc;
        any = true;
    }
    if (any) {
        HIPCHK(hipSetDevice(pd));
        HIPCHK(hipStreamSynchronize(mh->devs[(size_t)root].h->stream));
    }
    return LDPC_HIP_OK;
}
