// host_osd.h -- host side of OSD: list, kernel choice by size, status, second pass for syndromes outside the image
// Part of libldpc_hip.so: included by bp_hip.hip (one translation unit), in the order given there.
#pragma once


// BP, then OSD-0 on the rows BP left unconverged; device pointers, on h->stream
// k = n - rank(H) over GF(2): how many non-pivot columns an OSD elimination leaves (independent of the column order)
static int osd_k(ldpc_hip_bp *h) {
    if (h->osd_k_cached >= 0) return h->osd_k_cached;
    const int m = h->m, n = h->n, W = (n + 63) / 64;
    std::vector<uint64_t> mat((size_t)(m ? m : 1) * (size_t)(W ? W : 1), 0);
    for (int i = 0; i < m; ++i)
        for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
            const int c = h->h_col_idx[(size_t)e];
            mat[(size_t)i * W + (size_t)(c >> 6)] |= 1ull << (c & 63);
        }
    int rank = 0;
    for (int c = 0; c < n && rank < m; ++c) {
        int p = -1;
        for (int i = rank; i < m; ++i)
            if ((mat[(size_t)i * W + (size_t)(c >> 6)] >> (c & 63)) & 1ull) { p = i; break; }
        if (p < 0) continue;
        for (int w = 0; w < W; ++w) std::swap(mat[(size_t)p * W + w], mat[(size_t)rank * W + w]);
        for (int i = 0; i < m; ++i)
            if (i != rank && ((mat[(size_t)i * W + (size_t)(c >> 6)] >> (c & 63)) & 1ull))
                for (int w = 0; w < W; ++w) mat[(size_t)i * W + w] ^= mat[(size_t)rank * W + w];
        ++rank;
    }
    h->osd_k_cached = n - rank;
    return h->osd_k_cached;
}

// After the OSD kernels: did every OSD output solve its syndrome?  (osd_status_kernel; read back with ldpc_hip_bposd_get_status)
// (the array was cleared on the way -- by the BP kernel or by osd_collect_kernel; osd0_reg_kernel writes its rows' entries itself and needs no pass)
static int osd_status_pass(ldpc_hip_bp *h, const OsdArgs &a, int64_t batch) {
    int64_t blocks = batch < 4096 ? batch : 4096;
    hipLaunchKernelGGL(osd_status_kernel, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, h->stream, a, (uint8_t *)h->osd_status.p);
    HIPCHK(hipGetLastError());
    h->osd_status_rows = batch;
    return LDPC_HIP_OK;
}

int bposd_device(ldpc_hip_bp *h, int osd_method, int osd_order, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                        double *llr, int32_t *iters, uint8_t *conv) {
    if (osd_method == 0)  // OSD_OFF: BpOsdDecoder still calls OsdDecoder::decode, which then has no LU object -- refuse instead
        return fail(LDPC_HIP_ERR_INVALID, "osd_method is OSD_OFF");
    const bool higher = osd_method >= 2 && osd_order > 0;  // osd_order == 0 takes the OSD-0 branch whatever the method (osd.hpp:114)
    const size_t B = (size_t)batch, n = (size_t)h->n;
    int rc;
    if (!llr) { if ((rc = h->osd_llr.ensure(B * n * 8 ? B * n * 8 : 1))) return rc; llr = (double *)h->osd_llr.p; }
    if (!conv) { if ((rc = h->osd_conv.ensure(B ? B : 1))) return rc; conv = (uint8_t *)h->osd_conv.p; }
    h->osd_status_rows = 0;
    // The counters of the OSD passes -- {listed, next} of the first at [0..1], of the second (rows outside the image) at [8..9] -- sit behind the
    // BP kernels' work pools, so that ONE fill serves both, and an on-chip BP kernel lists the rows it leaves unconverged and clears the
    // status array itself (osd_hook; otherwise osd_collect_kernel does both below): every launch costs 4 - 5 us whatever it does, and at
    // BASELINE config 5's 8 192 rows the eight small ones around BP and OSD-0 were a tenth of the step.
    if ((rc = h->counter.ensure(work_pool_bytes() + 64))) return rc;
    if ((rc = h->osd_list.ensure((B ? B : 1) * sizeof(int32_t)))) return rc;
    if ((rc = h->osd_status.ensure(B ? B : 1))) return rc;
    unsigned *const osd_ctr = (unsigned *)((char *)h->counter.p + work_pool_bytes());
    h->osd_hook = {(int32_t *)h->osd_list.p, osd_ctr, (uint8_t *)h->osd_status.p, !h->on("OSD_COLLECT_AFTER"), false};
    rc = decode_device(h, synd, batch, decoding, llr, iters, conv);
    h->osd_hook.armed = false;
    if (rc) return rc;
    if (h->m == 0 || h->n == 0) return LDPC_HIP_OK;
    OsdArgs a = {};
    a.m = h->m; a.n = h->n; a.words = (h->n + 1 + 63) / 64;
    a.batch = batch;
    a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx;
    a.synd = synd; a.llr = llr; a.conv = conv; a.decoding = decoding;
    a.method = osd_method; a.order = osd_order; a.wt = h->d_osd_wt;
    if (osd_method == 3 && osd_order > 0) {
        // OSD_CS pairs (i, j), i < j < osd_order, are listed i-major and only those with j < k = n - rank exist in the reference's
        // candidate strings (osd.hpp:91-99; beyond: a write past the string).  Their order in the list -- which is all an index is
        // used for: the earliest of equally light candidates wins -- does not depend on osd_order once it is >= k, so a larger
        // order is the same sweep as order k (and the kernels need not walk millions of pair numbers that name nothing).
        const int k_bound = (double)a.m * a.m * a.words < 4e9 ? osd_k(h) : a.n;
        if (a.order > k_bound) a.order = k_bound > 0 ? k_bound : 1;
    }
    // small matrices: the elimination runs in registers (osd0_reg_kernel<R, W>), LDS only holds the column order
    void (*reg0)(const OsdArgs) = nullptr;
    if (!higher && h->osd_reg && !h->osd_big) {
        if (a.m <= 64 && a.words <= 2) reg0 = osd0_reg_kernel<1, 2>;
        else if (a.m <= 128 && a.words <= 3) reg0 = osd0_reg_kernel<2, 3>;  // (BB [[144,12,12]]: 72 x 145 bits -- a fourth word would be a quarter more sort and XOR work)
        else if (a.m <= 128 && a.words <= 4) reg0 = osd0_reg_kernel<2, 4>;
        else if (a.m <= 256 && a.words <= 8) reg0 = osd0_reg_kernel<4, 8>;
    }
    // ... and with the columns permuted into their sorted order where the matrix allows (osd0_flat_kernel: m <= 128, n <= 256, rows of up to
    // eight entries): half the instructions per pivot, and a small batch's OSD stage lasts as long as its slowest row
    int flat_r = 0, flat_d = 0;
    if (reg0 && a.m <= 128 && a.n <= 256 && h->max_row_deg <= 8 && !h->on("OSD_NO_FLAT")) {
        static void (*const flat[2][4])(const OsdArgs) = {{osd0_flat_kernel<1, 2>, osd0_flat_kernel<1, 3>, osd0_flat_kernel<1, 5>, osd0_flat_kernel<1, 8>},
                                                          {osd0_flat_kernel<2, 2>, osd0_flat_kernel<2, 3>, osd0_flat_kernel<2, 5>, osd0_flat_kernel<2, 8>}};
        static const int flat_ds[4] = {2, 3, 5, 8};
        const int need = (a.n + 31) / 32;
        int q = 0;
        while (flat_ds[q] < need) ++q;
        flat_r = a.m <= 64 ? 1 : 2;
        flat_d = flat_ds[q];
        reg0 = flat[flat_r - 1][q];
        if (!h->osd_ell.p) {  // a row's entries as eight 16-bit column numbers, once per handle
            std::vector<uint16_t> ell((size_t)a.m * 8, (uint16_t)0xffff);
            for (int i = 0; i < a.m; ++i)
                for (int e = h->h_row_ptr[(size_t)i], k = 0; e < h->h_row_ptr[(size_t)i + 1]; ++e, ++k) ell[(size_t)i * 8 + (size_t)k] = (uint16_t)h->h_col_idx[(size_t)e];
            if ((rc = h->osd_ell.ensure(ell.size() * 2))) return rc;
            HIPCHK(hipMemcpy(h->osd_ell.p, ell.data(), ell.size() * 2, hipMemcpyHostToDevice));
        }
        a.ell = (const uint16_t *)h->osd_ell.p;
    }
    void (*regw)(const OsdArgs) = nullptr;
    if (higher && h->osd_reg && !h->osd_big && a.m <= 256 && a.words <= 8) {
        a.kwords = (osd_k(h) + 63) / 64;
        if (a.kwords < 1) a.kwords = 1;
        if (a.m <= 64 && a.words <= 2) regw = osdw_reg_kernel<1, 2>;
        else if (a.m <= 128 && a.words <= 3) regw = osdw_reg_kernel<2, 3>;
        else if (a.m <= 128 && a.words <= 4) regw = osdw_reg_kernel<2, 4>;
        else regw = osdw_reg_kernel<4, 8>;
    }
    size_t per_wave = flat_r ? osd_flat_lds_bytes(a.n, flat_r, flat_d)
                    : reg0 ? (size_t)a.n * 4
                    : regw ? (size_t)a.n * (8 * ((size_t)a.kwords + 2) + 4 + 4) + 64 * (size_t)a.kwords * 4
                    : higher ? (size_t)a.m * a.words * 8 + (size_t)a.m * 8 + (size_t)a.n * 8 + 3 * (size_t)a.n * 4 + (size_t)a.m * 4
                             : (size_t)a.m * a.words * 8 + (size_t)a.n * 8 + (size_t)a.n * 4 + (size_t)a.m * 4 + (size_t)a.n;
    per_wave = (per_wave + 15) & ~(size_t)15;
    // one workgroup per syndrome (osd_big_kernel: H in LDS if it fits, else in HBM) once the one-wavefront kernels would
    // leave fewer than four wavefronts on a CU; mode 0 keeps the one-wavefront kernels while they fit at all
    const bool big0 = !reg0 && !regw && (per_wave > 150u * 1024u || h->osd_big || (h->osd_reg && per_wave > 40u * 1024u));  // OSD-0 with the matrix in HBM (osd0_big_kernel)
    bool host_rank = false;
    if (reg0 || regw || big0) {  // H bit-packed by rows, once per handle
        // rank H bounds the pivots; working it out is a dense elimination on the host, worth it only for moderate sizes
        host_rank = reg0 || regw || (double)a.m * a.m * a.words < 4e9;
        a.rank = host_rank ? a.n - osd_k(h) : (a.m < a.n ? a.m : a.n);
        if (!h->osd_packed.p) {
            std::vector<uint64_t> packed((size_t)a.m * (size_t)a.words, 0);
            for (int i = 0; i < a.m; ++i)
                for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
                    const int c = h->h_col_idx[(size_t)e];
                    packed[(size_t)i * (size_t)a.words + (size_t)(c >> 6)] |= 1ull << (c & 63);
                }
            if ((rc = h->osd_packed.ensure(packed.size() * 8))) return rc;
            HIPCHK(hipMemcpy(h->osd_packed.p, packed.data(), packed.size() * 8, hipMemcpyHostToDevice));
        }
        a.packed = (const uint64_t *)h->osd_packed.p;
    }
    // wavefronts per workgroup: whichever of 1..4 lets most wavefronts reside on a CU (a workgroup's LDS is one
    // allocation, so large per-wavefront tables pack better in small workgroups); ties go to the larger workgroup
    int waves = 1, resident_best = 0;
    for (int w = 1; w <= 4; ++w) {
        if ((size_t)w * per_wave > 150u * 1024u) break;
        int resident = (int)((160u * 1024u) / ((size_t)w * per_wave)) * w;
        if (resident > 32) resident = 32;
        if (resident >= resident_best) { resident_best = resident; waves = w; }
    }
    a.lds_per_wave = (int32_t)per_wave;
    const size_t dyn = per_wave * (size_t)waves;
    const void *fn = reg0 ? (const void *)reg0 : regw ? (const void *)regw : higher ? (const void *)osdw_kernel : (const void *)osd0_kernel;
    if (!big0 && dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    // list the unconverged rows, then persistent wavefronts (as many as LDS lets reside) pull rows from the list
    a.list = (const int32_t *)h->osd_list.p;
    a.counters = osd_ctr;
    a.status = reg0 ? (uint8_t *)h->osd_status.p : nullptr;
    if (!h->osd_hook.done) {
        HIPCHK(hipMemsetAsync(osd_ctr, 0, 64, h->stream));
        hipLaunchKernelGGL(osd_collect_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, h->stream, conv, batch,
                           (int32_t *)h->osd_list.p, osd_ctr, (uint8_t *)h->osd_status.p);
    }
    h->osd_status_rows = batch;
    // the OSD kernels proper, over the rows of a.list: run once on the caller's syndromes and -- for a rank-deficient H -- once more
    // on the corrected syndromes of the rows that turned out to lie outside the image (osd_exact_kernel.h)
    auto run_osd = [&](OsdArgs a) -> int {
    if (big0) {
        OsdBigArgs A = {};
        A.hwords = (a.n + 63) / 64;
        A.pow2 = 1;
        while (A.pow2 < a.n) A.pow2 <<= 1;
        A.max_rank = a.rank;
        A.kwords = !higher ? 0 : host_rank ? (a.n - a.rank + 63) / 64 : A.hwords;  // planes of T; rank unknown: room for every column
        if (higher && A.kwords < 1) A.kwords = 1;
        // LDS: [pivot columns 2 m, hit list 2 m, syndrome column m + 1] [column order 2 pow2] and then, phase by phase in the SAME room:
        //   sort: keys 8 n;  fill of the working copy: sorted positions 2 n;  elimination: look-ahead words 8 m, combination table;
        //   higher order, once the elimination is over: column info 2 n, plane masks + compress moves 56 hwords, four T planes 32 (m + 1);
        // last [H: hwords planes of m words, if it fits].  (An [[1600,64]] code: 38 KiB, four workgroups per CU.)
        if (a.m > 32767 || a.n > 32767)
            return fail(LDPC_HIP_ERR_UNSUPPORTED, "OSD on the device: %d x %d is beyond the 16-bit row / column tables of the workgroup kernel", a.m, a.n);
        const size_t fixed = ((size_t)a.m * 5 + 1 + 7) & ~(size_t)7;  // bytes before `ord`
        const size_t phase = (fixed + (size_t)A.pow2 * 2 + 15) & ~(size_t)15;
        // blocked elimination (osd_block_eliminate): up to eight rows per thread in registers -> m <= 2048, and the combination table
        // of the block's pivot rows in LDS; LDPC_HIP_OSD_UNBLOCKED=1 keeps the one-pivot-per-step loop (A/B measurements)
        const bool blocked = a.m <= OSD_BLOCK_ROWS && !h->on("OSD_UNBLOCKED");
        const size_t pbuf_bytes = 16 * OSD_PIECE * 16 * 8;  // [group of four pivots][plane of the round][combination]
        size_t room = (size_t)a.n * 8;
        const size_t elim = (size_t)a.m * 8 + (blocked ? pbuf_bytes : 0);
        if (elim > room) room = elim;
        // the staged T planes, one buffer of 8 (m + 1) bytes per wavefront that weighs candidates: four, unless fewer let more
        // workgroups stay resident (tall matrices: at 1728 rows four buffers are 55 KiB and leave ONE workgroup per CU) -- weighing is
        // about a quarter of an OSD row, so halving its wavefronts costs ~ 25 %, a second resident workgroup gains ~ 70 %
        // -- IF there are more rows than resident workgroups; a handful of rows is about latency and wants all four.  How many rows the
        // previous OSD call on this handle listed is the guide (copied back asynchronously, never waited for; first call: an eighth of the batch).
        A.nplanes = 4;
        size_t weigh = 0;
        if (higher) {
            const unsigned seen = h->h_flag ? ((volatile unsigned *)h->h_flag)[8] : 0u;
            const double rows = seen ? (double)seen : (double)batch / 8.0 + 1.0;
            double best = 1e300;
            for (int nb = 4; nb >= 1; nb >>= 1) {
                const size_t wb = (((size_t)a.n * 2 + 7) & ~(size_t)7) + 56 * (size_t)A.hwords + ((size_t)a.m + 1) * 8 * (size_t)nb;
                const size_t tot = phase + (wb > room ? wb : room);
                int pc = (int)((160u * 1024u) / (tot + 1024));
                if (pc > 4) pc = 4;
                if (pc < 1) pc = 1;
                const double cost = std::ceil(rows / (256.0 * pc)) * (1.0 + 0.25 * (4.0 / nb - 1.0));  // rounds of resident workgroups x time of a row
                if (cost < best - 1e-9) { best = cost; A.nplanes = nb; weigh = wb; }
            }
            if (h->sw("OSD_PLANES") > 0) {  // (tests, measurements)
                const int nb = h->sw("OSD_PLANES");
                if (nb == 1 || nb == 2 || nb == 4) { A.nplanes = nb; weigh = (((size_t)a.n * 2 + 7) & ~(size_t)7) + 56 * (size_t)A.hwords + ((size_t)a.m + 1) * 8 * (size_t)nb; }
            }
        }
        if (weigh > room) room = weigh;
        size_t lds = phase + room;
        A.extra_off = (int32_t)phase;
        A.pbuf_off = blocked ? (int32_t)(phase + (size_t)a.m * 8) : -1;
        if (lds > 150u * 1024u)
            return fail(LDPC_HIP_ERR_UNSUPPORTED, "OSD on the device: the column order%s of a %d x %d matrix need%s %zu bytes of LDS, 150 KiB available",
                        higher ? " and the candidate tables" : "", a.m, a.n, higher ? "" : "s", lds);
        lds = (lds + 15) & ~(size_t)15;
        const size_t mat_bytes = (size_t)A.hwords * a.m * 8;
        const bool mat_lds = !h->osd_big && lds + mat_bytes <= 150u * 1024u;
        if (mat_lds) { A.mat_off = (int32_t)lds; lds += mat_bytes; }
        a.lds_per_wave = (int32_t)fixed;
        A.slot_stride = (int64_t)((mat_lds ? 0 : A.hwords) + A.kwords) * a.m;
        if (A.slot_stride < 1) A.slot_stride = 1;
        int per_cu = (int)((160u * 1024u) / (lds + 1024));  // (+ the kernel's static LDS)
        if (per_cu > 4) per_cu = 4;
        if (h->sw("OSD_PER_CU") >= 1 && h->sw("OSD_PER_CU") < per_cu) per_cu = h->sw("OSD_PER_CU");  // (measurements)
        if (per_cu < 1) per_cu = 1;
        int64_t slots = 256 * (int64_t)per_cu;
        if (slots > batch) slots = batch;
        const int64_t cap = (int64_t)(4ull << 30) / (A.slot_stride * 8);  // at most 4 GiB of working copies
        if (slots > cap) slots = cap > 0 ? cap : 1;
        if ((rc = h->osd_scratch.ensure((size_t)slots * (size_t)A.slot_stride * 8))) return rc;
        A.scratch = (uint64_t *)h->osd_scratch.p;
        A.o = a;
        void (*bk)(const OsdBigArgs) = higher ? (mat_lds ? osd_big_kernel<true, true> : osd_big_kernel<true, false>)
                                              : (mat_lds ? osd_big_kernel<false, true> : osd_big_kernel<false, false>);
        if (lds > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)bk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bk, dim3((unsigned)slots), dim3(256), (unsigned)lds, h->stream, A);
        HIPCHK(hipGetLastError());
        return LDPC_HIP_OK;
    }
    int groups_per_cu = (int)((160u * 1024u) / dyn);
    if (groups_per_cu * waves > 32) groups_per_cu = 32 / waves;
    if (groups_per_cu < 1) groups_per_cu = 1;
    int64_t blocks = 256 * (int64_t)groups_per_cu;
    if (blocks > (batch + waves - 1) / waves) blocks = (batch + waves - 1) / waves;
    if (reg0) hipLaunchKernelGGL(reg0, dim3((unsigned)blocks), dim3((unsigned)(waves * 64)), (unsigned)dyn, h->stream, a);
    else if (regw) hipLaunchKernelGGL(regw, dim3((unsigned)blocks), dim3((unsigned)(waves * 64)), (unsigned)dyn, h->stream, a);
    else if (higher) hipLaunchKernelGGL(osdw_kernel, dim3((unsigned)blocks), dim3((unsigned)(waves * 64)), (unsigned)dyn, h->stream, a);
    else hipLaunchKernelGGL(osd0_kernel, dim3((unsigned)blocks), dim3((unsigned)(waves * 64)), (unsigned)dyn, h->stream, a);
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
    };
    if ((rc = run_osd(a))) return rc;
    if (big0 && h->h_flag) HIPCHK(hipMemcpyAsync(&h->h_flag[8], a.counters, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));  // rows listed: the next call's guide
    if (!reg0 && (rc = osd_status_pass(h, a, batch))) return rc;
    // Rows whose syndrome lies outside the image of H (status 2; only a rank-deficient H has any): the reference's answer depends on
    // which rows its linked-list elimination made pivot rows.  One workgroup per such row re-enacts that choice and writes the syndrome
    // that keeps exactly those rows (osd_exact_kernel.h); the same OSD kernels then run once more over these rows.  No host round trip:
    // both launches size themselves from device-side counters and cost a few microseconds when there is nothing to do.
    const bool rank_known = (double)a.m * a.m * a.words < 4e9;
    if (rank_known && a.n - osd_k(h) < a.m && a.m <= 8192 && !h->on("OSD_NO_EXACT")) {
        // Footprint (documented in ldpc_hip.h): at most 64 workgroups' working copies, capped at 256 MiB, plus one corrected syndrome and
        // one list entry per row of the batch.  If the device cannot spare that, the second pass is skipped -- the first-pass solutions
        // stand and the affected rows keep status 2 -- rather than failing a decode whose outputs are already complete.
        const size_t slot_words = osd_exact_slot_words(a.m, a.n);
        int64_t slots = 64;
        if (slots > batch) slots = batch;
        const int64_t cap = (int64_t)((256ull << 20) / (slot_words * 8));
        bool room = cap >= 1;
        if (room) {
            if (slots > cap) slots = cap;
            if (h->osd_fix_synd.ensure(B * (size_t)a.m) || h->osd_fix_list.ensure(B * sizeof(int32_t)) ||
                h->osd_fix_scratch.ensure((size_t)slots * slot_words * 8)) {
                (void)hipGetLastError();  // out of memory: not an error of this decode
                g_last_error.clear();
                room = false;
            }
        }
        if (room) {
            OsdExactArgs X = {};
            X.o = a;
            X.status = (const uint8_t *)h->osd_status.p;
            X.corrected = (uint8_t *)h->osd_fix_synd.p;
            X.list2 = (int32_t *)h->osd_fix_list.p;
            X.counters2 = osd_ctr + 8;
            X.scratch = (uint64_t *)h->osd_fix_scratch.p;
            X.slot_words = (int64_t)slot_words;
            X.hw = (a.n + 63) / 64;
            const size_t xl = osd_exact_lds_bytes(a.m);
            if (xl > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)osd_exact_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)xl));
            hipLaunchKernelGGL(osd_exact_kernel, dim3((unsigned)slots), dim3(256), (unsigned)xl, h->stream, X);
            HIPCHK(hipGetLastError());
            OsdArgs a2 = a;
            a2.synd = (const uint8_t *)h->osd_fix_synd.p;
            a2.list = (const int32_t *)h->osd_fix_list.p;
            a2.counters = osd_ctr + 8;
            a2.status = nullptr;  // (these rows keep their 2)
            if ((rc = run_osd(a2))) return rc;
        }
    }
    return LDPC_HIP_OK;
}
