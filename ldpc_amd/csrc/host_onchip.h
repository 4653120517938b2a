// host_onchip.h -- host side of the on-chip kernels: plans (LDS / occupancy), tables and launches of bp_small / bp_wave / bp_wave_ps / bp_edge
// Part of libldpc_hip.so: included by bp_hip.hip (one translation unit), in the order given there.
#pragma once


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

// bp_wave_kernel: template bounds, launch shape and LDS split; waves == 0: not applicable (degrees, table range, LDS)
// the priors as bp_wave_kernel's LDS copy would hold them: llr0, 1.0 for the padding columns, DBL_MAX at np (a row's phantom entries)
__global__ void wave_prior_pad_kernel(const double *llr0, int n, int np, double *out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < np + 2) out[q] = q < n ? llr0[q] : q == np ? DBL_MAX : 1.0;
}

struct WavePlan {
    int dr = 0, dc = 0, waves = 0, groups_per_cu = 0, mp = 0, np = 0;
    size_t shared = 0, per_wave = 0;
    bool llr_direct = false;
    bool prior_global = false;  // min-sum: the priors are read from a padded device array instead of an LDS copy (WaveArgs.prior_g)
    bool team = false;  // the workgroup's wavefronts share ONE syndrome (bp_wave_kernel<..., TEAM>): `waves` = wavefronts of a team
    void (*kern)(const WaveArgs) = nullptr, (*kern_team)(const WaveArgs) = nullptr;
};

template <int METHOD, int MATH>
static void pick_wave(int max_row, int max_col, WavePlan &p) {
#define LDPC_PICK_WAVE(R, C) { p.dr = R; p.dc = C; p.kern = bp_wave_kernel<METHOD, MATH, R, C, false>; p.kern_team = bp_wave_kernel<METHOD, MATH, R, C, true>; return; }
    if (max_row <= 4 && max_col <= 2) LDPC_PICK_WAVE(4, 2)
    if (max_row <= 4 && max_col <= 4) LDPC_PICK_WAVE(4, 4)
    if (max_row <= 6 && max_col <= 3) LDPC_PICK_WAVE(6, 3)
    if (max_row <= 8 && max_col <= 4) LDPC_PICK_WAVE(8, 4)
    if (max_row <= 8) LDPC_PICK_WAVE(8, 8)
    // rows of up to 16 entries (hamming(5) as a full parity-check matrix): min-sum only -- product-sum takes the lane = entry kernel there
    // (bp_wave_ps_kernel<., 16 | 32, 8>), and its lane = node form would need more registers than a 16-wavefront workgroup has
    if constexpr (METHOD == LDPC_HIP_MINIMUM_SUM) LDPC_PICK_WAVE(16, 8)
#undef LDPC_PICK_WAVE
}

static WavePlan plan_wave(const ldpc_hip_bp *h, bool forced, bool want_llr, int64_t batch) {
    WavePlan p;
    if (h->m <= 0 || h->n <= 0 || h->nnz <= 0 || h->max_row_deg > 16 || h->max_col_deg > 8) return p;
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) pick_wave<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg, p);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) pick_wave<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg, p);
    else pick_wave<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg, p);
    if (!p.kern) return p;
    p.mp = (h->m + 63) / 64 * 64;
    p.np = (h->n + 63) / 64 * 64;
    const size_t rm = (size_t)p.dr * p.mp, cn = (size_t)p.dc * p.np;
    if (rm + 2 >= 65536 || p.np + 1 >= 65536) return p;               // positions and column numbers are 16 bits
    if (!forced && rm > 2 * (size_t)h->nnz + 1024) return p;          // a few heavy rows would pad every row
    p.shared = wave_lds_shared(p.mp, p.np, p.dr, p.dc, h->bp_method == LDPC_HIP_PRODUCT_SUM);
    p.per_wave = wave_lds_private(p.mp, p.np, p.dr, want_llr);
    (void)cn;
    // one workgroup per compute unit with as many wavefronts as LDS (160 KiB) and the 16-wave workgroup limit allow;
    // small codes fit several such workgroups
    const size_t lds = 160u * 1024u - 64u;  // (the kernels' few bytes of static LDS come on top of the dynamic part)
    if (want_llr) {
        // the LDS copy of the log-ratios is a convenience (the store of every iteration stays on chip); where it costs a
        // resident wavefront and few are resident, every bit pass stores them straight to HBM instead
        const size_t lean = wave_lds_private(p.mp, p.np, p.dr, false);
        const size_t w_copy = p.shared + p.per_wave > lds ? 0 : (lds - p.shared) / p.per_wave;
        const size_t w_lean = p.shared + lean > lds ? 0 : (lds - p.shared) / lean;
        if (w_copy < 4 && w_lean > w_copy) { p.per_wave = lean; p.llr_direct = true; }
    }
    if (p.shared + p.per_wave > lds) return p;
    size_t w = (lds - p.shared) / p.per_wave;
    if (w > 16) w = 16;
    // Few resident wavefronts hide little latency, but a wavefront stops when ITS syndrome has converged while a streamed
    // tile runs until its slowest of 64 has.  Measured on 432..864-row window matrices (tools/bench_window.py): min-sum
    // with 3 / 2 / 1 wavefronts per CU is 12x / 7x / 1.5x faster than streaming when most syndromes converge early and
    // 2.2x faster at 3 when most do not; product-sum 3x / 2.4x / 0.7x and about level.
    // Where LDS leaves room for only a few syndromes per CU, one wavefront each leaves the CU idle: the wavefronts of a workgroup
    // then share ONE syndrome (TEAM), as many as its bit pass has rounds of 64 U columns for (small_mode 5 forces, 4 forbids it);
    // and a code whose bit pass takes one wavefront several rounds is quicker that way whatever the room.
    // Measured (round 2, min-sum / product-sum, large batches): 768 x 1600 15.5 -> 4.9 ms / 50 -> 16 ms, 1200 x 2400 21 -> 3.5 ms,
    // surface d = 41 / 31 / 21 25 -> 11 / 24 -> 12.6 / 11.0 -> 10.0 ms, 300 x 600 1.03 -> 0.79 ms; d = 13, 17 (the bit pass of one
    // wavefront is a single round of 64 U columns already) 7 % slower -- hence the second condition.
    const bool ms = h->bp_method == LDPC_HIP_MINIMUM_SUM;
    const int u = ms ? (p.dr <= 4 ? 4 : p.dr <= 8 ? 2 : 1) : (p.dr <= 6 ? 2 : 1);  // the kernel's nodes per lane in flight
    // A batch of no more than one syndrome per wavefront slot is about latency: a team (two wavefronts at least) then too --
    // surface d = 9 .. 17, BB144 at 512 / 4 096 syndromes: 1.25 - 1.6x / 1.0 - 1.3x faster, at 65 536 up to 16 % slower.
    const bool team = h->small_mode == 5 || (h->small_mode != 4 && (w < 6 || 2 * p.np > 3 * 64 * u || batch <= 256 * (int64_t)w));
    if (team) {
        int tw = (p.np + 64 * u - 1) / (64 * u);
        if (h->sw("TEAM_WAVES") >= 1) tw = h->sw("TEAM_WAVES");  // (measurements)
        if (tw < 2) tw = 2;
        if (tw > 16) tw = 16;
        p.team = true;
        p.waves = tw;
        p.kern = p.kern_team;
        if (ms) {  // the LDS copy of the priors, 8 (np + 2) bytes: worth reading them from memory where that fits another workgroup
            const size_t lean_shared = wave_lds_shared(p.mp, p.np, p.dr, p.dc, false, false);
            if (lds / (lean_shared + p.per_wave) > lds / (p.shared + p.per_wave) && !h->on("TEAM_PRIOR_LDS")) { p.shared = lean_shared; p.prior_global = true; }
        }
        // (the kernel's ~100 VGPRs allow 16 wavefronts per CU: two teams of eight beat one of thirteen -- 768 x 1600: 4.0 vs 4.8 ms)
        p.groups_per_cu = (int)(lds / (p.shared + p.per_wave));
        if (p.groups_per_cu >= 2 && p.waves > 8 && h->sw("TEAM_WAVES") < 1) p.waves = 8;
        if (p.groups_per_cu * p.waves > 16) p.groups_per_cu = 16 / p.waves;
        if (p.groups_per_cu < 1) p.groups_per_cu = 1;
        return p;
    }
    if (!forced && w < (h->bp_method == LDPC_HIP_MINIMUM_SUM ? 2 : 3)) return p;
    p.waves = (int)w;
    p.groups_per_cu = (int)(lds / (p.shared + (size_t)p.waves * p.per_wave));
    if (p.groups_per_cu * p.waves > 32) p.groups_per_cu = 32 / p.waves;  // 32 wavefronts per compute unit
    if (p.groups_per_cu < 1) p.groups_per_cu = 1;
    return p;
}

// structure-of-arrays position tables of bp_wave_kernel for the bounds (dr, dc): see bp_wave_kernel.h
static int ensure_wave_tables(ldpc_hip_bp *h, const WavePlan &p) {
    if (h->wave_dr == p.dr && h->wave_dc == p.dc) return LDPC_HIP_OK;
    const int m = h->m, n = h->n, mp = p.mp, np = p.np;
    const size_t rm = (size_t)p.dr * mp, cn = (size_t)p.dc * np;
    std::vector<uint8_t> rdeg((size_t)mp, 0), cdeg((size_t)np, 0);
    std::vector<uint16_t> wcol(rm, (uint16_t)np), wapos(cn, (uint16_t)(rm + 1));  // phantom defaults
    std::vector<int32_t> seen((size_t)n, 0);  // entries of column j met so far = rank of the next one inside the column
    for (int i = 0; i < m; ++i) {
        const int lo = h->h_row_ptr[(size_t)i];
        rdeg[(size_t)i] = (uint8_t)(h->h_row_ptr[(size_t)i + 1] - lo);
        for (int e = lo; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
            const int k = e - lo, j = h->h_col_idx[(size_t)e], kc = seen[(size_t)j]++;  // rows ascend: kc is the CSC order
            wcol[(size_t)k * mp + i] = (uint16_t)j;
            wapos[(size_t)kc * np + j] = (uint16_t)((size_t)k * mp + i);
        }
    }
    for (int j = 0; j < n; ++j) cdeg[(size_t)j] = (uint8_t)seen[(size_t)j];
    int rc;
    if ((rc = h->w_rdeg.ensure(rdeg.size())) || (rc = h->w_cdeg.ensure(cdeg.size())) || (rc = h->w_col.ensure(rm * 2)) ||
        (rc = h->w_apos.ensure(cn * 2))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));  // a previous launch may still read the old tables
    HIPCHK(hipMemcpy(h->w_rdeg.p, rdeg.data(), rdeg.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->w_cdeg.p, cdeg.data(), cdeg.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->w_col.p, wcol.data(), rm * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->w_apos.p, wapos.data(), cn * 2, hipMemcpyHostToDevice));
    h->wave_dr = p.dr;
    h->wave_dc = p.dc;
    return LDPC_HIP_OK;
}

// bp_wave_ps_kernel (product-sum, lane = entry): bounds, launch shape, LDS split; waves == 0: not applicable
struct WavePsPlan {
    int dr = 0, dc = 0, waves = 0, groups_per_cu = 0, np = 0;
    size_t shared = 0, per_wave = 0;
    bool team = false;  // a workgroup per syndrome (bp_wave_ps_kernel<..., TEAM>): `waves` = wavefronts of a team
    void (*kern)(const WavePsArgs) = nullptr, (*kern_team)(const WavePsArgs) = nullptr;
};

template <int MATH>
static void pick_wave_ps(int max_row, int max_col, WavePsPlan &p) {
#define LDPC_PICK_WAVE_PS(R, C) { p.dr = R; p.dc = C; p.kern = bp_wave_ps_kernel<MATH, R, C, false>; p.kern_team = bp_wave_ps_kernel<MATH, R, C, true>; return; }
    if (max_row <= 4 && max_col <= 2) LDPC_PICK_WAVE_PS(4, 2)
    if (max_row <= 4 && max_col <= 4) LDPC_PICK_WAVE_PS(4, 4)
    if (max_row <= 6 && max_col <= 3) LDPC_PICK_WAVE_PS(6, 3)
    if (max_row <= 8 && max_col <= 4) LDPC_PICK_WAVE_PS(8, 4)
    // heavier rows (classical codes given as full parity-check matrices: hamming(5) has rows of 16, hamming(6) of 32): until round 4 these
    // fell to the workgroup ("slot") kernel, ~11 us of barriers per iteration -- a single decode() of hamming(5) took 230 us
    if (max_row <= 16) LDPC_PICK_WAVE_PS(16, 8)
    LDPC_PICK_WAVE_PS(32, 8)
#undef LDPC_PICK_WAVE_PS
}

static WavePsPlan plan_wave_ps(const ldpc_hip_bp *h, bool forced, bool want_llr, int64_t batch) {
    WavePsPlan p;
    if (h->bp_method != LDPC_HIP_PRODUCT_SUM || h->m <= 0 || h->n <= 0 || h->nnz <= 0 || h->max_row_deg > 32 || h->max_col_deg > 8) return p;
    const bool heavy = h->max_row_deg > 8 || h->max_col_deg > 4;  // the (16, 8) / (32, 8) variants
    if (h->math_mode == LDPC_HIP_MATH_FAST) pick_wave_ps<1>(h->max_row_deg, h->max_col_deg, p);
    else pick_wave_ps<0>(h->max_row_deg, h->max_col_deg, p);
    p.np = (h->n + 63) / 64 * 64;
    const size_t rm = (size_t)p.dr * h->m;
    if (rm + 2 >= 65536 || (size_t)p.np + 1 >= 65536) return p;
    // padding would dominate (a heavy code's phantom column entries only cost unrolled LDS reads of the +0.0 slot: judged by its rows alone)
    if (!forced && (rm > 2 * (size_t)h->nnz || (!heavy && (size_t)p.dc * h->n > 2 * (size_t)h->nnz))) return p;
    p.shared = wave_ps_lds_shared(h->m, p.np, p.dr, p.dc);
    p.per_wave = wave_ps_lds_private(h->m, p.np, p.dr, want_llr);
    const size_t lds = 160u * 1024u - 64u;  // (the kernels' few bytes of static LDS come on top of the dynamic part)
    if (p.shared + p.per_wave > lds) return p;
    size_t w = (lds - p.shared) / p.per_wave;
    if (w > 16) w = 16;
    if (!forced && w < 8) return p;
    if (p.dr > 16 && w > 4) w = 4;  // (the kernel's launch bound: bp_wave_ps_kernel<., 32, .>)
    // A batch so small that every wavefront decodes only a few syndromes takes as long as its slowest syndrome: then the
    // workgroup's wavefronts share one (TEAM), one round of 64 entries each per pass.  LDPC_HIP_PS_TEAM=0 / 1 overrides (measurements).
    bool team = batch <= 256 * (int64_t)w * 8;  // (BB144, w = 16: 0.96 -> 0.57 ms at 8 192 syndromes, 1.52 -> 1.37 ms at 32 768, 4.2 -> 4.5 ms at 131 072)
    if (h->sw("PS_TEAM") >= 0) team = h->sw("PS_TEAM") != 0;
    if (team) {
        const size_t rounds = ((size_t)p.dr * h->m + 63) / 64;
        int tw = (int)(rounds < 2 ? 2 : rounds > 8 ? 8 : rounds);
        if (h->sw("PS_TEAM_WAVES") >= 1 && h->sw("PS_TEAM_WAVES") <= 16) tw = h->sw("PS_TEAM_WAVES");  // (measurements)
        if (p.dr > 16 && tw > 4) tw = 4;
        p.team = true;
        p.waves = tw;
        p.kern = p.kern_team;
        p.groups_per_cu = (int)(lds / (p.shared + p.per_wave));
        if (p.groups_per_cu * p.waves > 28) p.groups_per_cu = 28 / p.waves;  // (this kernel's 65 VGPRs allow 7 wavefronts per SIMD)
        if (p.groups_per_cu < 1) p.groups_per_cu = 1;
        return p;
    }
    p.waves = (int)w;
    p.groups_per_cu = (int)(lds / (p.shared + (size_t)p.waves * p.per_wave));
    if (p.groups_per_cu * p.waves > 32) p.groups_per_cu = 32 / p.waves;
    if (p.groups_per_cu < 1) p.groups_per_cu = 1;
    return p;
}

static int ensure_wave_ps_tables(ldpc_hip_bp *h, const WavePsPlan &p) {
    if (h->wave_ps_dr == p.dr && h->wave_ps_dc == p.dc) return LDPC_HIP_OK;
    const int m = h->m, n = h->n, np = p.np;
    const size_t rm = (size_t)p.dr * m, cn = (size_t)p.dc * np;
    std::vector<uint8_t> rdeg((size_t)m, 0);
    std::vector<uint16_t> wcol(rm, (uint16_t)np), wepos(cn, (uint16_t)rm);  // phantom defaults
    std::vector<int32_t> seen((size_t)n, 0);
    for (int i = 0; i < m; ++i) {
        const int lo = h->h_row_ptr[(size_t)i];
        rdeg[(size_t)i] = (uint8_t)(h->h_row_ptr[(size_t)i + 1] - lo);
        for (int e = lo; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
            const int k = e - lo, j = h->h_col_idx[(size_t)e], kc = seen[(size_t)j]++;
            wcol[(size_t)i * p.dr + k] = (uint16_t)j;
            wepos[(size_t)j * p.dc + kc] = (uint16_t)((size_t)i * p.dr + k);
        }
    }
    int rc;
    if ((rc = h->wp_rdeg.ensure(rdeg.size())) || (rc = h->wp_col.ensure(rm * 2)) || (rc = h->wp_epos.ensure(cn * 2))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(h->wp_rdeg.p, rdeg.data(), rdeg.size(), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->wp_col.p, wcol.data(), rm * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->wp_epos.p, wepos.data(), cn * 2, hipMemcpyHostToDevice));
    h->wave_ps_dr = p.dr;
    h->wave_ps_dc = p.dc;
    return LDPC_HIP_OK;
}

static int decode_wave_ps(ldpc_hip_bp *h, const WavePsPlan &p, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                          int32_t *iters, uint8_t *conv) {
    int rc;
    if ((rc = ensure_wave_ps_tables(h, p))) return rc;
    // (a team per syndrome and no more syndromes than resident teams -- e.g. ONE decode(): static assignment, no work counter to reset)
    const bool static_teams = p.team && batch <= 256 * (int64_t)p.groups_per_cu;
    if ((rc = h->counter.ensure(work_pool_bytes()))) return rc;
    // (BP + OSD: the OSD counters sit right behind the work pools -- bposd_device saw to the room -- and share the fill)
    const bool osd_hook = h->osd_hook.armed && h->osd_hook.count == (unsigned *)((char *)h->counter.p + work_pool_bytes());
    if (osd_hook && static_teams) HIPCHK(hipMemsetAsync((char *)h->counter.p + work_pool_bytes(), 0, 64, h->stream));
    if (!static_teams) HIPCHK(hipMemsetAsync(h->counter.p, 0, work_pool_bytes() + (osd_hook ? 64 : 0), h->stream));
    WavePsArgs a = {};
    a.pool_per = work_pool_share(batch, 0);
    a.m = h->m; a.n = h->n; a.np = p.np; a.max_iter = h->max_iter;
    a.batch = batch;
    a.rdeg = (const uint8_t *)h->wp_rdeg.p; a.col = (const uint16_t *)h->wp_col.p; a.epos = (const uint16_t *)h->wp_epos.p;
    a.llr0 = h->d_llr0;
    a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    if (osd_hook) { a.osd_list = h->osd_hook.list; a.osd_count = h->osd_hook.count; a.osd_status = h->osd_hook.status; h->osd_hook.done = true; }
    a.next = static_teams ? nullptr : (unsigned long long *)h->counter.p;
    a.clk = h->d_clk;
    a.lds_shared = (int32_t)p.shared; a.lds_per_wave = (int32_t)p.per_wave;
    a.min_rdeg = h->m;
    for (int i = 0; i < h->m; ++i) a.min_rdeg = std::min(a.min_rdeg, h->h_row_ptr[(size_t)i + 1] - h->h_row_ptr[(size_t)i]);
    const size_t dyn = p.shared + (size_t)(p.team ? 1 : p.waves) * p.per_wave;
    if (dyn > 48u * 1024u)
        HIPCHK(hipFuncSetAttribute((const void *)p.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    int64_t groups = p.team ? batch : (batch + p.waves - 1) / p.waves;
    const int64_t resident = 256 * (int64_t)p.groups_per_cu;
    if (groups > resident) groups = resident;
    h->accumulated_ms = 0.f;
    const bool timed = !(h->untimed_call && static_teams);
    // (the timing events ride on the dispatch itself -- hipExtLaunchKernelGGL attaches them to the kernel's own start and completion -- instead of
    // two hipEventRecord around it: those are a barrier packet each on the stream, 5.8 us of a 0.38 ms step, profiles/r6_c5_step_fusion.txt)
    if (timed) hipExtLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3((unsigned)(p.waves * 64)), (unsigned)dyn, h->stream, h->ev0, h->ev1, 0, a);
    else hipLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3((unsigned)(p.waves * 64)), (unsigned)dyn, h->stream, a);
    h->timed = timed;
    HIPCHK(hipGetLastError());
#ifdef LDPC_WPS_PROF  // measurement build: print and clear the kernel's cycle sums (tools/wave_ps_phases.py)
    {
        HIPCHK(hipStreamSynchronize(h->stream));
        unsigned long long v[12] = {};
        HIPCHK(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_wps_prof), sizeof v));
        fprintf(stderr, "[wps_prof] team %d waves %d groups %lld batch %lld : checkA %llu checkB %llu bitA %llu bitB+synd %llu close %llu setup/out %llu pull %llu iterations %llu syndromes %llu\n",
                (int)p.team, p.waves, (long long)groups, (long long)batch, v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7], v[8]);
        unsigned long long z[12] = {};
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wps_prof), z, sizeof z));
    }
#endif
    return LDPC_HIP_OK;
}

// Wavefront-per-syndrome on-chip variant (bp_wave_kernel).  Device pointers, on h->stream.
static int decode_wave(ldpc_hip_bp *h, const WavePlan &p, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                       int32_t *iters, uint8_t *conv) {
    int rc;
    if ((rc = ensure_wave_tables(h, p))) return rc;
    const bool static_teams = p.team && batch <= 256 * (int64_t)p.groups_per_cu;  // (as decode_wave_ps: no work counter for a handful of syndromes)
    if ((rc = h->counter.ensure(work_pool_bytes()))) return rc;
    // (BP + OSD: the OSD counters sit right behind the work pools -- bposd_device saw to the room -- and share the fill)
    const bool osd_hook = h->osd_hook.armed && h->osd_hook.count == (unsigned *)((char *)h->counter.p + work_pool_bytes());
    if (osd_hook && static_teams) HIPCHK(hipMemsetAsync((char *)h->counter.p + work_pool_bytes(), 0, 64, h->stream));
    if (!static_teams) HIPCHK(hipMemsetAsync(h->counter.p, 0, work_pool_bytes() + (osd_hook ? 64 : 0), h->stream));
    WaveArgs a = {};
    a.pool_per = work_pool_share(batch, 0);
    a.m = h->m; a.n = h->n; a.mp = p.mp; a.np = p.np; a.max_iter = h->max_iter;
    a.ms_scaling_factor = h->ms_scaling_factor;
    a.batch = batch;
    a.rdeg = (const uint8_t *)h->w_rdeg.p; a.cdeg = (const uint8_t *)h->w_cdeg.p;
    a.col = (const uint16_t *)h->w_col.p; a.apos = (const uint16_t *)h->w_apos.p;
    a.llr0 = h->d_llr0;
    if (p.prior_global) {
        if ((rc = h->w_prior.ensure(sizeof(double) * (size_t)(p.np + 2)))) return rc;
        hipLaunchKernelGGL(wave_prior_pad_kernel, dim3((unsigned)((p.np + 2 + 255) / 256)), dim3(256), 0, h->stream, h->d_llr0, h->n, p.np, (double *)h->w_prior.p);
        a.prior_g = (const double *)h->w_prior.p;
    }
    a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    if (osd_hook) { a.osd_list = h->osd_hook.list; a.osd_count = h->osd_hook.count; a.osd_status = h->osd_hook.status; h->osd_hook.done = true; }
    a.llr_direct = p.llr_direct ? 1 : 0;
    a.next = static_teams ? nullptr : (unsigned long long *)h->counter.p;
    a.clk = h->d_clk;
    a.lds_shared = (int32_t)p.shared; a.lds_per_wave = (int32_t)p.per_wave;
    const size_t dyn = p.shared + (size_t)(p.team ? 1 : p.waves) * p.per_wave;
    if (dyn > 48u * 1024u)
        HIPCHK(hipFuncSetAttribute((const void *)p.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    int64_t groups = p.team ? batch : (batch + p.waves - 1) / p.waves;
    const int64_t resident = 256 * (int64_t)p.groups_per_cu;
    if (groups > resident) groups = resident;
    h->accumulated_ms = 0.f;
    const bool timed = !(h->untimed_call && static_teams);
    // (the timing events ride on the dispatch itself -- hipExtLaunchKernelGGL attaches them to the kernel's own start and completion -- instead of
    // two hipEventRecord around it: those are a barrier packet each on the stream, 5.8 us of a 0.38 ms step, profiles/r6_c5_step_fusion.txt)
    if (timed) hipExtLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3((unsigned)(p.waves * 64)), (unsigned)dyn, h->stream, h->ev0, h->ev1, 0, a);
    else hipLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3((unsigned)(p.waves * 64)), (unsigned)dyn, h->stream, a);
    h->timed = timed;
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// ---- bp_edge_kernel (min-sum, lane = edge, messages in registers): rows <= 4, columns 1 .. 2 entries, 4 m <= 1024 slots ----
struct EdgePlan {
    int rounds = 0;  // 0: not applicable
    bool uniform = false;  // every column has the same prior: the form without prior registers (bp_edge_kernel<R, true>)
    void (*kern)(const EdgeArgs) = nullptr;
};

static EdgePlan plan_edge(const ldpc_hip_bp *h) {
    EdgePlan p;
    if (h->bp_method != LDPC_HIP_MINIMUM_SUM || h->m <= 0 || h->n <= 0 || h->nnz <= 0) return p;
    if (h->max_row_deg > 4 || h->max_col_deg > 2 || h->n > 65535) return p;
    const int rounds = (4 * h->m + 63) / 64;
    if (rounds > 16) return p;
    std::vector<char> seen((size_t)h->n, 0);  // a column without entries has no lane to write its outputs
    for (int32_t j : h->h_col_idx) seen[(size_t)j] = 1;
    for (char c : seen) if (!c) return p;
#define LDPC_EDGE_ROW(...) {nullptr, bp_edge_kernel<1, __VA_ARGS__>, bp_edge_kernel<2, __VA_ARGS__>, bp_edge_kernel<3, __VA_ARGS__>, bp_edge_kernel<4, __VA_ARGS__>, \
        bp_edge_kernel<5, __VA_ARGS__>, bp_edge_kernel<6, __VA_ARGS__>, bp_edge_kernel<7, __VA_ARGS__>, bp_edge_kernel<8, __VA_ARGS__>, bp_edge_kernel<9, __VA_ARGS__>, \
        bp_edge_kernel<10, __VA_ARGS__>, bp_edge_kernel<11, __VA_ARGS__>, bp_edge_kernel<12, __VA_ARGS__>, bp_edge_kernel<13, __VA_ARGS__>, bp_edge_kernel<14, __VA_ARGS__>, \
        bp_edge_kernel<15, __VA_ARGS__>, bp_edge_kernel<16, __VA_ARGS__>}
    static void (*const kerns[3][17])(const EdgeArgs) = {LDPC_EDGE_ROW(false), LDPC_EDGE_ROW(true), LDPC_EDGE_ROW(true, true)};
#undef LDPC_EDGE_ROW
    p.uniform = true;
    for (int j = 1; j < h->n && p.uniform; ++j)
        p.uniform = std::memcmp(&h->channel_probs[(size_t)j], &h->channel_probs[0], sizeof(double)) == 0;
    p.rounds = rounds;
    // the form without the clamp to DBL_MAX (bp_edge_kernel.h, NOCLAMP): finite prior, |alpha| <= 1 (0 = the adaptive 1 - 2^-it), rows of two or more
    bool noclamp = p.uniform && std::isfinite(std::log((1 - h->channel_probs[0]) / h->channel_probs[0])) && std::fabs(h->ms_scaling_factor) <= 1.0 && !h->on("EDGE_CLAMP");
    for (int i = 0; i < h->m && noclamp; ++i) noclamp = h->h_row_ptr[(size_t)i + 1] - h->h_row_ptr[(size_t)i] >= 2;
    p.kern = kerns[p.uniform ? (noclamp ? 2 : 1) : 0][rounds];
    return p;
}

__global__ void edge_prior_kernel(const double *llr0, const int32_t *scol, const uint8_t *kind, int slots, double *out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < slots) out[s] = kind[s] ? llr0[scol[s]] : __builtin_inf();  // phantom lanes: +inf (bp_edge_kernel.h)
}

// slot tables of bp_edge_kernel: entry k of row i sits in slot 4 i + k (see bp_edge_kernel.h)
static int ensure_edge_tables(ldpc_hip_bp *h, const EdgePlan &p) {
    if (h->edge_rounds == p.rounds) return LDPC_HIP_OK;
    const int slots = p.rounds * 64;
    std::vector<uint16_t> partner((size_t)slots, (uint16_t)(slots + 1));  // phantom lanes read the slot that holds +inf
    std::vector<uint8_t> kind((size_t)slots, 0);
    std::vector<int32_t> scol((size_t)slots, 0), first((size_t)h->n, -1);
    for (int i = 0; i < h->m; ++i)
        for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
            const int s = 4 * i + (e - h->h_row_ptr[(size_t)i]), j = h->h_col_idx[(size_t)e];
            scol[(size_t)s] = j;
            if (first[(size_t)j] < 0) { first[(size_t)j] = s; kind[(size_t)s] = 1; partner[(size_t)s] = (uint16_t)slots; }  // rows ascend: the column's first entry (bp.hpp:278); alone so far: the +0.0 slot
            else { kind[(size_t)s] = 2; partner[(size_t)s] = (uint16_t)first[(size_t)j]; partner[(size_t)first[(size_t)j]] = (uint16_t)s; }
        }
    int rc;
    if ((rc = h->e_partner.ensure((size_t)slots * 2)) || (rc = h->e_kind.ensure((size_t)slots)) || (rc = h->e_scol.ensure((size_t)slots * 4)) ||
        (rc = h->e_prior.ensure((size_t)slots * 8))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));  // a previous launch may still read the old tables
    HIPCHK(hipMemcpy(h->e_partner.p, partner.data(), (size_t)slots * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->e_kind.p, kind.data(), (size_t)slots, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->e_scol.p, scol.data(), (size_t)slots * 4, hipMemcpyHostToDevice));
    h->edge_rounds = p.rounds;
    return LDPC_HIP_OK;
}

// How the lane = edge kernels share a batch among their wavefronts (work_pool_next, bp_device_common.h): one syndrome each to
// start with, the rest in WORK_POOLS slices behind a work counter each.  Syndromes per visit: 1, except for the smallest codes
// (one round: a syndrome takes a few microseconds) on large batches.  EDGE_STATIC_PCT / EDGE_CHUNK: measurement overrides.
template <typename ARGS>
static int edge_work_split(ldpc_hip_bp *h, int rounds, int64_t batch, int64_t groups, ARGS &a) {
    int rc;
    if ((rc = h->counter.ensure(work_pool_bytes()))) return rc;
    HIPCHK(hipMemsetAsync(h->counter.p, 0, work_pool_bytes(), h->stream));
    a.next = (unsigned long long *)h->counter.p;
    a.clk = h->d_clk;
    int64_t static_per = 1;
    if (h->sw("EDGE_STATIC_PCT") >= 0) static_per = (batch * (h->sw("EDGE_STATIC_PCT") > 100 ? 100 : h->sw("EDGE_STATIC_PCT")) / 100) / groups;
    a.static_per = (int32_t)static_per;
    a.dyn_base = (int32_t)(static_per * groups);
    a.pool_per = work_pool_share(batch, a.dyn_base);
    int64_t c = rounds >= 2 ? 1 : batch / (groups * 16);
    if (h->sw("EDGE_CHUNK") > 0) c = h->sw("EDGE_CHUNK");
    a.chunk = (int32_t)(c < 1 ? 1 : c > 8 ? 8 : c);
    return LDPC_HIP_OK;
}

static int decode_edge(ldpc_hip_bp *h, const EdgePlan &p, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                       int32_t *iters, uint8_t *conv) {
    int rc;
    if ((rc = ensure_edge_tables(h, p))) return rc;
    const int slots = p.rounds * 64;
    // (the priors may have changed since the last call: ldpc_hip_bp_set_channel)
    hipLaunchKernelGGL(edge_prior_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, h->stream, h->d_llr0, (const int32_t *)h->e_scol.p,
                       (const uint8_t *)h->e_kind.p, slots, (double *)h->e_prior.p);
    EdgeArgs a = {};
    a.m = h->m; a.n = h->n; a.max_iter = h->max_iter;
    a.ms_scaling_factor = h->ms_scaling_factor;
    a.batch = batch;
    a.prior_s = (const double *)h->e_prior.p; a.partner = (const uint16_t *)h->e_partner.p;
    a.prior_u = std::log((1 - h->channel_probs[0]) / h->channel_probs[0]);  // as upload_priors (bp.hpp:150-151); read by the uniform form only
    a.kind = (const uint8_t *)h->e_kind.p; a.scol = (const int32_t *)h->e_scol.p;
    a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    const size_t dyn = edge_lds_bytes(p.rounds);
    // one wavefront per workgroup, as many resident as registers (4 or 5 per SIMD) and LDS allow
    int64_t per_cu = (int64_t)((160u * 1024u) / (dyn + 64));
    const int64_t by_regs = p.uniform ? 20 : 16;
    if (per_cu > by_regs) per_cu = by_regs;
    int64_t groups = batch < 256 * per_cu ? batch : 256 * per_cu;
    if ((rc = edge_work_split(h, p.rounds, batch, groups, a))) return rc;
    h->accumulated_ms = 0.f;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3(64), (unsigned)dyn, h->stream, a);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// ---- bp_edge8_kernel (min-sum, lane = edge, rows in 8-lane groups): rows <= 8, columns 1 .. 4 entries, 8 m <= 64 R slots ----
struct Edge8Plan {
    int rounds = 0, dc = 0;  // rounds == 0: not applicable
    bool uniform = false;
    void (*kern)(const Edge8Args) = nullptr;
};

static Edge8Plan plan_edge8(const ldpc_hip_bp *h) {
    Edge8Plan p;
    if (h->bp_method != LDPC_HIP_MINIMUM_SUM || h->m <= 0 || h->n <= 0 || h->nnz <= 0) return p;
    if (h->max_row_deg > 8 || h->max_col_deg > 4 || h->n > 65535) return p;
    const int dc = h->max_col_deg <= 3 ? 3 : 4;
    const int need = (8 * h->m + 63) / 64;
    static const int r3[] = {2, 3, 4, 5, 6, 7, 8, 9, 10, 12}, r4[] = {2, 3, 4, 5, 6, 7, 8, 9};
    int rounds = 0;
    for (int v : r3) if (dc == 3 && !rounds && v >= need) rounds = v;
    for (int v : r4) if (dc == 4 && !rounds && v >= need) rounds = v;
    if (!rounds) return p;
    std::vector<char> seen((size_t)h->n, 0);  // a column without entries has no lane to write its outputs
    for (int32_t j : h->h_col_idx) seen[(size_t)j] = 1;
    for (char c : seen) if (!c) return p;
    p.uniform = true;
    for (int j = 1; j < h->n && p.uniform; ++j)
        p.uniform = std::memcmp(&h->channel_probs[(size_t)j], &h->channel_probs[0], sizeof(double)) == 0;
#define LDPC_E8(R, C) if (rounds == R && dc == C) p.kern = p.uniform ? bp_edge8_kernel<R, C, true> : bp_edge8_kernel<R, C, false>;
    LDPC_E8(2, 3) LDPC_E8(3, 3) LDPC_E8(4, 3) LDPC_E8(5, 3) LDPC_E8(6, 3) LDPC_E8(7, 3) LDPC_E8(8, 3) LDPC_E8(9, 3) LDPC_E8(10, 3) LDPC_E8(12, 3)
    LDPC_E8(2, 4) LDPC_E8(3, 4) LDPC_E8(4, 4) LDPC_E8(5, 4) LDPC_E8(6, 4) LDPC_E8(7, 4) LDPC_E8(8, 4) LDPC_E8(9, 4)
#undef LDPC_E8
    p.rounds = rounds;
    p.dc = dc;
    return p;
}

// slot tables of bp_edge8_kernel: entry k of row i sits in slot 8 i + k (see bp_edge_kernel.h)
static int ensure_edge8_tables(ldpc_hip_bp *h, const Edge8Plan &p) {
    const int key = 1000 + p.rounds * 8 + p.dc;
    if (h->edge_rounds == key) return LDPC_HIP_OK;
    const int slots = p.rounds * 64, dc = p.dc;
    std::vector<uint16_t> cpos((size_t)dc * slots, (uint16_t)(slots + 1));  // phantom lanes read the slot that holds +inf
    std::vector<uint8_t> kind((size_t)slots, 0);
    std::vector<int32_t> scol((size_t)slots, slots + 2);  // phantom lanes: the dummy slot
    std::vector<std::vector<int>> col_slots((size_t)h->n);
    for (int i = 0; i < h->m; ++i)
        for (int e = h->h_row_ptr[(size_t)i]; e < h->h_row_ptr[(size_t)i + 1]; ++e) {
            const int s = 8 * i + (e - h->h_row_ptr[(size_t)i]), j = h->h_col_idx[(size_t)e];
            scol[(size_t)s] = j;
            kind[(size_t)s] = (uint8_t)(1 + col_slots[(size_t)j].size());  // rows ascend: the column's order (bp.hpp:278)
            col_slots[(size_t)j].push_back(s);
        }
    for (int j = 0; j < h->n; ++j)
        for (int s : col_slots[(size_t)j])
            for (int q = 0; q < dc; ++q)
                cpos[(size_t)q * slots + s] = (uint16_t)(q < (int)col_slots[(size_t)j].size() ? col_slots[(size_t)j][(size_t)q] : slots);  // beyond the column: +0.0
    int rc;
    if ((rc = h->e_partner.ensure(cpos.size() * 2)) || (rc = h->e_kind.ensure((size_t)slots)) || (rc = h->e_scol.ensure((size_t)slots * 4)) ||
        (rc = h->e_prior.ensure((size_t)slots * 8))) return rc;
    HIPCHK(hipStreamSynchronize(h->stream));  // a previous launch may still read the old tables
    HIPCHK(hipMemcpy(h->e_partner.p, cpos.data(), cpos.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->e_kind.p, kind.data(), (size_t)slots, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->e_scol.p, scol.data(), (size_t)slots * 4, hipMemcpyHostToDevice));
    h->edge_rounds = key;
    return LDPC_HIP_OK;
}

static int decode_edge8(ldpc_hip_bp *h, const Edge8Plan &p, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                        int32_t *iters, uint8_t *conv) {
    int rc;
    if ((rc = ensure_edge8_tables(h, p))) return rc;
    const int slots = p.rounds * 64;
    hipLaunchKernelGGL(edge_prior_kernel, dim3((unsigned)((slots + 255) / 256)), dim3(256), 0, h->stream, h->d_llr0, (const int32_t *)h->e_scol.p,
                       (const uint8_t *)h->e_kind.p, slots, (double *)h->e_prior.p);
    Edge8Args a = {};
    a.m = h->m; a.n = h->n; a.max_iter = h->max_iter;
    a.ms_scaling_factor = h->ms_scaling_factor;
    a.batch = batch;
    a.prior_s = (const double *)h->e_prior.p; a.cpos = (const uint16_t *)h->e_partner.p;
    a.prior_u = std::log((1 - h->channel_probs[0]) / h->channel_probs[0]);
    a.kind = (const uint8_t *)h->e_kind.p; a.scol = (const int32_t *)h->e_scol.p;
    a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    const size_t dyn = edge_lds_bytes(p.rounds);
    int64_t per_cu = (int64_t)((160u * 1024u) / (dyn + 64));
    const int64_t by_regs = p.uniform ? 20 : 16;
    if (per_cu > by_regs) per_cu = by_regs;
    int64_t groups = batch < 256 * per_cu ? batch : 256 * per_cu;
    if ((rc = edge_work_split(h, p.rounds, batch, groups, a))) return rc;
    h->accumulated_ms = 0.f;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3(64), (unsigned)dyn, h->stream, a);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// Which on-chip kernel (if any) takes this batch: called by decode_device (host_stream.h) before it falls back to the streamed tiles.
int decode_onchip(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv, bool *took) {
    *took = false;
    if (h->small_mode != 0 && h->m > 0 && h->n > 0 && h->nnz > 0 && (int64_t)h->nnz * 16 < (1 << 22) && batch < (1ll << 30)) {  // (32-bit syndrome indices in the work pools)
        // small code: keep the messages on chip.  Bounded degrees: one wavefront per syndrome (bp_wave_kernel).
        // Otherwise the slot kernel -- auto: the most resident syndromes (<= 4) per workgroup that still leave
        // four workgroups per CU (<= 39.5 KiB each); forced: whatever fits in 150 KiB
        if (h->small_mode < 2 || h->small_mode == 6) {  // (mode 6: as -1 wherever the lane = edge variants do not apply) product-sum: one lane per entry keeps the lanes busy with transcendentals
            const WavePsPlan pp = plan_wave_ps(h, h->small_mode == 1, llr != nullptr, batch);
            if (pp.waves) { *took = true; return decode_wave_ps(h, pp, synd, batch, decoding, llr, iters, conv); }
        }
        if (h->small_mode == -1 || h->small_mode == 1 || h->small_mode == 6) {  // min-sum on the surface-code family: lane = edge
            const EdgePlan ep = plan_edge(h);
            if (ep.rounds) { *took = true; return decode_edge(h, ep, synd, batch, decoding, llr, iters, conv); }
            const Edge8Plan e8 = plan_edge8(h);  // heavier nodes (rows <= 8, columns <= 4): rows in 8-lane groups
            if (e8.rounds) { *took = true; return decode_edge8(h, e8, synd, batch, decoding, llr, iters, conv); }
        }
        if (h->small_mode != 2) {
            const WavePlan wp = plan_wave(h, h->small_mode == 1 || (h->small_mode >= 3 && h->small_mode != 6), llr != nullptr, batch);
            if (wp.waves) { *took = true; return decode_wave(h, wp, synd, batch, decoding, llr, iters, conv); }
        }
        int slots = 0;
        const size_t budget = (h->small_mode == 1 || h->small_mode == 2) ? 150u * 1024u : 39u * 1024u + 512u;
        for (int sl = 4; sl >= 1 && !slots; --sl)
            if (small_lds_bytes(h, sl) <= budget) slots = sl;
        if (slots) { *took = true; return decode_small(h, synd, batch, decoding, llr, iters, conv, slots); }
    }
    return LDPC_HIP_OK;
}

// ---- a single decode() through the resident workgroup ------------------------------------------------------------------------------------
// What the reference's callers do is `for shot: decoder.decode(shot)`: one syndrome per call, host arrays.  A launch and its completion
// cost ~15 us of such a call and the kernel's table set-up ~10 us more (profiles/r4_single_decode_latency.txt) -- so the workgroup that
// decoded one syndrome STAYS (bp_wave_ps_kernel<., ., ., TEAM>, WavePsArgs::mail): the next call writes its syndrome into the handle's
// host-mapped block, bumps the request word, and spins on the served word; the workgroup polls the request word over PCIe, decodes with
// the tables it already has in LDS, writes the results into the block and bumps the served word.  It leaves after `linger` (100 us,
// LDPC_HIP_RESIDENT_LINGER_US) without a request -- it cannot outlive a caller by more than that, and a torch.cuda.synchronize() behind a
// decode() waits that long at most -- or at once when told to (resident_retire: parameters or priors changed, handle destroyed).
// The block's layout for one syndrome is decode_batch_staged's (host_decode_abi.h); the syndrome is already in it.
void resident_retire(ldpc_hip_bp *h) {
    auto &r = h->res;
    if (!h->pin_host || !r.stream) return;
    volatile unsigned *mail = reinterpret_cast<volatile unsigned *>(h->pin_host + ldpc_hip_bp::PIN_MAIL);
    if (r.launched) {
        __atomic_store_n(const_cast<unsigned *>(mail + 3), 1u, __ATOMIC_SEQ_CST);
        (void)hipStreamSynchronize(r.stream);
        __atomic_store_n(const_cast<unsigned *>(mail + 3), 0u, __ATOMIC_SEQ_CST);
        r.launched = false;
    }
}

int decode_onchip_resident(ldpc_hip_bp *h, bool want_llr, bool *took) {
    *took = false;
    (void)want_llr;
    // (the same predicate as decode_onchip's bp_wave_ps branch: a caller who FORCED another small-code kernel -- modes 3, 4, 5, 2 -- gets that kernel)
    if (h->sw("RESIDENT") == 0 || !h->pin_host || h->schedule != 1 || !(h->small_mode < 2 || h->small_mode == 6) || h->small_mode == 0) return LDPC_HIP_OK;
    if (h->m <= 0 || h->n <= 0 || h->nnz <= 0 || (int64_t)h->nnz * 16 >= (1 << 22)) return LDPC_HIP_OK;
    const WavePsPlan p = plan_wave_ps(h, h->small_mode == 1, true, 1);  // (the posteriors are always formed: one plan whatever the caller asks for)
    if (!p.waves || !p.team) return LDPC_HIP_OK;
    auto &r = h->res;
    int rc;
    if (!r.stream) {
        if (hipStreamCreateWithFlags(&r.stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&r.ended, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (r.stream) { (void)hipStreamDestroy(r.stream); r.stream = nullptr; }
            return LDPC_HIP_OK;
        }
    }
    unsigned *mail = reinterpret_cast<unsigned *>(h->pin_host + ldpc_hip_bp::PIN_MAIL);
    unsigned long long alpha_bits;
    std::memcpy(&alpha_bits, &h->ms_scaling_factor, 8);
    const unsigned long long key[6] = {((unsigned long long)(unsigned)h->max_iter << 32) | (unsigned)h->math_mode, h->priors_version, alpha_bits,
                                       ((unsigned long long)(unsigned)p.dr << 48) | ((unsigned long long)(unsigned)p.dc << 32) | (unsigned)p.waves,
                                       (unsigned long long)(uintptr_t)h->d_llr0, (unsigned long long)(unsigned)h->small_mode};
    if (r.launched && std::memcmp(key, r.key, sizeof key) != 0) resident_retire(h);  // (launched for other parameters: it leaves, a new one comes)
    const size_t m = (size_t)h->m, n = (size_t)h->n;
    auto up16 = [](size_t v) { return (v + 15) & ~(size_t)15; };
    const size_t o_dec = up16(m), o_llr = o_dec + up16(n), o_it = o_llr + up16(n * 8), o_cv = o_it + up16(4);
    auto launch = [&]() -> int {
        if ((rc = ensure_wave_ps_tables(h, p))) return rc;
        WavePsArgs a = {};
        a.m = h->m; a.n = h->n; a.np = p.np; a.max_iter = h->max_iter;
        a.batch = 1;
        a.rdeg = (const uint8_t *)h->wp_rdeg.p; a.col = (const uint16_t *)h->wp_col.p; a.epos = (const uint16_t *)h->wp_epos.p;
        a.llr0 = h->d_llr0;
        unsigned char *dv = h->pin_dev;
        a.synd = dv; a.decoding = dv + o_dec; a.llr = (double *)(dv + o_llr); a.iters = (int32_t *)(dv + o_it); a.conv = dv + o_cv;
        a.next = nullptr;
        a.clk = nullptr;
        a.lds_shared = (int32_t)p.shared; a.lds_per_wave = (int32_t)p.per_wave;
        a.min_rdeg = h->m;
        for (int i = 0; i < h->m; ++i) a.min_rdeg = std::min(a.min_rdeg, h->h_row_ptr[(size_t)i + 1] - h->h_row_ptr[(size_t)i]);
        a.mail = reinterpret_cast<unsigned *>(h->pin_dev + ldpc_hip_bp::PIN_MAIL);
        a.served0 = __atomic_load_n(mail + 1, __ATOMIC_ACQUIRE);
        const int linger_us = h->sw("RESIDENT_LINGER_US") > 0 ? h->sw("RESIDENT_LINGER_US") : 100;
        a.linger_ticks = (unsigned)linger_us * 100u;  // (the constant-rate counter runs at 100 MHz)
        const size_t dyn = p.shared + p.per_wave;
        if (dyn > 48u * 1024u) HIPCHK(hipFuncSetAttribute((const void *)p.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
        __atomic_store_n(mail + 3, 0u, __ATOMIC_SEQ_CST);
        __atomic_store_n(mail + 2, 1u, __ATOMIC_SEQ_CST);
        hipLaunchKernelGGL(p.kern, dim3(1), dim3((unsigned)(p.waves * 64)), (unsigned)dyn, r.stream, a);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(r.ended, r.stream));
        std::memcpy(r.key, key, sizeof key);
        r.launched = true;
        return LDPC_HIP_OK;
    };
    if (r.launched && __atomic_load_n(mail + 2, __ATOMIC_ACQUIRE) == 0u) {  // it said it was leaving: see it gone before the next one is launched
        HIPCHK(hipStreamSynchronize(r.stream));
        r.launched = false;
    }
    if (!r.launched) {
        if (r.seq == 0) { __atomic_store_n(mail + 0, 0u, __ATOMIC_SEQ_CST); __atomic_store_n(mail + 1, 0u, __ATOMIC_SEQ_CST); }
        if ((rc = launch())) return rc;
    }
    const unsigned seq = ++r.seq ? r.seq : ++r.seq;  // (never 0: the words' initial value)
    __atomic_store_n(mail + 0, seq, __ATOMIC_SEQ_CST);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(mail + 1, __ATOMIC_ACQUIRE) == seq) break;
        if (__atomic_load_n(mail + 2, __ATOMIC_SEQ_CST) == 0u) {
            // it is leaving (or gone): either it saw this request on its last look and serves it before it goes, or it did not
            const hipError_t q = hipEventQuery(r.ended);
            if (q == hipSuccess) {
                r.launched = false;
                if (__atomic_load_n(mail + 1, __ATOMIC_ACQUIRE) == seq) break;
                if ((rc = launch())) return rc;  // (the new one finds the request waiting)
            } else if (q != hipErrorNotReady) {
                return fail(LDPC_HIP_ERR_DEVICE, "the resident decode kernel failed: %s", hipGetErrorString(q));
            } else {
                (void)hipGetLastError();
            }
        }
        if ((spins & 0xfffu) == 0xfffu && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(20)) {
            // The resident workgroup was never scheduled (the GPU is busy with someone else's long kernel) or is stuck: tell it to leave, wait
            // for its stream, and if it did serve the request on its way out take that; otherwise withdraw the request (the next resident
            // kernel starts from served = request: no stale sequence number) and let the ordinary launch path decode this call.
            resident_retire(h);
            if (__atomic_load_n(mail + 1, __ATOMIC_ACQUIRE) == seq) break;
            __atomic_store_n(mail + 0, 0u, __ATOMIC_SEQ_CST);
            __atomic_store_n(mail + 1, 0u, __ATOMIC_SEQ_CST);
            r.seq = 0;
            return LDPC_HIP_OK;  // (*took stays false)
        }
        if ((spins & 0x3ffu) == 0x3ffu) std::this_thread::yield();  // (a core shared with the caller's other threads is not held hostage)
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    h->timed = false;
    h->accumulated_ms = 0.f;
    *took = true;
    return LDPC_HIP_OK;
}
