// bp_hip.hip -- libldpc_hip.so: batched flooding belief propagation for gfx950 (MI355X, CDNA4).
//
// What it replaces (reference = quantumgizmos/ldpc @ /root/reference):
//   ldpc::bp::BpDecoder::bp_decode_parallel          src_cpp/bp.hpp:192-325
//   ldpc::bp::BpDecoder::initialise_log_domain_bp    src_cpp/bp.hpp:147-157
//   ldpc::gf2sparse::GF2Sparse::mulvec               src_cpp/gf2sparse.hpp:177-214
// for a BATCH of independent syndromes.  This is not a translation of that code: the reference walks
// a doubly linked list per syndrome on one CPU thread; here
//
//   * the unit of data parallelism is the syndrome.  64 syndromes form a TILE; lane l of every
//     wavefront owns syndrome l of its tile, so each per-edge message access of a wavefront is one
//     fully coalesced 512-byte row  msg[tile][edge][0..63]  (batch-minor layout),
//   * one workgroup owns one tile for the WHOLE decode (all iterations): its wavefronts stride over
//     the checks (check pass) and then over the bits (bit pass) of that tile, separated by
//     workgroup barriers only -- tiles never talk to each other, so there is no grid-wide sync,
//     no atomics and one kernel launch per decode,
//   * every lane walks its node's <= DR (row) / <= DC (column) edges SEQUENTIALLY in ascending
//     column / row order, i.e. in the reference's linked-list order (sparse_matrix_base.hpp:423-482
//     keeps both lists sorted), so every floating-point operation is performed in the reference's
//     association order -- min-sum is bit-identical, and so is product-sum with the libm-exact routines of bp_math.h,
//   * hard decisions are kept bit-packed per (tile, bit) as the wavefront's ballot; the syndrome
//     test of bp.hpp:300-302 is an XOR-gather of those 64-bit words per check,
//   * a syndrome that converges freezes its outputs (bp.hpp:300-308 early return); a tile whose 64
//     syndromes are all done retires its workgroup.
//
// No MFMA: the path is a sparse gather/scatter bound by HBM bandwidth (4 * nnz * 8 bytes per
// syndrome-iteration) and by FP64 transcendentals.
//
// Built with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared
//
// One translation unit.  Device code lives in the headers included below, this file holds the handle and the C ABI:
//   bp_device_common.h   argument blocks, buffer-descriptor message addressing, per-node arithmetic, LDS-DMA helpers
//   bp_math.h            tanh / log / division: bit-identical twins of the host libm + the fast variants
//   bp_stream_kernel.h   bp_decode_kernel        persistent workgroup per 64-syndrome tile (register / LDS-ring variants)
//   bp_spread_kernels.h  bp_spread_*_kernel      one launch per pass, a tile spread over the chip (small batches, stragglers)
//   bp_small_kernel.h    bp_small_kernel         messages resident in LDS (surface / bivariate-bicycle sized codes), slots per workgroup
//   bp_wave_kernel.h     bp_wave_kernel, bp_wave_ps_kernel   same regime, bounded degrees: one wavefront per syndrome, no workgroup barriers
//   bp_edge_kernel.h     bp_edge_kernel                      min-sum, rows <= 4 / columns <= 2 (surface-code family): lane = edge, messages in registers
//                        (lane = node; for product-sum lane = entry)
//   bp_serial_kernels.h  bp_serial_kernel, bp_softinfo_kernel   serial schedule, soft-syndrome serial min-sum
//   osd_kernels.h        osd0[_reg]_kernel, osdw[_reg]_kernel, osd_big_kernel   OSD-0 / OSD-E / OSD-CS post-processing
//   io_kernels.h         pack / unpack / transpose, H v, b8 shot data, synthetic BSC shots
//   multi_device.h       ldpc_hip_bp_multi_*: a batch sharded over several GPUs inside one process (host code only)

#include "bp_device_common.h"
#include "bp_stream_kernel.h"
#include "bp_spread_kernels.h"
#include "bp_serial_kernels.h"
#include "bp_relative_kernel.h"
#include "bp_small_kernel.h"
#include "bp_wave_kernel.h"
#include "bp_edge_kernel.h"
#include "osd_kernels.h"
#include "osd_exact_kernel.h"
#include "io_kernels.h"

#include <chrono>
#include <random>

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

static thread_local std::string g_last_error;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return fail(_e == hipErrorOutOfMemory ? LDPC_HIP_ERR_NOMEM : LDPC_HIP_ERR_DEVICE, \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,      \
                        __LINE__);                                                            \
    } while (0)

struct DeviceBuf {  // grow-only device allocation
    void *p = nullptr;
    size_t cap = 0;
    int ensure(size_t bytes) {
        if (bytes <= cap) return 0;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess) {
            p = nullptr;
            return fail(LDPC_HIP_ERR_NOMEM, "hipMalloc(%zu bytes) failed: %s", bytes,
                        hipGetErrorString(e));
        }
        cap = bytes;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

// Measurement / test switches (none changes a result; profiles/README.md lists them).  They live in the handle: seeded ONCE, at
// creation, from the environment variables LDPC_HIP_<NAME>, changed afterwards only through ldpc_hip_bp_set_debug_switch -- no
// getenv on the decode path, and nothing a test can change under a live handle by accident.
static const char *const k_switch_names[] = {"TEAM_WAVES", "TEAM_PRIOR_LDS", "PS_TEAM", "EXPLICIT_INIT", "DEBUG_HANDOFF", "REPACK_RESTART",
                                             "OSD_UNBLOCKED", "OSD_PLANES", "OSD_PER_CU", "OSD_NO_EXACT", "NO_PINNED_PATH", "KEEP_LAST_MESSAGES"};
constexpr int k_n_switches = (int)(sizeof(k_switch_names) / sizeof(k_switch_names[0]));

struct ldpc_hip_bp {
    int32_t switches[k_n_switches];  // -1 = not set
    int sw(const char *name) const {  // value of a switch, -1 when it is not set
        for (int i = 0; i < k_n_switches; ++i)
            if (!std::strcmp(name, k_switch_names[i])) return switches[i];
        return -1;
    }
    bool on(const char *name) const { return sw(name) > 0; }
    int device = 0;
    int32_t m = 0, n = 0, nnz = 0;
    int32_t max_iter = 1, bp_method = 0;
    double ms_scaling_factor = 1.0;
    int32_t max_row_deg = 0, max_col_deg = 0;
    int32_t waves_per_wg = 0;  // 0 = auto
    int32_t math_mode = LDPC_HIP_MATH_LIBM_EXACT;
    bool regular = false;   // every row has the same weight and every column has the same weight
    int32_t ring_depth = 2; // LDS-DMA ring slots per wavefront for regular matrices (0 = register variant)
    int32_t small_mode = -1; // on-chip kernels for small codes: -1 auto, 0 never, 1 whenever one fits, 2 the slot kernel only
    std::vector<int32_t> h_row_ptr, h_col_idx;  // host copy of the CSR arrays
    int wave_dr = 0, wave_dc = 0;  // template bounds the uploaded SoA position tables of bp_wave_kernel were built for (0: none)
    int wave_ps_dr = 0, wave_ps_dc = 0;  // likewise for bp_wave_ps_kernel
    DeviceBuf wp_rdeg, wp_col, wp_epos;
    DeviceBuf w_rdeg, w_cdeg, w_col, w_apos, w_prior;
    DeviceBuf d_edge0;       // [n] initial edge values of the streamed kernel (BpArgs::edge0)
    // continuation of a first pass (decode_stream_repacked): decode_device takes its message state from here and counts on from cont_it_start
    double *cont_A = nullptr;
    int32_t cont_it_start = 0;
    bool keep_state = false;       // this decode_device call is a first pass: its last bit pass must leave the messages behind
    int64_t last_chunk_tiles = 0;  // tiles per chunk of the last streamed decode (== its tile count: the whole batch's state is resident)
    DeviceBuf rp_msg;
    int edge_rounds = 0;     // rounds the uploaded slot tables of bp_edge_kernel were built for (0: none)
    DeviceBuf e_partner, e_kind, e_scol, e_prior;
    int32_t handoff = -1;    // straggler hand-off threshold in tiles: -1 auto (256), 0 off
    DeviceBuf tile_state, handoff_list;
    unsigned *h_counters = nullptr;  // pinned host copy of the device counters
    // Per-pass rounds are queued without waiting for the device.  The kernel that finalises the last running tile writes
    // the decode's sequence number into this host-mapped word; the host merely LOOKS at it before queueing the next round
    // (no synchronisation) and stops queueing once it matches -- rounds queued past that point find nothing to do.
    unsigned *h_flag = nullptr, *d_flag = nullptr;
    unsigned flag_seq = 0;
    int32_t schedule = 1;    // ldpc::bp::BpSchedule (bp.hpp:28-32): 0 serial, 1 parallel, 2 serial_relative
    // What the reference keeps in the decoder OBJECT from one decode to the next (bp.hpp:67, 75): serial_schedule_order -- the
    // arrangement serial_relative re-sorts and the random schedule re-shuffles every iteration -- and the generator of the shuffles.
    std::vector<int32_t> sched_state;
    std::mt19937 sched_rng;
    int32_t sched_seed_raw = 0;  // random_schedule_seed as given (the soft-syndrome routine seeds its own engine with it)
    bool random_serial = false;
    DeviceBuf rel_ord, rel_dbit, sched_orders, sched_order0;
    int32_t *d_csc_row = nullptr, *d_order = nullptr;
    bool custom_order = false;
    DeviceBuf counter;
    std::vector<double> channel_probs;

    int32_t *d_row_ptr = nullptr, *d_col_idx = nullptr, *d_col_ptr = nullptr, *d_csc_edge = nullptr;
    double *d_llr0 = nullptr;
    double *d_osd_wt = nullptr;  // [n] log(1 / p_j), the candidate weights of higher-order OSD
    bool osd_reg = true;  // register-resident elimination for small matrices (ldpc_hip_bp_set_osd_kernel)
    bool osd_big = false; // OSD-0 through osd0_big_kernel whatever the size (testing)
    int osd_k_cached = -1;  // n - rank(H), computed on first use
    int32_t osd_method = 1, osd_order = 0;  // ldpc::osd::OsdMethod (osd.hpp:18-23) used by ldpc_hip_bposd_decode_batch

    hipStream_t own_stream = nullptr, stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_mid = nullptr;  // ev_mid: end of the persistent kernel, when one ran
    hipEvent_t ev_done = nullptr;  // end of the last call that queued work on `stream` (orders a change of stream after it)
    bool work_queued = false;
    bool timed = false, timed_mid = false;
    float accumulated_ms = 0.f, accumulated_persistent_ms = 0.f;

    DeviceBuf msgA, msgC, par, nzm, invalid, dec, dcur, llr_t;       // workspace
    DeviceBuf st_synd, st_dec, st_llr, st_iters, st_conv, st_misc;  // staging for host pointers
    // small calls with host buffers (a single decode()): one host-mapped, coherent block that the kernels read and write in place --
    // no copy commands at all, one launch sequence and one wait
    unsigned char *pin_host = nullptr, *pin_dev = nullptr;
    static constexpr size_t PIN_BYTES = 512u * 1024u;
    DeviceBuf osd_llr, osd_conv;                                    // BP outputs OSD-0 needs when the caller does not ask for them
    DeviceBuf osd_scratch;                                          // working copies of H for osd0_big_kernel
    DeviceBuf osd_packed;                                           // [m][words] H bit-packed by rows (register OSD kernels)
    DeviceBuf osd_list, osd_counters;                               // rows BP left unconverged + {count, next}
    DeviceBuf osd_status;                                           // [batch] of the last BP + OSD decode: 0 BP converged, 1 OSD solved, 2 s outside image(H)
    DeviceBuf osd_fix_synd, osd_fix_list, osd_fix_counters, osd_fix_scratch;  // second OSD pass over the rows outside the image (osd_exact_kernel.h)
    int64_t osd_status_rows = 0;
    DeviceBuf rp_synd, rp_dec, rp_llr, rp_iters, rp_conv;           // repacked second pass of the serial schedule
    int32_t serial_kernel = -1;                                     // -1 auto, 0 one wavefront per tile, 1 level-parallel workgroup per tile
    bool order_visits_all = true;                                   // false: some bit is never updated (its outputs stay 0)
    bool levels_valid = false;                                      // lvl_* describe the current schedule order
    int32_t n_levels = 0;
    DeviceBuf lvl_ptr, lvl_bits;
    // repacking of the streamed parallel schedule (decode_stream_repacked), steered by what the previous decode looked like
    DeviceBuf sp_hist, sp_iters;     // iteration histogram of the last streamed decode (256 bins) / iteration counts when the caller wants none
    unsigned *h_hist = nullptr;      // pinned copy of the histogram
    hipEvent_t ev_hist = nullptr;    // the copy has landed
    bool hist_pending = false;
    int32_t hist_max_iter = 0;
    int32_t repack_iters = -1;                                      // first-pass iterations: -1 auto (max_iter / 8), 0 = no repacking
    DeviceBuf soft_S, soft_in, soft_out;                             // soft-syndrome decoding: scaled analog syndromes, staging
    DeviceBuf b8_in, b8_out, b8_synd, b8_dec, obs_row_ptr, obs_col_idx;  // bit-packed shot I/O and the observables matrix
    int32_t obs_k = -1;                                              // rows of the observables matrix (-1: not set)
    int64_t max_chunk_tiles = 0;                                     // 0 = decide from free memory
};

static int upload_priors(ldpc_hip_bp *h) {
    // bp.hpp:150-151, evaluated by the host libm so that priors are bit-identical to the reference's
    std::vector<double> llr0((size_t)h->n);
    for (int j = 0; j < h->n; ++j)
        llr0[(size_t)j] = std::log((1 - h->channel_probs[(size_t)j]) / h->channel_probs[(size_t)j]);
    HIPCHK(hipMemcpy(h->d_llr0, llr0.data(), sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice));
    for (int j = 0; j < h->n; ++j) llr0[(size_t)j] = std::log(1 / h->channel_probs[(size_t)j]);  // osd.hpp:134
    HIPCHK(hipMemcpy(h->d_osd_wt, llr0.data(), sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice));
    return 0;
}

// grid of the one-dimensional element-wise kernels (io_kernels.h): they run grid-stride loops, so the grid is capped --
// item counts like batch * n exceed what one launch dimension can carry for large batches of large codes
static dim3 flat_grid(size_t items) {
    size_t blocks = (items + 255) / 256;
    if (blocks > (1u << 22)) blocks = 1u << 22;
    if (blocks < 1) blocks = 1;
    return dim3((unsigned)blocks);
}

// end of a call that only queued work: remembered so that a later change of stream is ordered after it (set_stream)
static int mark_queued(ldpc_hip_bp *h, int rc) {
    if (rc) return rc;
    HIPCHK(hipEventRecord(h->ev_done, h->stream));
    h->work_queued = true;
    return LDPC_HIP_OK;
}

static bool is_device_ptr(const void *p) {
    if (!p) return true;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // unregistered host memory reports an error: clear it
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

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
    if (e == hipSuccess) e = hipEventCreate(&h->ev_hist);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipHostMalloc((void **)&h->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) { std::memset(h->h_flag, 0, 64); e = hipHostGetDevicePointer((void **)&h->d_flag, h->h_flag, 0); }
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
    (void)hipStreamSynchronize(h->stream);
    for (DeviceBuf *b : {&h->msgA, &h->msgC, &h->par, &h->nzm, &h->invalid, &h->dec, &h->dcur, &h->llr_t,
                         &h->st_synd, &h->st_dec, &h->st_llr, &h->st_iters, &h->st_conv, &h->st_misc, &h->osd_llr, &h->osd_conv, &h->osd_packed, &h->osd_scratch, &h->sp_hist, &h->sp_iters, &h->osd_list, &h->osd_counters, &h->osd_status, &h->osd_fix_synd, &h->osd_fix_list, &h->osd_fix_counters, &h->osd_fix_scratch, &h->rel_ord, &h->rel_dbit, &h->sched_orders, &h->sched_order0, &h->lvl_ptr, &h->lvl_bits, &h->rp_synd, &h->rp_dec, &h->rp_llr, &h->rp_iters, &h->rp_conv, &h->counter, &h->w_rdeg, &h->w_cdeg, &h->w_col, &h->w_apos, &h->w_prior, &h->d_edge0, &h->rp_msg, &h->e_partner, &h->e_kind, &h->e_scol, &h->e_prior, &h->wp_rdeg, &h->wp_col, &h->wp_epos,
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
    if (h->ev_hist) (void)hipEventDestroy(h->ev_hist);
    if (h->ev_done) (void)hipEventDestroy(h->ev_done);
    if (h->h_flag) (void)hipHostFree(h->h_flag);
    if (h->pin_host) (void)hipHostFree(h->pin_host);
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
    if (mode < -1 || mode > 1) return fail(LDPC_HIP_ERR_INVALID, "mode must be -1 (automatic), 0 (one wavefront per tile) or 1 (level-parallel)");
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
    *persistent_ms = pers;
    *per_pass_ms = total - pers;
    return LDPC_HIP_OK;
}

}  // extern "C"

typedef void (*bp_kernel_t)(const BpArgs);
typedef void (*spread_kernel_t)(const SpreadArgs);

template <int METHOD, int MATH>
static void pick_spread_m(int max_row, int max_col, bool nt, spread_kernel_t &kc, spread_kernel_t &kb) {
    if (nt) {
        kc = max_row <= 8 ? bp_spread_check_kernel<METHOD, MATH, 8, 1> : bp_spread_check_kernel<METHOD, MATH, 16, 1>;
        kb = max_col <= 4 ? bp_spread_bit_kernel<METHOD, MATH, 4, 1> : bp_spread_bit_kernel<METHOD, MATH, 8, 1>;
    } else {
        kc = max_row <= 8 ? bp_spread_check_kernel<METHOD, MATH, 8, 0> : bp_spread_check_kernel<METHOD, MATH, 16, 0>;
        kb = max_col <= 4 ? bp_spread_bit_kernel<METHOD, MATH, 4, 0> : bp_spread_bit_kernel<METHOD, MATH, 8, 0>;
    }
}

struct KernelChoice {
    bp_kernel_t fn;
    int ring_slot_bytes;  // 0: register-prefetch variant, no dynamic LDS
    int ring_depth;
};

template <int METHOD, int MATH>
static KernelChoice pick_kernel(int max_row, int max_col, int ring_depth) {
    // Register arrays are sized by the template bounds, so the common regular codes get exact fits:
    // (3,6)-LDPC / bivariate-bicycle rows of 6 and columns of 3 use the LDS-DMA ring variant.
    if (ring_depth == 2 && max_row == 6 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 2>, 3 * 1024, 2};
    if (ring_depth >= 3 && max_row == 6 && max_col == 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 3>, 3 * 1024, 3};
    if (ring_depth >= 2 && max_row == 8 && max_col == 4) return {bp_decode_kernel<METHOD, MATH, 8, 4, 3>, 4 * 1024, 3};
    if (max_row <= 4 && max_col <= 3) return {bp_decode_kernel<METHOD, MATH, 4, 3, 0>, 0, 0};
    if (max_row <= 6 && max_col <= 3) return {bp_decode_kernel<METHOD, MATH, 6, 3, 0>, 0, 0};
    if (max_row <= 8 && max_col <= 4) return {bp_decode_kernel<METHOD, MATH, 8, 4, 0>, 0, 0};
    if (max_row <= 8 && max_col <= 8) return {bp_decode_kernel<METHOD, MATH, 8, 8, 0>, 0, 0};
    if (max_col <= 8) return {bp_decode_kernel<METHOD, MATH, 16, 8, 0>, 0, 0};
    return {bp_decode_kernel<METHOD, MATH, 16, 16, 0>, 0, 0};  // heavier nodes take the streaming path inside
}


static int decode_device(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                         double *llr, int32_t *iters, uint8_t *conv, bool may_repack = true);

// Serial schedule: one wavefront per 64-syndrome tile (bp_serial_kernel).  Device pointers, on h->stream.
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
    return LDPC_HIP_OK;
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
                              int32_t *iters, uint8_t *conv, const int32_t *orders = nullptr, int n_orders = 0) {
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
    if (h->serial_kernel != 0 && h->n > 0 && !orders) {  // (a schedule that changes per iteration has no fixed levels)
        if ((rc = ensure_serial_levels(h))) return rc;
        const double per_level = (double)h->n / (double)(h->n_levels ? h->n_levels : 1);
        if (h->serial_kernel == 1 || per_level >= 2.0) {
            level_waves = (int)(per_level + 0.999);
            if (level_waves > 8) level_waves = 8;
            if (level_waves < 1) level_waves = 1;
        }
    }
    if (level_waves) {
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
        a.orders = orders; a.n_orders = n_orders;
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
static int decode_serial_random(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                                int32_t *iters, uint8_t *conv) {
    const int n = h->n, max_iter = h->max_iter;
    // the arrangements of iterations 1 .. max_iter: std::shuffle on the object's std::mt19937, as RandomListShuffle does (rng.hpp:128-130)
    if ((size_t)max_iter * (size_t)(n ? n : 1) > ((size_t)1 << 28))
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "random serial schedule: max_iter x n = %d x %d orders exceed the 1 GiB table of per-iteration orders; lower max_iter", max_iter, n);
    std::vector<int32_t> orders((size_t)max_iter * (size_t)(n ? n : 1));
    {
        std::mt19937 g = h->sched_rng;
        std::vector<int> v(h->sched_state.begin(), h->sched_state.end());
        for (int it = 0; it < max_iter; ++it) {
            std::shuffle(v.begin(), v.end(), g);
            std::copy(v.begin(), v.end(), orders.begin() + (size_t)it * (size_t)n);
        }
    }
    int rc;
    if ((rc = h->sched_orders.ensure(orders.size() * sizeof(int32_t) + 16))) return rc;  // (+16: max_iter = 0 leaves the table empty)
    if (!iters) { if ((rc = h->sp_iters.ensure((size_t)batch * 4))) return rc; iters = (int32_t *)h->sp_iters.p; }
    HIPCHK(hipStreamSynchronize(h->stream));
    if (!orders.empty()) HIPCHK(hipMemcpy(h->sched_orders.p, orders.data(), orders.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    if ((rc = decode_serial_pass(h, max_iter, synd, batch, decoding, llr, iters, conv, (const int32_t *)h->sched_orders.p, max_iter))) return rc;
    int32_t last = 0;
    HIPCHK(hipMemcpyAsync(&last, iters + (batch - 1), sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> v(h->sched_state.begin(), h->sched_state.end());
    for (int it = 0; it < last; ++it) std::shuffle(v.begin(), v.end(), h->sched_rng);  // the last row consumed `last` shuffles
    std::copy(v.begin(), v.end(), h->sched_state.begin());
    return LDPC_HIP_OK;
}

static int decode_serial_relative(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                                  int32_t *iters, uint8_t *conv) {
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

static int decode_serial(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                         int32_t *iters, uint8_t *conv) {
    if (h->random_serial) return decode_serial_random(h, synd, batch, decoding, llr, iters, conv);  // (takes precedence, bp.hpp:467-469)
    if (h->schedule == 2) return decode_serial_relative(h, synd, batch, decoding, llr, iters, conv);
    int k1 = h->repack_iters < 0 ? h->max_iter / 8 : h->repack_iters;
    if (h->repack_iters < 0 && k1 < 2) k1 = 2;
    if (k1 <= 0 || k1 >= h->max_iter || batch <= 4 * LDPC_WAVE)
        return decode_serial_pass(h, h->max_iter, synd, batch, decoding, llr, iters, conv);
    const size_t B = (size_t)batch, m1 = (size_t)(h->m ? h->m : 1), n1 = (size_t)(h->n ? h->n : 1);
    int rc;
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
static int soft_info_device(ldpc_hip_bp *h, const double *soft, int64_t batch, double cutoff, double sigma, uint8_t *decoding,
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
    std::vector<int32_t> orders;
    int32_t *d_iters_last = nullptr;
    if (shuffled) {
        level_waves = 0;  // the levels belong to one fixed order
        if ((size_t)h->max_iter * (size_t)h->n > ((size_t)1 << 28))
            return fail(LDPC_HIP_ERR_UNSUPPORTED, "random serial schedule: max_iter x n = %d x %d orders exceed the 1 GiB table of per-iteration orders; lower max_iter", h->max_iter, h->n);
        orders.resize((size_t)h->max_iter * (size_t)h->n);
        std::vector<int> v(h->sched_state.begin(), h->sched_state.end());
        for (int it = 0; it < h->max_iter; ++it) {
            std::shuffle(v.begin(), v.end(), std::default_random_engine(h->sched_seed_raw));
            std::copy(v.begin(), v.end(), orders.begin() + (size_t)it * (size_t)h->n);
        }
        if ((rc = h->sched_orders.ensure(orders.size() * sizeof(int32_t) + 16))) return rc;
        if (!iters) { if ((rc = h->sp_iters.ensure((size_t)batch * 4))) return rc; iters = (int32_t *)h->sp_iters.p; }
        d_iters_last = iters + (batch - 1);
        HIPCHK(hipStreamSynchronize(h->stream));
        if (!orders.empty()) HIPCHK(hipMemcpy(h->sched_orders.p, orders.data(), orders.size() * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    void (*soft_kern)(const SoftArgs);
    if (h->max_row_deg <= 4 && h->max_col_deg <= 2) soft_kern = level_waves ? bp_softinfo_level_kernel<2, 4> : bp_softinfo_kernel<2, 4>;
    else if (h->max_row_deg <= 6 && h->max_col_deg <= 3) soft_kern = level_waves ? bp_softinfo_level_kernel<3, 6> : bp_softinfo_kernel<3, 6>;
    else if (h->max_row_deg <= 8 && h->max_col_deg <= 4) soft_kern = level_waves ? bp_softinfo_level_kernel<4, 8> : bp_softinfo_kernel<4, 8>;
    else soft_kern = level_waves ? bp_softinfo_level_kernel<0, 0> : bp_softinfo_kernel<0, 0>;
    if (lds > 48u * 1024u)
        HIPCHK(hipFuncSetAttribute((const void *)soft_kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
        if (shuffled && h->max_iter > 0) { a.orders = (const int32_t *)h->sched_orders.p; a.n_orders = h->max_iter; }
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
        if (last > h->max_iter) last = h->max_iter;
        if (last > 0) std::copy(orders.begin() + (size_t)(last - 1) * (size_t)h->n, orders.begin() + (size_t)last * (size_t)h->n, h->sched_state.begin());
    }
    return LDPC_HIP_OK;
}

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
    if (max_col <= 4) LDPC_PICK_WAVE(8, 4)
    LDPC_PICK_WAVE(8, 8)
#undef LDPC_PICK_WAVE
}

static WavePlan plan_wave(const ldpc_hip_bp *h, bool forced, bool want_llr, int64_t batch) {
    WavePlan p;
    if (h->m <= 0 || h->n <= 0 || h->nnz <= 0 || h->max_row_deg > 8 || h->max_col_deg > 8) return p;
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) pick_wave<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg, p);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) pick_wave<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg, p);
    else pick_wave<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg, p);
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
    const int u = ms ? (p.dr <= 4 ? 4 : 2) : (p.dr <= 6 ? 2 : 1);  // the kernel's nodes per lane in flight
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
    LDPC_PICK_WAVE_PS(8, 4)
#undef LDPC_PICK_WAVE_PS
}

static WavePsPlan plan_wave_ps(const ldpc_hip_bp *h, bool forced, bool want_llr, int64_t batch) {
    WavePsPlan p;
    if (h->bp_method != LDPC_HIP_PRODUCT_SUM || h->m <= 0 || h->n <= 0 || h->nnz <= 0 || h->max_row_deg > 8 || h->max_col_deg > 4) return p;
    if (h->math_mode == LDPC_HIP_MATH_FAST) pick_wave_ps<1>(h->max_row_deg, h->max_col_deg, p);
    else pick_wave_ps<0>(h->max_row_deg, h->max_col_deg, p);
    p.np = (h->n + 63) / 64 * 64;
    const size_t rm = (size_t)p.dr * h->m;
    if (rm + 2 >= 65536 || (size_t)p.np + 1 >= 65536) return p;
    if (!forced && (rm > 2 * (size_t)h->nnz || (size_t)p.dc * h->n > 2 * (size_t)h->nnz)) return p;  // padding would dominate
    p.shared = wave_ps_lds_shared(h->m, p.np, p.dr, p.dc);
    p.per_wave = wave_ps_lds_private(h->m, p.np, p.dr, want_llr);
    const size_t lds = 160u * 1024u - 64u;  // (the kernels' few bytes of static LDS come on top of the dynamic part)
    if (p.shared + p.per_wave > lds) return p;
    size_t w = (lds - p.shared) / p.per_wave;
    if (w > 16) w = 16;
    if (!forced && w < 8) return p;
    // A batch so small that every wavefront decodes only a few syndromes takes as long as its slowest syndrome: then the
    // workgroup's wavefronts share one (TEAM), one round of 64 entries each per pass.  LDPC_HIP_PS_TEAM=0 / 1 overrides (measurements).
    bool team = batch <= 256 * (int64_t)w * 8;  // (BB144, w = 16: 0.96 -> 0.57 ms at 8 192 syndromes, 1.52 -> 1.37 ms at 32 768, 4.2 -> 4.5 ms at 131 072)
    if (h->sw("PS_TEAM") >= 0) team = h->sw("PS_TEAM") != 0;
    if (team) {
        const size_t rounds = ((size_t)p.dr * h->m + 63) / 64;
        int tw = (int)(rounds < 2 ? 2 : rounds > 8 ? 8 : rounds);
        p.team = true;
        p.waves = tw;
        p.kern = p.kern_team;
        p.groups_per_cu = (int)(lds / (p.shared + p.per_wave));
        if (p.groups_per_cu * p.waves > 28) p.groups_per_cu = 28 / p.waves;  // (this kernel's 59 VGPRs allow 7 wavefronts per SIMD)
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
    if ((rc = h->counter.ensure(8))) return rc;
    HIPCHK(hipMemsetAsync(h->counter.p, 0, 8, h->stream));
    WavePsArgs a = {};
    a.m = h->m; a.n = h->n; a.np = p.np; a.max_iter = h->max_iter;
    a.batch = batch;
    a.rdeg = (const uint8_t *)h->wp_rdeg.p; a.col = (const uint16_t *)h->wp_col.p; a.epos = (const uint16_t *)h->wp_epos.p;
    a.llr0 = h->d_llr0;
    a.synd = synd; a.decoding = decoding; a.llr = llr; a.iters = iters; a.conv = conv;
    a.next = (unsigned long long *)h->counter.p;
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
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3((unsigned)(p.waves * 64)), (unsigned)dyn, h->stream, a);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// Wavefront-per-syndrome on-chip variant (bp_wave_kernel).  Device pointers, on h->stream.
static int decode_wave(ldpc_hip_bp *h, const WavePlan &p, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                       int32_t *iters, uint8_t *conv) {
    int rc;
    if ((rc = ensure_wave_tables(h, p))) return rc;
    if ((rc = h->counter.ensure(8))) return rc;
    HIPCHK(hipMemsetAsync(h->counter.p, 0, 8, h->stream));
    WaveArgs a = {};
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
    a.llr_direct = p.llr_direct ? 1 : 0;
    a.next = (unsigned long long *)h->counter.p;
    a.lds_shared = (int32_t)p.shared; a.lds_per_wave = (int32_t)p.per_wave;
    const size_t dyn = p.shared + (size_t)(p.team ? 1 : p.waves) * p.per_wave;
    if (dyn > 48u * 1024u)
        HIPCHK(hipFuncSetAttribute((const void *)p.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    int64_t groups = p.team ? batch : (batch + p.waves - 1) / p.waves;
    const int64_t resident = 256 * (int64_t)p.groups_per_cu;
    if (groups > resident) groups = resident;
    h->accumulated_ms = 0.f;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3((unsigned)(p.waves * 64)), (unsigned)dyn, h->stream, a);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
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
#define LDPC_EDGE_ROW(U) {nullptr, bp_edge_kernel<1, U>, bp_edge_kernel<2, U>, bp_edge_kernel<3, U>, bp_edge_kernel<4, U>, bp_edge_kernel<5, U>, \
        bp_edge_kernel<6, U>, bp_edge_kernel<7, U>, bp_edge_kernel<8, U>, bp_edge_kernel<9, U>, bp_edge_kernel<10, U>, bp_edge_kernel<11, U>, \
        bp_edge_kernel<12, U>, bp_edge_kernel<13, U>, bp_edge_kernel<14, U>, bp_edge_kernel<15, U>, bp_edge_kernel<16, U>}
    static void (*const kerns[2][17])(const EdgeArgs) = {LDPC_EDGE_ROW(false), LDPC_EDGE_ROW(true)};
#undef LDPC_EDGE_ROW
    p.uniform = true;
    for (int j = 1; j < h->n && p.uniform; ++j)
        p.uniform = std::memcmp(&h->channel_probs[(size_t)j], &h->channel_probs[0], sizeof(double)) == 0;
    p.rounds = rounds;
    p.kern = kerns[p.uniform ? 1 : 0][rounds];
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

static int decode_edge(ldpc_hip_bp *h, const EdgePlan &p, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                       int32_t *iters, uint8_t *conv) {
    int rc;
    if ((rc = ensure_edge_tables(h, p))) return rc;
    if ((rc = h->counter.ensure(8))) return rc;
    HIPCHK(hipMemsetAsync(h->counter.p, 0, 8, h->stream));
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
    a.next = (unsigned long long *)h->counter.p;
    const size_t dyn = edge_lds_bytes(p.rounds);
    // one wavefront per workgroup, as many resident as registers (4 or 5 per SIMD) and LDS allow
    int64_t per_cu = (int64_t)((160u * 1024u) / (dyn + 64));
    const int64_t by_regs = p.uniform ? 20 : 16;
    if (per_cu > by_regs) per_cu = by_regs;
    int64_t groups = batch < 256 * per_cu ? batch : 256 * per_cu;
    // a visit to the work counter costs ~1 us under load and one word serves ~88 of them per us: pull several syndromes at a
    // time once there are many per wavefront (the tail then is at most `chunk` syndromes of one wavefront)
    int64_t chunk = batch / (groups * 16);
    a.chunk = (int32_t)(chunk < 1 ? 1 : chunk > 8 ? 8 : chunk);
    h->accumulated_ms = 0.f;
    HIPCHK(hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(p.kern, dim3((unsigned)groups), dim3(64), (unsigned)dyn, h->stream, a);
    HIPCHK(hipEventRecord(h->ev1, h->stream));
    h->timed = true;
    HIPCHK(hipGetLastError());
    return LDPC_HIP_OK;
}

// nt: non-temporal cache policy for the message traffic (tiles that outgrow the 256 MB MALL; see MsgBufT)
static void pick_spread(const ldpc_hip_bp *h, bool nt, spread_kernel_t &kc, spread_kernel_t &kb) {
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) pick_spread_m<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg, nt, kc, kb);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) pick_spread_m<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg, nt, kc, kb);
    else pick_spread_m<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg, nt, kc, kb);
}

// Everything below runs on h->stream with device pointers only.
static int decode_stream_repacked(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                                  double *llr, int32_t *iters, uint8_t *conv);
static int decode_device(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                         double *llr, int32_t *iters, uint8_t *conv, bool may_repack) {
    const int64_t tiles_total = (batch + LDPC_WAVE - 1) / LDPC_WAVE;
    if (tiles_total == 0) return LDPC_HIP_OK;
    if (h->schedule == 0 || h->schedule == 2) return decode_serial(h, synd, batch, decoding, llr, iters, conv);
    if (h->small_mode != 0 && h->m > 0 && h->n > 0 && h->nnz > 0 && (int64_t)h->nnz * 16 < (1 << 22)) {
        // small code: keep the messages on chip.  Bounded degrees: one wavefront per syndrome (bp_wave_kernel).
        // Otherwise the slot kernel -- auto: the most resident syndromes (<= 4) per workgroup that still leave
        // four workgroups per CU (<= 39.5 KiB each); forced: whatever fits in 150 KiB
        if (h->small_mode != 2 && h->small_mode < 3) {  // product-sum: one lane per entry keeps the lanes busy with transcendentals
            const WavePsPlan pp = plan_wave_ps(h, h->small_mode == 1, llr != nullptr, batch);
            if (pp.waves) return decode_wave_ps(h, pp, synd, batch, decoding, llr, iters, conv);
        }
        if (h->small_mode == -1 || h->small_mode == 1 || h->small_mode == 6) {  // min-sum on the surface-code family: lane = edge
            const EdgePlan ep = plan_edge(h);
            if (ep.rounds) return decode_edge(h, ep, synd, batch, decoding, llr, iters, conv);
        }
        if (h->small_mode != 2) {
            const WavePlan wp = plan_wave(h, h->small_mode == 1 || h->small_mode >= 3, llr != nullptr, batch);
            if (wp.waves) return decode_wave(h, wp, synd, batch, decoding, llr, iters, conv);
        }
        int slots = 0;
        const size_t budget = (h->small_mode == 1 || h->small_mode == 2) ? 150u * 1024u : 39u * 1024u + 512u;
        for (int sl = 4; sl >= 1 && !slots; --sl)
            if (small_lds_bytes(h, sl) <= budget) slots = sl;
        if (slots) return decode_small(h, synd, batch, decoding, llr, iters, conv, slots);
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
    const int handoff = h->handoff < 0 ? 256 : h->handoff;
    h->last_chunk_tiles = chunk;
    if ((rc = h->tile_state.ensure(sizeof(TileState) * (size_t)chunk))) return rc;
    if ((rc = h->handoff_list.ensure(sizeof(int32_t) * (size_t)chunk))) return rc;
    if ((rc = h->counter.ensure(16))) return rc;
    if (!h->h_counters) HIPCHK(hipHostMalloc((void **)&h->h_counters, 16, hipHostMallocDefault));

    const int ring = h->regular ? h->ring_depth : 0;
    KernelChoice kern;
    if (h->bp_method == LDPC_HIP_MINIMUM_SUM) kern = pick_kernel<LDPC_HIP_MINIMUM_SUM, 0>(h->max_row_deg, h->max_col_deg, ring);
    else if (h->math_mode == LDPC_HIP_MATH_FAST) kern = pick_kernel<LDPC_HIP_PRODUCT_SUM, 1>(h->max_row_deg, h->max_col_deg, ring);
    else kern = pick_kernel<LDPC_HIP_PRODUCT_SUM, 0>(h->max_row_deg, h->max_col_deg, ring);
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
        if (h->m > 0) {
            dim3 g((unsigned)((h->m + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(pack_syndromes_kernel, g, dim3(256), 0, st, synd + b0 * h->m, nb, h->m,
                               (uint64_t *)h->par.p, (uint64_t *)h->nzm.p, (uint64_t *)h->invalid.p);
        }
        BpArgs a = {};
        a.m = h->m; a.n = h->n; a.nnz = h->nnz; a.max_iter = h->max_iter;
        a.ms_scaling_factor = h->ms_scaling_factor;
        a.batch = nb;
        a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx;
        a.col_ptr = h->d_col_ptr; a.csc_edge = h->d_csc_edge;
        a.llr0 = h->d_llr0;
        a.A = (double *)h->msgA.p; a.C = (double *)h->msgC.p;
        if (h->cont_A) { a.A = h->cont_A + (size_t)t0 * (size_t)h->nnz * LDPC_WAVE; a.it_start = h->cont_it_start; }
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
        HIPCHK(hipMemsetAsync(h->counter.p, 0, 16, st));

        // Wavefronts per workgroup (one workgroup = one 64-syndrome tile).  Register variant: 128 VGPRs,
        // 16 wavefronts per CU -> 4-wave workgroups once there are >= 4 tiles per CU.  Ring variant:
        // ~70 VGPRs and 6 KiB of LDS per wavefront -> 24 wavefronts per CU as two 12-wave workgroups
        // (3 wavefronts on each SIMD; measured best on MI355X, profiles/; 6-wave workgroups place
        // unevenly on the 4 SIMDs and 8-wave ones leave a ragged last round at 1024 tiles).
        int waves = h->waves_per_wg;
        if (waves <= 0) {
            if (kern.ring_depth) waves = tiles >= 512 ? 12 : 16;
            else waves = tiles >= 1024 ? 4 : (tiles >= 512 ? 8 : 16);
        }
        if (waves > 16) waves = 16;
        // ring variant: each wavefront owns RING slots of dynamic LDS; stay below the 160 KiB of a CU
        // + the parking space of the exact product-sum check row (LDPC_NEAR_BYTES per wavefront, behind the rings)
        const size_t near_bytes = (h->bp_method == LDPC_HIP_PRODUCT_SUM && h->math_mode == LDPC_HIP_MATH_LIBM_EXACT) ? LDPC_NEAR_BYTES : 0;
        const size_t lds_per_wave = (size_t)kern.ring_slot_bytes * (size_t)kern.ring_depth + near_bytes;
        while (lds_per_wave * (size_t)waves > 144u * 1024u) --waves;
        const size_t dyn_lds = lds_per_wave * (size_t)waves;
        if (dyn_lds > 48u * 1024u)
            HIPCHK(hipFuncSetAttribute((const void *)kern.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
        if (h->timed) {  // fold the previous chunk's time before the events are re-recorded
            float prev = 0.f;
            HIPCHK(hipEventSynchronize(h->ev1));
            HIPCHK(hipEventElapsedTime(&prev, h->ev0, h->ev1));
            h->accumulated_ms += prev;
            if (h->timed_mid) {
                HIPCHK(hipEventElapsedTime(&prev, h->ev0, h->ev_mid));
                h->accumulated_persistent_ms += prev;
            }
        }
        h->timed_mid = false;
        HIPCHK(hipEventRecord(h->ev0, st));
        SpreadArgs sa = {};
        sa.bp = a;
        sa.host_flag = h->d_flag;
        sa.seq = ++h->flag_seq ? h->flag_seq : ++h->flag_seq;  // never 0 (the word's initial value)
        // per-pass rounds: `grid_tiles` workgroup rows; how many of them have a tile is known to the host only when the
        // batch skips the persistent kernel (sa.n_tiles >= 0), otherwise the kernels read it from counters[1]
        unsigned grid_tiles = 0;
        int first_round = 1;  // a tile parked by the persistent kernel has completed >= 1 iteration
        if (handoff > 0 && tiles <= handoff && h->max_iter - a.it_start > 1) {
            // so few tiles that they would each sit on one compute unit: per-pass launches from the start
            grid_tiles = (unsigned)tiles;
            sa.n_tiles = (int32_t)tiles;
            sa.nodes = tiles <= 8 ? 1 : 4;
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
            HIPCHK(hipEventRecord(h->ev_mid, st));
            h->timed_mid = true;
            if (handoff > 0 && h->max_iter > 1) {
                // the persistent kernel parks at most `handoff` tiles (it starts parking when that many are unfinished);
                // how many it did park stays on the device
                grid_tiles = (unsigned)(tiles < handoff ? tiles : handoff);
                sa.n_tiles = -1;
                sa.nodes = 4;
            }
        }
        if (grid_tiles > 0) {
            // finish the parked tiles with chip-wide per-pass launches: check, bit, syndrome test, bookkeeping.  Every
            // round is queued at once; the host never waits.  A tile that is final (or a workgroup row without a tile)
            // leaves each kernel at its first instruction, and once the device has reported "nothing left" through
            // the host-mapped flag the host stops queueing -- which only matters when max_iter is far larger than
            // the iterations needed (the reference's default max_iter = n).
            spread_kernel_t kc, kb;
            // messages of the tiles in flight: 2 arrays x nnz x 512 B each; beyond ~the MALL they are streamed, not cached
            pick_spread(h, (double)grid_tiles * 2.0 * (double)per_tile_msg > 384.0 * 1024.0 * 1024.0, kc, kb);
            const unsigned per_wg = 4u * (unsigned)sa.nodes;
            const dim3 gc((unsigned)(h->m ? (h->m + per_wg - 1) / per_wg : 1), grid_tiles), gb((unsigned)(h->n ? (h->n + per_wg - 1) / per_wg : 1), grid_tiles);
            const dim3 gs((unsigned)(h->m ? (h->m + 255) / 256 : 1), grid_tiles), gf((unsigned)(h->n ? (h->n + 63) / 64 : 1), grid_tiles);
            const int rounds = h->max_iter - (first_round ? first_round : a.it_start);  // (a tile parked by the persistent kernel knows its own it0)
            const volatile unsigned *flag = h->h_flag;
            for (int round = 0; round < rounds; ++round) {
                if (*flag == sa.seq) break;  // a look, not a wait
                sa.round = round;
                hipLaunchKernelGGL(kc, gc, dim3(256), 0, st, sa);
                hipLaunchKernelGGL(kb, gb, dim3(256), 0, st, sa);
                hipLaunchKernelGGL(bp_spread_synd_kernel, gs, dim3(256), 0, st, sa);
                hipLaunchKernelGGL(bp_spread_finish_kernel, gf, dim3(256), 0, st, sa);
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
            dim3 g((unsigned)((h->n + 255) / 256), (unsigned)tiles);
            hipLaunchKernelGGL(unpack_decoding_kernel, g, dim3(256), 0, st,
                               (const uint64_t *)h->dec.p, nb, h->n, decoding + b0 * h->n);
            if (llr) {
                dim3 gt((unsigned)((h->n + LDPC_WAVE - 1) / LDPC_WAVE), (unsigned)tiles);
                hipLaunchKernelGGL(transpose_llr_kernel, gt, dim3(256), 0, st,
                                   (const double *)h->llr_t.p, nb, h->n, llr + (size_t)b0 * h->n);
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
// gather = reading one message array of every tile and writing the live share = (1 + (1 - F(k))) / 4 of an iteration (an
// iteration moves four arrays), plus the first pass's outputs for rows that are decoded on.  No work is wasted when nothing
// converges (the first call, and every call whose predecessor says "plain", run plain); results do not depend on any of this.
static int stream_first_pass_length(ldpc_hip_bp *h) {
    if (h->repack_iters > 0) return h->repack_iters < h->max_iter ? h->repack_iters : 0;
    if (!h->hist_pending || h->hist_max_iter != h->max_iter) return 0;
    if (hipEventSynchronize(h->ev_hist) != hipSuccess) return 0;
    const int full = h->max_iter, top = full < 255 ? full : 255;
    double total = 0;
    for (int j = 0; j < 256; ++j) total += h->h_hist[j];
    if (total <= 0) return 0;
    std::vector<double> F((size_t)top + 1, 0.0);  // F[j]: converged within j iterations
    double acc = 0;
    for (int j = 1; j <= top; ++j) { acc += h->h_hist[j]; F[(size_t)j] = acc / total; }
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
        const double cost = prefix + 0.25 * (1.0 + live) + 0.1 + live * rest;
        if (cost < best) { best = cost; best_k = k; }
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

static int decode_stream_repacked(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                                  double *llr, int32_t *iters, uint8_t *conv) {
    const int full = h->max_iter;
    const size_t B = (size_t)batch, m1 = (size_t)h->m, n1 = (size_t)h->n;
    int rc;
    if (!conv) { if ((rc = h->osd_conv.ensure(B))) return rc; conv = (uint8_t *)h->osd_conv.p; }
    if (!iters) { if ((rc = h->sp_iters.ensure(B * 4))) return rc; iters = (int32_t *)h->sp_iters.p; }
    const int k1 = stream_first_pass_length(h);
    if (k1 < 2 || k1 >= full) {
        if ((rc = decode_device(h, synd, batch, decoding, llr, iters, conv, false))) return rc;
        return stream_leave_histogram(h, iters, conv, batch);
    }
    if (!h->h_counters) HIPCHK(hipHostMalloc((void **)&h->h_counters, 16, hipHostMallocDefault));
    h->max_iter = k1;
    h->keep_state = true;
    rc = decode_device(h, synd, batch, decoding, llr, iters, conv, false);
    h->keep_state = false;
    h->max_iter = full;
    if (rc) return rc;
    if ((rc = h->osd_list.ensure(B * sizeof(int32_t)))) return rc;
    if ((rc = h->osd_counters.ensure(2 * sizeof(unsigned)))) return rc;
    HIPCHK(hipMemsetAsync(h->osd_counters.p, 0, 2 * sizeof(unsigned), h->stream));
    hipLaunchKernelGGL(osd_collect_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, h->stream, conv, batch,
                       (int32_t *)h->osd_list.p, (unsigned *)h->osd_counters.p);
    HIPCHK(hipMemcpyAsync(&h->h_counters[2], h->osd_counters.p, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));  // the size of the second pass is needed on the host
    const int64_t cnt = (int64_t)h->h_counters[2];
    if (cnt > 0) {
        float ms1 = 0.f;
        (void)ldpc_hip_bp_last_kernel_ms(h, &ms1);
        const size_t C = (size_t)cnt;
        if ((rc = h->rp_synd.ensure(C * m1)) || (rc = h->rp_dec.ensure(C * n1)) || (rc = h->rp_iters.ensure(C * 4)) ||
            (rc = h->rp_conv.ensure(C)) || (llr && (rc = h->rp_llr.ensure(C * n1 * 8)))) return rc;
        const int32_t *list = (const int32_t *)h->osd_list.p;
        auto grid = [](size_t items) { return flat_grid(items); };
        hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, grid(C * m1), dim3(256), 0, h->stream, synd, list, cnt, h->m, (uint8_t *)h->rp_synd.p);
        HIPCHK(hipGetLastError());
        // the listed rows' message state after k1 iterations, lane by lane, into dense tiles -- possible when the first pass kept the
        // whole batch's messages resident (one chunk) and ran the streamed kernels (they leave bit_to_check in msgA)
        const int64_t tiles1 = (batch + LDPC_WAVE - 1) / LDPC_WAVE, tiles2 = (cnt + LDPC_WAVE - 1) / LDPC_WAVE;
        const size_t per_tile = sizeof(double) * (size_t)h->nnz * LDPC_WAVE;
        bool carry_on = h->last_chunk_tiles >= tiles1 && h->nnz > 0 && !h->on("REPACK_RESTART");
        if (carry_on && h->rp_msg.ensure(per_tile * (size_t)tiles2)) { carry_on = false; (void)hipGetLastError(); }
        if (carry_on) {
            const int epw = 16;
            const dim3 gg((unsigned)((h->nnz + 4 * epw - 1) / (4 * epw)), (unsigned)tiles2);
            HIPCHK(hipEventRecord(h->ev0, h->stream));  // (the compaction belongs to this decode's kernel time)
            hipLaunchKernelGGL(gather_lane_state_kernel, gg, dim3(256), 0, h->stream, (const double *)h->msgA.p, list, cnt, h->nnz, epw, (double *)h->rp_msg.p);
            HIPCHK(hipEventRecord(h->ev1, h->stream));
            HIPCHK(hipGetLastError());
            HIPCHK(hipEventSynchronize(h->ev1));
            float gms = 0.f;
            HIPCHK(hipEventElapsedTime(&gms, h->ev0, h->ev1));
            ms1 += gms;
            h->cont_A = (double *)h->rp_msg.p;
            h->cont_it_start = k1;
        }
        rc = decode_device(h, (const uint8_t *)h->rp_synd.p, cnt, (uint8_t *)h->rp_dec.p, llr ? (double *)h->rp_llr.p : nullptr,
                           (int32_t *)h->rp_iters.p, (uint8_t *)h->rp_conv.p, false);
        h->cont_A = nullptr;
        h->cont_it_start = 0;
        if (rc) return rc;
        h->accumulated_ms += ms1;  // both passes count as this decode's kernel time
        hipLaunchKernelGGL(scatter_rows_kernel<uint8_t>, grid(C * n1), dim3(256), 0, h->stream, (const uint8_t *)h->rp_dec.p, list, cnt, h->n, decoding);
        if (llr) hipLaunchKernelGGL(scatter_rows_kernel<double>, grid(C * n1), dim3(256), 0, h->stream, (const double *)h->rp_llr.p, list, cnt, h->n, llr);
        hipLaunchKernelGGL(scatter_rows_kernel<int32_t>, grid(C), dim3(256), 0, h->stream, (const int32_t *)h->rp_iters.p, list, cnt, 1, iters);
        hipLaunchKernelGGL(scatter_rows_kernel<uint8_t>, grid(C), dim3(256), 0, h->stream, (const uint8_t *)h->rp_conv.p, list, cnt, 1, conv);
        HIPCHK(hipGetLastError());
    }
    return stream_leave_histogram(h, iters, conv, batch);
}

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
static int osd_status_pass(ldpc_hip_bp *h, const OsdArgs &a, int64_t batch) {
    int rc;
    if ((rc = h->osd_status.ensure((size_t)(batch ? batch : 1)))) return rc;
    HIPCHK(hipMemsetAsync(h->osd_status.p, 0, (size_t)batch, h->stream));
    int64_t blocks = batch < 4096 ? batch : 4096;
    hipLaunchKernelGGL(osd_status_kernel, dim3((unsigned)(blocks ? blocks : 1)), dim3(256), 0, h->stream, a, (uint8_t *)h->osd_status.p);
    HIPCHK(hipGetLastError());
    h->osd_status_rows = batch;
    return LDPC_HIP_OK;
}

static int bposd_device(ldpc_hip_bp *h, int osd_method, int osd_order, const uint8_t *synd, int64_t batch, uint8_t *decoding,
                        double *llr, int32_t *iters, uint8_t *conv) {
    if (osd_method == 0)  // OSD_OFF: BpOsdDecoder still calls OsdDecoder::decode, which then has no LU object -- refuse instead
        return fail(LDPC_HIP_ERR_INVALID, "osd_method is OSD_OFF");
    const bool higher = osd_method >= 2 && osd_order > 0;  // osd_order == 0 takes the OSD-0 branch whatever the method (osd.hpp:114)
    const size_t B = (size_t)batch, n = (size_t)h->n;
    int rc;
    if (!llr) { if ((rc = h->osd_llr.ensure(B * n * 8 ? B * n * 8 : 1))) return rc; llr = (double *)h->osd_llr.p; }
    if (!conv) { if ((rc = h->osd_conv.ensure(B ? B : 1))) return rc; conv = (uint8_t *)h->osd_conv.p; }
    h->osd_status_rows = 0;
    if ((rc = decode_device(h, synd, batch, decoding, llr, iters, conv))) return rc;
    if (h->m == 0 || h->n == 0) return LDPC_HIP_OK;
    OsdArgs a = {};
    a.m = h->m; a.n = h->n; a.words = (h->n + 1 + 63) / 64;
    a.batch = batch;
    a.row_ptr = h->d_row_ptr; a.col_idx = h->d_col_idx;
    a.synd = synd; a.llr = llr; a.conv = conv; a.decoding = decoding;
    a.method = osd_method; a.order = osd_order; a.wt = h->d_osd_wt;
    // small matrices: the elimination runs in registers (osd0_reg_kernel<R, W>), LDS only holds the column order
    void (*reg0)(const OsdArgs) = nullptr;
    if (!higher && h->osd_reg && !h->osd_big) {
        if (a.m <= 64 && a.words <= 2) reg0 = osd0_reg_kernel<1, 2>;
        else if (a.m <= 128 && a.words <= 4) reg0 = osd0_reg_kernel<2, 4>;
        else if (a.m <= 256 && a.words <= 8) reg0 = osd0_reg_kernel<4, 8>;
    }
    void (*regw)(const OsdArgs) = nullptr;
    if (higher && h->osd_reg && !h->osd_big && a.m <= 256 && a.words <= 8) {
        a.kwords = (osd_k(h) + 63) / 64;
        if (a.kwords < 1) a.kwords = 1;
        if (a.m <= 64 && a.words <= 2) regw = osdw_reg_kernel<1, 2>;
        else if (a.m <= 128 && a.words <= 4) regw = osdw_reg_kernel<2, 4>;
        else regw = osdw_reg_kernel<4, 8>;
    }
    size_t per_wave = reg0 ? (size_t)a.n * 4
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
    if ((rc = h->osd_list.ensure(B * sizeof(int32_t)))) return rc;
    if ((rc = h->osd_counters.ensure(2 * sizeof(unsigned)))) return rc;
    HIPCHK(hipMemsetAsync(h->osd_counters.p, 0, 2 * sizeof(unsigned), h->stream));
    a.list = (const int32_t *)h->osd_list.p;
    a.counters = (unsigned *)h->osd_counters.p;
    hipLaunchKernelGGL(osd_collect_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, h->stream, conv, batch,
                       (int32_t *)h->osd_list.p, (unsigned *)h->osd_counters.p);
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
    if ((rc = osd_status_pass(h, a, batch))) return rc;
    // Rows whose syndrome lies outside the image of H (status 2; only a rank-deficient H has any): the reference's answer depends on
    // which rows its linked-list elimination made pivot rows.  One workgroup per such row re-enacts that choice and writes the syndrome
    // that keeps exactly those rows (osd_exact_kernel.h); the same OSD kernels then run once more over these rows.  No host round trip:
    // both launches size themselves from device-side counters and cost a few microseconds when there is nothing to do.
    const bool rank_known = (double)a.m * a.m * a.words < 4e9;
    if (rank_known && a.n - osd_k(h) < a.m && a.m <= 8192 && !h->on("OSD_NO_EXACT")) {
        const size_t slot_words = osd_exact_slot_words(a.m, a.n);
        int64_t slots = 512;
        if (slots > batch) slots = batch;
        const int64_t cap = (int64_t)((2ull << 30) / (slot_words * 8));  // at most 2 GiB of working copies
        if (cap >= 1) {
            if (slots > cap) slots = cap;
            if ((rc = h->osd_fix_synd.ensure(B * (size_t)a.m)) || (rc = h->osd_fix_list.ensure(B * sizeof(int32_t))) ||
                (rc = h->osd_fix_counters.ensure(2 * sizeof(unsigned))) || (rc = h->osd_fix_scratch.ensure((size_t)slots * slot_words * 8))) return rc;
            HIPCHK(hipMemsetAsync(h->osd_fix_counters.p, 0, 2 * sizeof(unsigned), h->stream));
            OsdExactArgs X = {};
            X.o = a;
            X.status = (const uint8_t *)h->osd_status.p;
            X.corrected = (uint8_t *)h->osd_fix_synd.p;
            X.list2 = (int32_t *)h->osd_fix_list.p;
            X.counters2 = (unsigned *)h->osd_fix_counters.p;
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
            a2.counters = (unsigned *)h->osd_fix_counters.p;
            if ((rc = run_osd(a2))) return rc;
        }
    }
    return LDPC_HIP_OK;
}

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
    if (osd_method == 3 && osd_order > 64)
        return fail(LDPC_HIP_ERR_UNSUPPORTED, "OSD_CS with osd_order > 64 is not available on the device");
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
    if (h_synd && h_dec && (!llr || h_llr) && (!iters || h_it) && (!conv || h_cv) && pin_need <= ldpc_hip_bp::PIN_BYTES && !h->on("NO_PINNED_PATH")) {
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
            if ((rc = osd >= 0 ? bposd_device(h, osd ? h->osd_method : 1, osd ? h->osd_order : 0, dv, batch, dv + o_dec, p_llr, p_it, p_cv)
                               : decode_device(h, dv, batch, dv + o_dec, p_llr, p_it, p_cv))) return rc;
            HIPCHK(hipStreamSynchronize(h->stream));
            std::memcpy(decoding, h->pin_host + o_dec, B * n);
            if (llr) std::memcpy(llr, h->pin_host + o_llr, B * n * 8);
            if (iters) std::memcpy(iters, h->pin_host + o_it, B * 4);
            if (conv) std::memcpy(conv, h->pin_host + o_cv, B);
            return LDPC_HIP_OK;
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

#include "multi_device.h"  // ldpc_hip_bp_multi_*: one decoder over several GPUs in one process (host code over the entry points above)

#ifdef LDPC_HIP_OSD_CLOCKS
extern "C" int ldpc_hip_debug_osd_clocks(unsigned long long *out, int reset) {
    if (out) HIPCHK(hipMemcpyFromSymbol(out, HIP_SYMBOL(osd_phase_clocks), sizeof(unsigned long long) * 16));
    if (reset) { unsigned long long z[16] = {}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(osd_phase_clocks), z, sizeof z)); }
    return LDPC_HIP_OK;
}
#endif
