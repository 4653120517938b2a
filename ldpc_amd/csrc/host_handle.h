// host_handle.h -- the handle (struct ldpc_hip_bp), error reporting, device buffers, measurement switches
// Part of libldpc_hip.so: included by every translation unit (bp_hip.hip = the C ABI; tu_stream / tu_serial / tu_onchip / tu_osd.hip = one kernel
// family each with its host side).  What one unit calls in another is declared at the end of this header.
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <random>
#include <atomic>
#include <thread>

#include <sys/mman.h>

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------

inline thread_local std::string g_last_error;  // (one per thread for the whole library: C++17 inline variable)

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
static const char *const k_switch_names[] = {"TEAM_WAVES", "TEAM_PRIOR_LDS", "PS_TEAM", "EXPLICIT_INIT", "DEBUG_HANDOFF",
                                             "OSD_UNBLOCKED", "OSD_PLANES", "OSD_PER_CU", "OSD_NO_EXACT", "NO_PINNED_PATH", "KEEP_LAST_MESSAGES", "PS_TEAM_WAVES", "EDGE_STATIC_PCT", "EDGE_CHUNK", "NO_HOST_PIPELINE", "NO_DIRECT_LLR", "HOST_CHUNK_ROWS", "TIME_SMALL_CALLS", "REL_LDS", "HOST_PIPE_TIMING", "REL_LEVELS", "REL_PROF", "REL_SCRATCH_IN_L", "SER_RING", "SER_WAVES", "SER_LANE_MAX", "SER_LANE_THREADS", "SER_WAVES2", "RESIDENT", "RESIDENT_LINGER_US", "NO_SPREAD_COMPACT", "HOST_TAPER", "FLOOD_LANES", "FLOOD_LANE_GROUPS", "SER_NO_REMAINDER", "SER_ROUND_TILES", "VAR_RING", "VAR_RING_UNITS", "SPREAD_NODES", "SPREAD_NODES2", "SER_VAR", "SER_VAR_UNITS", "REL_EXT", "REL_FIRST_ONCE", "REPACK2", "OSD_COLLECT_AFTER", "EDGE_CLAMP", "OSD_NO_FLAT"};
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
    DeviceBuf var_row_items, var_pair_items;  // item tables of the variable-degree ring (BpArgs::row_items, pair_items; host_stream.h: ensure_var_ring_items)
    bool var_items_built = false;
    // continuation of a first pass (decode_stream_repacked): decode_device takes its message state from here and counts on from cont_it_start
    double *cont_A = nullptr, *cont_C = nullptr;  // its bit_to_check (compacted) / check_to_bit arrays
    int32_t cont_it_start = 0;
    const int32_t *cont_row_map = nullptr;        // its rows in the caller's arrays (BpArgs::row_map)
    const unsigned *cont_rows_dev = nullptr;      // {rows, tiles} on the device (BpArgs::rows_dev)
    int64_t cont_alive[4] = {-1, -1, -1, -1}, cont_alive_total = 0;  // ... still running after the first pass + 0 .. 3 iterations, of how many
    int64_t cont_late_rows = -1;                  // rows the steering histogram expects to be still running 8 iterations into it (-1: unknown)
    int64_t cont_grid_tiles = 0;                  // grid.y of its tile-looping kernels (an estimate; they loop)
    const int32_t *cont_src_map = nullptr;        // where its rows sit in the tiles the state is gathered from (nullptr: cont_row_map -- tiles of the caller's rows)
    bool cont_extend = false;                     // a THIRD pass (second compaction): its kernels extend the second pass's timed interval
    bool keep_state = false;       // this decode_device call is a first pass: its last bit pass must leave the messages behind
    int64_t last_chunk_tiles = 0;  // tiles per chunk of the last streamed decode (== its tile count: the whole batch's state is resident)
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
    unsigned long long *d_clk = nullptr;  // {shader cycles, constant-rate ticks} summed over the workgroups of the long-running BP kernels (clock_probe_*)
    int32_t schedule = 1;    // ldpc::bp::BpSchedule (bp.hpp:28-32): 0 serial, 1 parallel, 2 serial_relative
    // What the reference keeps in the decoder OBJECT from one decode to the next (bp.hpp:67, 75): serial_schedule_order -- the
    // arrangement serial_relative re-sorts and the random schedule re-shuffles every iteration -- and the generator of the shuffles.
    std::vector<int32_t> sched_state;
    std::mt19937 sched_rng;
    int32_t sched_seed_raw = 0;  // random_schedule_seed as given (the soft-syndrome routine seeds its own engine with it)
    bool random_serial = false;
    DeviceBuf rel_ord, rel_dbit, sched_orders, sched_order0;
    DeviceBuf sched_lvl_bits, sched_lvl_ptr;      // the random schedule's orders once more, level-major, and their level bounds (host_serial.h: random_orders_*)
    DeviceBuf rl_edge, rl_chk, rl_cdeg, rl_last;  // per-column tables and the last row's final order of bp_relative_lds_kernel
    int rl_dc = 0;                                // stride the tables were built for (0: none)
    DeviceBuf flood_src;                          // decode_stream_repacked's second compaction: positions of the third pass's rows in the second pass's tiles
    DeviceBuf rl_first;                           // the order after the first iteration's sort of the current call (RelLdsArgs::first_order)
    DeviceBuf rl_rec, rl_ext_A;                   // EXT form: the per-entry records [n][dc] u64; the wavefronts' message slots [workgroups][wavefronts][nnz] f64
    bool rl_rec_valid = false;
    // The random schedule's table of per-iteration orders (host_serial.h: random_orders_*), kept on the device between calls as a
    // ring of max_iter rows: a call consumes as many rows as its LAST row ran iterations, and only those are generated anew.
    struct RandomOrders {
        bool valid = false;
        int kind = 0, rows = 0, n = 0, first = 0;   // kind 0: one std::mt19937 stream (bp.hpp:467-469); 1: a new default_random_engine(seed) per shuffle (bp.hpp:573-577)
        int32_t seed_raw = 0;
        std::mt19937 rng_end;                       // generator behind the table's last row (kind 0)
        std::vector<int> row_end;                   // the table's last row
        std::vector<int32_t> csc_ptr, csc_row;      // the checks of every bit (levels of an order: host_serial.h random_orders_levels)
        std::vector<int32_t> n_levels;              // levels of every row of the ring (0: row not built)
        std::vector<int32_t> expect_state;          // what the handle's order / generator must be for the table to be current
        std::mt19937 expect_rng;
    } rnd;
    int32_t *d_csc_row = nullptr, *d_order = nullptr;
    bool custom_order = false;
    DeviceBuf counter;
    // BP + OSD: where an on-chip BP kernel that can do so lists the rows it leaves unconverged (bposd_device arms it around its decode_device call;
    // `done` says a kernel honoured it -- and then also zeroed the counters, which sit behind the work pools in `counter`)
    struct { int32_t *list; unsigned *count; uint8_t *status; bool armed, done; } osd_hook = {nullptr, nullptr, nullptr, false, false};
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
    hipEvent_t evp0 = nullptr, evp1 = nullptr, evp_mid = nullptr;  // the same of the first pass of a two-pass decode (decode_stream_repacked)
    bool timed_prev = false, timed_prev_mid = false;
    hipEvent_t ev_done = nullptr;  // end of the last call that queued work on `stream` (orders a change of stream after it)
    bool work_queued = false;
    bool timed = false, timed_mid = false;
    bool untimed_call = false;  // a single decode() through the host-mapped block: the two timing events cost more than they tell (ldpc_hip_bp_last_kernel_ms then says 0)
    float accumulated_ms = 0.f, accumulated_persistent_ms = 0.f;

    DeviceBuf msgA, msgC, par, nzm, invalid, dec, dcur, llr_t;       // workspace
    DeviceBuf st_synd, st_dec, st_llr, st_iters, st_conv, st_misc;  // staging for host pointers
    // small calls with host buffers (a single decode()): one host-mapped, coherent block that the kernels read and write in place --
    // no copy commands at all, one launch sequence and one wait
    unsigned char *pin_host = nullptr, *pin_dev = nullptr;
    static constexpr size_t PIN_BYTES = 512u * 1024u;
    static constexpr size_t PIN_MAIL = PIN_BYTES - 64;  // the last 64 bytes: the resident kernel's mailbox (WavePsArgs::mail)
    // A single decode() of a small code (host buffers, product-sum, parallel schedule) is served by a RESIDENT workgroup that stays for
    // `linger` after its last request (host_onchip.h: decode_onchip_resident): no launch, no completion, tables already in LDS.
    struct Resident {
        hipStream_t stream = nullptr;
        hipEvent_t ended = nullptr;       // behind the resident kernel's launch: has it left?
        bool launched = false;            // a launch whose end has not been seen yet
        unsigned seq = 0;                 // the last request number handed out
        unsigned long long key[6] = {};   // what the kernel in flight was launched for (parameters, priors, plan): a change retires it
    } res;
    unsigned long long priors_version = 0;  // bumped whenever the priors on the device change (upload_priors)
    // large calls with host buffers: pinned double-buffered chunks, so that PCIe and the host's own copies overlap the kernels
    // (host_decode_abi.h: decode_batch_pipelined)
    struct HostPipe {
        static constexpr int NB = 3;  // chunks in flight: one being decoded, one on its way out over PCIe, one being copied into the caller's arrays
        hipStream_t s_in = nullptr, s_out = nullptr;
        hipEvent_t ev_in[NB] = {}, ev_cmp[NB] = {}, ev_out[NB] = {};
        unsigned char *pin_in[NB] = {}, *pin_out[NB] = {};
        size_t pin_in_cap = 0, pin_out_cap = 0;
        DeviceBuf d_in[NB], d_dec[NB], d_llr[NB], d_it[NB], d_cv[NB];
    } pipe;
    DeviceBuf osd_llr, osd_conv;                                    // BP outputs OSD-0 needs when the caller does not ask for them
    DeviceBuf osd_scratch;                                          // working copies of H for osd0_big_kernel
    DeviceBuf osd_packed;                                           // [m][words] H bit-packed by rows (register OSD kernels)
    DeviceBuf osd_ell;                                              // [m][8] a row's entries as 16-bit column numbers (osd0_flat_kernel)
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
    std::vector<int32_t> h_lvl_ptr, h_lvl_bits;                     // host copies (the streamed serial kernel's position records are built from them)
    DeviceBuf flood_list2, flood_pos;                                // rows a second pass of a few rounds left, and every row's place in that pass's tiles
    DeviceBuf flood_lane_scratch;                                    // bp_flood_lane_kernel: a workgroup's two row-major message arrays, 1024 workgroups
    DeviceBuf ser_pos_e0;                                            // ... and the initial values of every position's other entries (first iteration)
    DeviceBuf ser_rows[2], ser_synd2;                                // decode_serial_streamed: the rows of a compacted pass (numbers in the caller's arrays), their syndromes
    DeviceBuf ser_pos_tab;                                           // bp_serial_stream_kernel: one record per position of the level-major order
    bool ser_pos_valid = false;                                     // ... describing the current levels
    DeviceBuf ser_var_init;                                         // ... [nnz][64] initial segments shared by all tiles (SerialArgs::var_init), rewritten by every decode that uses it
    DeviceBuf ser_var_items, ser_var_wq, ser_var_lane_items, ser_var_lane_lvl;  // bp_serial_var_kernel.h: item streams per (level, wavefront); the level-major item list of the lane kernel
    bool ser_var_valid = false;                                     // ... describing the current levels, for ser_var_waves wavefronts per tile
    int ser_var_waves = 0;
    int32_t min_col_deg = 0;
    // repacking of the streamed parallel schedule (decode_stream_repacked), steered by what the previous decode looked like
    DeviceBuf sp_hist, sp_iters;     // iteration histogram of the last streamed decode (256 bins) / iteration counts when the caller wants none
    unsigned *h_hist = nullptr;      // pinned copy of the histogram
    hipEvent_t ev_hist = nullptr;    // the copy has landed
    bool hist_pending = false;
    int32_t hist_max_iter = 0;
    unsigned hist_landed[256] = {};  // the last histogram whose copy was SEEN complete (never waited for)
    bool hist_landed_valid = false;
    int32_t hist_landed_max_iter = 0;
    int32_t repack_iters = -1;                                      // first-pass iterations: -1 auto (max_iter / 8), 0 = no repacking
    DeviceBuf soft_S, soft_in, soft_out;                             // soft-syndrome decoding: scaled analog syndromes, staging
    DeviceBuf b8_in, b8_out, b8_synd, b8_dec, obs_row_ptr, obs_col_idx;  // bit-packed shot I/O and the observables matrix
    int32_t obs_k = -1;                                              // rows of the observables matrix (-1: not set)
    int64_t max_chunk_tiles = 0;                                     // 0 = decide from free memory
};

static int upload_priors(ldpc_hip_bp *h) {
    ++h->priors_version;
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

static bool is_pinned_host_ptr(const void *p) {  // page-locked host memory (hipHostMalloc / hipHostRegister): a copy engine can write it directly
    if (!p) return false;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeHost;
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

// ---- what the translation units call in each other (device pointers, on h->stream) -------------------------------------------------
// tu_stream.hip: the dispatch of a batch to a kernel family, and the streamed kernels themselves
int decode_device(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv,
                  bool may_repack = true);
// tu_onchip.hip: ONE syndrome in the handle's host-mapped block (already copied in) through the resident workgroup; *took = false: not
// applicable (then the ordinary path).  resident_retire: the resident workgroup leaves now (destroy, or a change it must not outlive)
int decode_onchip_resident(ldpc_hip_bp *h, bool want_llr, bool *took);
void resident_retire(ldpc_hip_bp *h);
// tu_onchip.hip: the kernels that keep a syndrome's messages on chip; *took = false: no such kernel applies to this matrix
int decode_onchip(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv, bool *took);
// tu_serial.hip: serial / serial_relative / random serial schedules, soft-syndrome decoding
int decode_serial(ldpc_hip_bp *h, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr, int32_t *iters, uint8_t *conv);
int soft_info_device(ldpc_hip_bp *h, const double *soft, int64_t batch, double cutoff, double sigma, uint8_t *decoding, double *llr,
                     int32_t *iters, uint8_t *conv, double *soft_out);
// tu_osd.hip: BP followed by ordered-statistics post-processing of the rows it left unconverged
int bposd_device(ldpc_hip_bp *h, int osd_method, int osd_order, const uint8_t *synd, int64_t batch, uint8_t *decoding, double *llr,
                 int32_t *iters, uint8_t *conv);
