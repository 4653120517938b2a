/*
 * ldpc_hip.h -- C ABI of libldpc_hip.so: batched flooding-schedule belief propagation on MI355X.
 *
 * This is the drop-in boundary for the one hot path of quantumgizmos/ldpc that this library
 * replaces: ldpc.BpDecoder.decode (src_python/ldpc/bp_decoder/_bp_decoder.pyx:642-695) ->
 * ldpc::bp::BpDecoder::decode (src_cpp/bp.hpp:159-190) -> bp_decode_parallel (bp.hpp:192-325),
 * applied to a BATCH of independent syndromes.  The reference has no FFI for this path today (its
 * Cython layer includes bp.hpp directly, _bp_decoder.pxd:9-83); these entry points are what a
 * binding for a device decoder would bind, one per reference member it stands in for.
 * INTEGRATION.md shows the Cython/ctypes stub a maintainer would add.
 *
 * Conventions
 *   - plain C types only; every function returns 0 (LDPC_HIP_OK) or a negative ldpc_hip_status;
 *     ldpc_hip_last_error() returns a thread-local message for the last failure.  No exceptions
 *     and no aborts cross this boundary (the reference's decode() is not `except +`,
 *     _bp_decoder.pxd:78).
 *   - the caller owns every buffer it passes; the library owns device memory, streams and events
 *     inside the handle.  Data pointers may be host or device pointers (detected per call).
 *   - a handle is single-consumer, like the reference object (its messages live inside H,
 *     bp.hpp:42-48); distinct handles are independent.  An ldpc_hip_bp handle lives on one GPU; an
 *     ldpc_hip_bp_multi handle (end of this file) shards a batch over several GPUs of the node.
 *   - a handle owns ONE workspace and launches on ONE stream at a time (ldpc_hip_bp_set_stream).  The
 *     *_async entry points only queue work; queue the next call on the same stream, or change the stream
 *     first -- ldpc_hip_bp_set_stream orders the new stream after everything the handle has queued so far.
 */
#ifndef LDPC_HIP_H
#define LDPC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LDPC_HIP_OK = 0,
    LDPC_HIP_ERR_INVALID = -1,  /* bad argument (shape, range, NULL) */
    LDPC_HIP_ERR_DEVICE = -2,   /* HIP runtime error (message has hipGetErrorString) */
    LDPC_HIP_ERR_NOMEM = -3,    /* workspace does not fit */
    LDPC_HIP_ERR_UNSUPPORTED = -4
} ldpc_hip_status;

/* ldpc::bp::BpMethod, bp.hpp:23-26 */
#define LDPC_HIP_PRODUCT_SUM 0
#define LDPC_HIP_MINIMUM_SUM 1

/*
 * Decoder description = the arguments of ldpc::bp::BpDecoder::BpDecoder (bp.hpp:77-88) that
 * the parallel schedule uses, with the parity-check matrix as CSR instead of a BpSparse&
 * (Py2BpSparse, _bp_decoder.pyx:9-49, builds that from the same coordinates).
 */
typedef struct {
    int32_t m;                    /* check_count  (bp.hpp:56) */
    int32_t n;                    /* bit_count    (bp.hpp:57) */
    int32_t nnz;                  /* = csr_row_ptr[m] */
    const int32_t *csr_row_ptr;   /* [m+1], host */
    const int32_t *csr_col_idx;   /* [nnz], host, strictly ascending inside each row */
    const double *channel_probs;  /* [n], host: channel_probabilities (bp.hpp:55) */
    int32_t max_iter;             /* maximum_iterations (bp.hpp:58), >= 1 */
    int32_t bp_method;            /* LDPC_HIP_PRODUCT_SUM | LDPC_HIP_MINIMUM_SUM (bp.hpp:59) */
    double ms_scaling_factor;     /* bp.hpp:61; 0.0 selects alpha = 1 - 2^-it (bp.hpp:222-228) */
    int32_t device;               /* HIP device ordinal; -1 = current device */
} ldpc_hip_bp_desc;

typedef struct ldpc_hip_bp ldpc_hip_bp; /* opaque handle */

/* replaces: BpDecoderBase.__cinit__ -> new BpDecoderCpp(...)  (_bp_decoder.pyx:118-132) */
int ldpc_hip_bp_create(const ldpc_hip_bp_desc *desc, ldpc_hip_bp **out);

/* replaces: BpDecoderBase.__dealloc__ (_bp_decoder.pyx:162-165) */
void ldpc_hip_bp_destroy(ldpc_hip_bp *h);

/* replaces: writes to bpd.channel_probabilities by the error_rate / error_channel setters and
 * update_channel_probs (_bp_decoder.pyx:180-223).  Priors log((1-p)/p) (bp.hpp:150-151) are
 * evaluated on the host in double precision and uploaded. */
int ldpc_hip_bp_set_channel(ldpc_hip_bp *h, const double *channel_probs, int32_t n);

/* replaces: the max_iter / bp_method / ms_scaling_factor setters (_bp_decoder.pyx:342-394,487-499) */
int ldpc_hip_bp_set_params(ldpc_hip_bp *h, int32_t max_iter, int32_t bp_method,
                           double ms_scaling_factor);

/* replaces: the schedule / serial_schedule_order setters (_bp_decoder.pyx:415-483).  `schedule` uses
 * ldpc::bp::BpSchedule's values (bp.hpp:28-32): 1 = PARALLEL (flooding, default), 0 = SERIAL, 2 = SERIAL_RELATIVE;
 * `serial_schedule_order` (n entries, or NULL for 0..n-1; bp.hpp:110-124) is the order a serial sweep walks.
 * SERIAL with a fixed order runs the tile-wide serial kernels.  Two variants keep STATE in the reference's decoder object
 * and change the order while decoding (bp.hpp:467-483): SERIAL_RELATIVE re-sorts it at the start of every iteration
 * (std::sort by descending prior, then by descending posterior of the previous iteration -- reproduced swap for swap,
 * since ties make the unstable sort's arrangement observable), the random serial schedule (ldpc_hip_bp_set_random_serial)
 * re-shuffles it (std::shuffle on a std::mt19937).  For these, every row of a call starts from the handle's current state
 * (order, generator) and the call leaves the state of its LAST row: a one-row call is exactly one BpDecoder::decode, a
 * sequence of one-row calls exactly a sequence of decodes on one reference object, and a batch on a fresh handle equals a
 * new reference object per row.  Such calls return after the device has finished (the state comes back to the host).
 * This call (re)sets the state to `serial_schedule_order`. */
int ldpc_hip_bp_set_schedule(ldpc_hip_bp *h, int32_t schedule, const int32_t *serial_schedule_order);
/* replaces: the random_serial_schedule / random_schedule_seed setters (_bp_decoder.pyx:527-579 -> bp.hpp:142-145): with
 * `enable` the serial sweep shuffles the order before every iteration; `seed` re-seeds the generator (0 = from the clock, as
 * rng.hpp:117-123 does).  Takes precedence over SERIAL_RELATIVE, as in bp.hpp:467-469.
 * ldpc_hip_bp_soft_info_decode_batch honours the flag the way soft_info_decode_serial does (bp.hpp:573-577): at the top of every
 * iteration that still runs the order is rearranged by std::shuffle with a NEW std::default_random_engine((int32_t)seed) -- the
 * seed as given, 0 included --; rows and state as described above. */
int ldpc_hip_bp_set_random_serial(ldpc_hip_bp *h, int32_t enable, uint32_t seed);
/* the handle's current serial_schedule_order (n entries): what bpd.serial_schedule_order holds after a decode */
int ldpc_hip_bp_get_schedule_order(ldpc_hip_bp *h, int32_t *order);

/* Launch stream (a hipStream_t) for decode calls; NULL selects the handle's own (non-blocking) stream,
 * LDPC_HIP_STREAM_LEGACY_DEFAULT the device's legacy default stream (hipStream_t 0, which is what
 * e.g. torch.cuda.current_stream().cuda_stream reports when no stream context is active). */
#define LDPC_HIP_STREAM_LEGACY_DEFAULT ((void *)1)
int ldpc_hip_bp_set_stream(ldpc_hip_bp *h, void *hip_stream);

/*
 * replaces: BpDecoder.decode for `batch` syndromes at once (_bp_decoder.pyx:676-695 marshalling +
 * bp.hpp:192-325).  Row b of every array belongs to syndrome b.
 *
 *   syndromes  [batch x m] uint8, C-contiguous       in   (bytes as the reference sees them: any
 *                                                          non-zero byte flips the product-sum sign,
 *                                                          min-sum uses the byte's parity, a byte > 1
 *                                                          can never converge: bp.hpp:213,236,300)
 *   decoding   [batch x n] uint8                      out  bpd.decoding          (bp.hpp:62)
 *   llr        [batch x n] float64, or NULL           out  bpd.log_prob_ratios   (bp.hpp:65)
 *   iterations [batch]     int32,  or NULL            out  bpd.iterations        (bp.hpp:69)
 *   converge   [batch]     uint8,  or NULL            out  bpd.converge          (bp.hpp:71)
 *
 * Per-syndrome early exit is preserved: the outputs of row b are those of the first iteration whose
 * hard decision reproduces syndrome b (bp.hpp:300-308), or of iteration max_iter.
 * Synchronous with respect to the host: returns after the results are in the output buffers.
 * Host buffers (the reference API's own mode: NumPy in, NumPy out): a call of at most a few syndromes on a small code works in a
 * host-mapped block (no copy commands).  ONE syndrome of a small code, product-sum, parallel schedule -- the reference's
 * `for shot: decoder.decode(shot)` -- is served by a RESIDENT workgroup (bp_wave_ps_kernel's team form; csrc/host_onchip.h): the kernel
 * that decoded the last syndrome is still there, tables in LDS, polling a request word in the block; the call writes its syndrome,
 * bumps the word and spins on the served word -- no launch, no completion (BB144: 31 -> 22 us a call, hamming(5): 60 -> 50 us;
 * profiles/r5_single_decode_latency.txt).  The workgroup leaves by itself 100 us after its last request ("RESIDENT_LINGER_US"), at
 * once when the handle's parameters or priors change or the handle is destroyed; "RESIDENT" 0 = a launch per call.  A large batch (>= 64 MiB of data in at least three chunks, everything in host memory, BP
 * alone, rows independent of one another: the parallel and the fixed-order serial schedule) is cut into chunks of whole tiles --
 * <= 16 384 rows, ~256 MiB of results -- that move through PINNED double buffers on two copy streams: while the kernels decode
 * chunk c, chunk c + 1 is on its way in, chunk c - 1 on its way out, and the calling thread copies chunk c - 2 from the pinned
 * buffer into the caller's (pageable) arrays; with log-ratios the last quarter of the batch goes in chunks that halve down to 4 096 rows
 * (the last chunk's results cross PCIe with nothing left to overlap them: "HOST_TAPER" 0 = uniform chunks).  Results are those of one undivided call.  Debug switches "NO_HOST_PIPELINE",
 * "HOST_CHUNK_ROWS" (tests, measurements).  Measured (bench.py `host_io`): 0.93 of the device-resident rate without LLRs.
 */
int ldpc_hip_bp_decode_batch(ldpc_hip_bp *h, const uint8_t *syndromes, int64_t batch,
                             uint8_t *decoding, double *llr, int32_t *iterations,
                             uint8_t *converge);

/* Same, but only enqueues on the launch stream and returns without waiting for the device (no host synchronisation:
 * the number of tiles the streaming kernel hands to the per-pass kernels stays on the device, and the per-pass rounds
 * are queued for all of max_iter -- a host-mapped flag the device sets when nothing is left merely lets the host stop
 * queueing early).  All pointers must be device pointers.  Exceptions: a batch that does not fit the workspace in one
 * piece (> 2 097 152 syndromes, or device memory) waits between pieces; the repacked schedules (ldpc_hip_bp_set_repack)
 * wait where noted there. */
int ldpc_hip_bp_decode_batch_async(ldpc_hip_bp *h, const uint8_t *syndromes, int64_t batch,
                                   uint8_t *decoding, double *llr, int32_t *iterations,
                                   uint8_t *converge);

/*
 * replaces: BpOsdDecoder.decode with osd_method = OSD_0 for `batch` syndromes
 * (src_python/ldpc/bposd_decoder/_bposd_decoder.pyx:125-134): belief propagation as above, then for every
 * row BP left unconverged ldpc::osd::OsdDecoder::decode with osd_order 0 (src_cpp/osd.hpp:110-117) =
 * soft_decision_col_sort (sort.hpp:48-62; stable, ties by ascending column) + RowReduce::fast_solve
 * (gf2sparse_linalg.hpp:298-401) + lu_solve (:237-288), all on the device.  `decoding` receives the BP
 * decision for converged rows and the OSD-0 solution otherwise; llr / iterations / converge are BP's
 * (bpd.log_prob_ratios, bpd.iterations, bpd.converge), any of them may be NULL.
 * The bit-packed [H | s] of one syndrome must fit in LDS (m * ceil((n+1)/64) * 8 + 13 n + 4 m <= 150 KiB),
 * else LDPC_HIP_ERR_UNSUPPORTED.  Syndromes outside the image of H: see ldpc_hip_bposd_get_status.
 */
int ldpc_hip_bposd0_decode_batch(ldpc_hip_bp *h, const uint8_t *syndromes, int64_t batch,
                                 uint8_t *decoding, double *llr, int32_t *iterations,
                                 uint8_t *converge);
int ldpc_hip_bposd0_decode_batch_async(ldpc_hip_bp *h, const uint8_t *syndromes, int64_t batch,
                                       uint8_t *decoding, double *llr, int32_t *iterations,
                                       uint8_t *converge);

/*
 * BP + ordered-statistics decoding of any order: replaces BpOsdDecoder.decode's per-shot path for
 * osd_method OSD_E / OSD_CS as well (ldpc::osd::OsdDecoder::decode, src_cpp/osd.hpp:103-187; candidate strings
 * osd.hpp:75-101).  ldpc_hip_bp_set_osd stores the method and order the BpOsdDecoder setters write
 * (_bposd_decoder.pyx:161-234): osd_method 0 = OSD_OFF, 1 = OSD_0, 2 = OSD_E (exhaustive), 3 = OSD_CS
 * (combination sweep) -- the values of ldpc::osd::OsdMethod (osd.hpp:18-23); osd_order >= 0, 0 with OSD_0.
 * ldpc_hip_bposd_decode_batch then post-processes every row BP left unconverged: column sort by log-ratio,
 * reduced row echelon form over that order, the OSD-0 solution, and the sweep over the candidate strings on the
 * non-pivot columns; a candidate replaces the current solution only if its weight sum_j x_j log(1 / p_j)
 * (added up in ascending j, as osd.hpp:171-176 does) is STRICTLY smaller.  osd_order == 0 takes the OSD-0
 * branch whatever the method (osd.hpp:114).  Outputs as for ldpc_hip_bposd0_decode_batch.
 * Limits: the LDS bound of OSD-0 (plus 8 m + 4 n bytes); OSD_E osd_order <= 24 (2^24 candidates per syndrome; the reference
 * itself keeps 2^order strings of k bytes in memory and warns above 15), else LDPC_HIP_ERR_UNSUPPORTED.  OSD_CS takes any
 * osd_order: the k weight-one strings, then the pairs (i, j), i < j < osd_order, among ALL k = n - rank non-pivot columns
 * (pairs whose second index reaches past k are skipped -- the reference writes past its candidate string there, osd.hpp:92-96,
 * undefined behaviour).  Syndromes outside the image of H: see below.
 */
int ldpc_hip_bp_set_osd(ldpc_hip_bp *h, int32_t osd_method, int32_t osd_order);
/*
 * Syndromes outside the image of H.  Every H e is inside; a rank-deficient H (toric-code checks, any H with dependent
 * rows) fed with a syndrome that breaks a dependency is not, and then NO x satisfies H x = s.  The reference still
 * returns a vector: RowReduce::fast_solve never reaches its early stop, eliminates every column, and lu_solve
 * (gf2sparse_linalg.hpp:237-288) solves the equations of the PIVOT ROWS only -- which rows those are is decided by the
 * sparsity heuristic of its linked-list elimination (fewest entries across L and U, ties by the position in the pivot
 * column's linked list, :149-163 / :318-333 -- a position that swap_rows, insert_entry and add_rows leave history-dependent).
 * The bit-packed device eliminations take the first candidate row, which for a syndrome inside the image cannot change the
 * solution.  Rows outside the image are detected (H x != s after the first pass) and go through a second pass: one workgroup
 * per such row re-enacts the reference's elimination on that data structure (csrc/osd_exact_kernel.h) to learn ITS pivot
 * rows, writes the syndrome that agrees with s on those rows and lies in the image, and the same OSD kernels run again on it
 * -- OSD-0, OSD_E and OSD_CS alike.  The output is then the reference's vector, bit for bit (tests/golden/outimage_*.npz,
 * every row), and the row stays flagged: after any ldpc_hip_bposd*_decode_batch call, ldpc_hip_bposd_get_status fills
 * status[batch] (host or device pointer) with
 *   0  BP converged, OSD did not run            1  OSD ran and H x = s
 *   2  OSD ran, s is outside the image of H: x does not (cannot) satisfy H x = s; it solves the reference's pivot rows.
 * `batch` must be the batch size of that decode.  The second pass needs rank H (worked out on the host for m * m * n / 64 < 4e9)
 * and m <= 8192; beyond that a flagged row keeps the solution of the device's own pivot rows.  Both extra launches size
 * themselves from device-side counters: no host round trip, a few microseconds when no row is flagged, nothing at all for
 * a full-rank H.  Device memory of the second pass (per handle, allocated on the first BP + OSD call on a rank-deficient H and
 * kept): the working copies of at most 64 workgroups, (m * ceil(n / 64) + n * m / 4) 64-bit words each, capped at 256 MiB in
 * total, plus batch * (m + 4) bytes for the corrected syndromes and their list; the first such call also pays the host
 * elimination that finds rank H.  If that memory cannot be had the pass is skipped: the decode still succeeds and flagged rows
 * keep the device's own pivot-row solution (status 2).
 */
int ldpc_hip_bposd_get_status(ldpc_hip_bp *h, uint8_t *status, int64_t batch);
/* Serial schedule: a 64-syndrome tile is decoded by one wavefront, which runs until its slowest syndrome is done.  With
 * repacking a first pass of `first_pass_iters` iterations runs over the whole batch and the rows it leaves unconverged are
 * packed into dense tiles and decoded again from the start with the full max_iter (deterministic: same results).
 * -1 = automatic (max_iter / 8, default), 0 = off.  With repacking the call waits once for the device.
 * The STREAMED serial kernels (ldpc_hip_bp_set_serial_kernel: (3,6)-regular-shaped matrices beyond LDS, e.g. the n = 10 000 code)
 * decode in PASSES and never start again: the first pass ends after first_pass_iters iterations (-1 = automatic: 4), the next after
 * twice as many -- or, when a pass left less than 40 % of its rows, after one more iteration.  After a pass the rows still decoding
 * are counted (the call waits for the device once per pass) and either carry on in their tiles (most rows left), or have their
 * message state compacted, lane by lane, into dense tiles, or -- 2048 rows and fewer -- finish a workgroup per syndrome
 * (bp_serial_lane_kernel: the one hopeless syndrome of a batch costs milliseconds instead of keeping a 64-syndrome tile on one
 * compute unit for max_iter iterations).  0 = one pass.  Same results whatever the passes (tests/test_gpu_serial_stream.py).
 * The streamed parallel schedule (codes too large for the on-chip kernels, batches of >= 32768 syndromes) cuts a decode in two
 * as well, but CARRIES ON instead of starting again: after the first pass the message state of the unconverged rows is
 * gathered, lane by lane, out of the first pass's 64-syndrome tiles into dense tiles and the second pass continues from
 * iteration first_pass_iters + 1 (a tile moves all 64 lanes' messages until its slowest syndrome is done; after the
 * compaction the tiles hold live lanes only).  There "automatic" prices a cut at every iteration against the plain run with
 * the iteration histogram the previous decode on the handle left behind, and runs plain when that says so or says nothing
 * yet -- no work is wasted where nothing converges.  Nothing of this waits for the device (the *_async entry points stay
 * asynchronous): the histogram is used only once its copy is SEEN to have landed (a look at an event, never a wait -- decodes
 * queued back to back are steered by the last one that did land), and the second pass is queued at once for the most rows
 * there can be: the device lists the unconverged rows, counts them, and every kernel of the second pass reads that count and
 * reaches the caller's arrays through the list (no rows are copied out and back; no extra message memory: the compacted state
 * is gathered into the first pass's check_to_bit array, which is dead by then).  A batch that does not fit in one chunk of
 * device memory runs plain. */
int ldpc_hip_bp_set_repack(ldpc_hip_bp *h, int32_t first_pass_iters);
/* Serial schedule kernels: bits that share no check commute, so the schedule is cut into levels of mutually check-disjoint
 * bits (level = 1 + the highest level among the EARLIER bits sharing a check) and a workgroup runs a tile level by level
 * with its wavefronts sharing each level's bits -- same results as the bit-by-bit walk.  -1 = automatic (level-parallel
 * when a level holds >= 2 bits on average; STREAMED -- below -- where the matrix allows it and a level holds >= 32 bits),
 * 0 = one wavefront walks the tile bit by bit, 1 = always level-parallel (never streamed), 2 = streamed whenever the matrix allows
 * it.  Streamed (csrc/bp_serial_stream_kernel.h; matrices with one row weight 6 and one column weight 3): the level-parallel
 * form with the message traffic of the flooding kernel -- one record per position of the level-major order (the 15 segments to
 * fetch, the 3 to write, the bit) read through the scalar cache two steps ahead, the segments of the next position in flight into
 * a per-wavefront LDS ring (buffer_load ... lds, counted waits), no initial messages in memory (entries nobody has written yet
 * are the same in all lanes: a per-position table), a decode in passes (ldpc_hip_bp_set_repack) and, for a handful of syndromes
 * (what a pass leaves, or a batch of <= 256), a workgroup per syndrome with the level's bits across the lanes.  The n = 10 000
 * code at p = 0.05, B = 65 536: 171 k -> 692 k syndromes/s; its first pass moves 5.1 TB/s (profiles/r5_serial_stream_summary.txt). */
int ldpc_hip_bp_set_serial_kernel(ldpc_hip_bp *h, int32_t mode);
/* Where the elimination keeps [H | s]: -1 = automatic (in the wavefront's registers when m <= 256 and n <= 511, else
 * one wavefront per syndrome with [H | s] bit-packed in LDS while four of them fit a CU, else one workgroup per syndrome
 * with H in LDS or, beyond 150 KiB, in an HBM scratch slot -- as long as the column order and, for OSD_E / OSD_CS, the
 * candidate tables fit LDS: OSD-0 n <= ~14 000; otherwise LDPC_HIP_ERR_UNSUPPORTED), 0 = the one-wavefront LDS kernels
 * while they fit at all, 2 = a workgroup per syndrome with H in HBM whatever the size.  Results are identical. */
int ldpc_hip_bp_set_osd_kernel(ldpc_hip_bp *h, int32_t mode);
int ldpc_hip_bposd_decode_batch(ldpc_hip_bp *h, const uint8_t *syndromes, int64_t batch,
                                uint8_t *decoding, double *llr, int32_t *iterations,
                                uint8_t *converge);
int ldpc_hip_bposd_decode_batch_async(ldpc_hip_bp *h, const uint8_t *syndromes, int64_t batch,
                                      uint8_t *decoding, double *llr, int32_t *iterations,
                                      uint8_t *converge);

/*
 * replaces: ldpc::bp::BpDecoder::soft_info_decode_serial (src_cpp/bp.hpp:547-660), the method behind
 * SoftInfoBpDecoder.decode (_bp_decoder.pyx:712-785), over a batch of analog syndromes.
 * soft_syndromes [batch][m] FP64 readouts; each is scaled to 2 s / sigma^2, its sign gives the hard syndrome, and the
 * serial minimum-sum sweep treats a check whose scaled magnitude is below `cutoff` (and below the smallest incoming
 * message) as a virtual variable node that may be updated or flipped (bp.hpp:597-621).  Uses the handle's
 * max_iter, ms_scaling_factor (taken literally, no adaptive mode), channel probabilities and -- if one was set with
 * ldpc_hip_bp_set_schedule(serial, order) -- serial_schedule_order; bp_method and schedule are ignored (the reference
 * class forces minimum_sum / serial, pyx:751-752).  Outputs: decoding [batch][n]; llr, iterations, converge as for
 * decode_batch; soft_syndromes_out [batch][m] = the decoder's soft_syndrome member afterwards (pyx:787-798).  Any
 * output but decoding may be NULL.  Host or device pointers.  m <= 19200 (one word per check in LDS).
 */
int ldpc_hip_bp_soft_info_decode_batch(ldpc_hip_bp *h, const double *soft_syndromes, int64_t batch, double cutoff,
                                       double sigma, uint8_t *decoding, double *llr, int32_t *iterations,
                                       uint8_t *converge, double *soft_syndromes_out);

/*
 * Bit-packed shot data in, bit-packed predictions out: replaces the per-shot loop of the reference's sinter decoders
 * (sinter_decoders/sinter_bposd_decoder.py:115-130, sinter_belief_find_decoder.py; `decode_via_files` reads detection
 * events and writes observable predictions in stim's "b8" format: bit i of a shot is bit i % 8 of its byte i / 8, every
 * shot starts on a byte boundary).
 * ldpc_hip_bp_set_observables uploads the k x n observables matrix L (CSR, as `matrices.observables_matrix`).
 * ldpc_hip_bp_decode_b8: dets_b8 [batch][ceil(m/8)] -> decode every shot (BP; with_osd != 0: BP + the handle's OSD
 * method, see ldpc_hip_bp_set_osd) -> obs_b8 [batch][ceil(k/8)] = L x mod 2 (`(observables_matrix @ corr) % 2`) and/or
 * decoding_b8 [batch][ceil(n/8)] = x.  Either output may be NULL; iterations / converge as for decode_batch (NULL
 * allowed).  All-zero shots yield the zero correction, converge 1, iterations 0 (the Python-level shortcut,
 * _bposd_decoder.pyx:118-123).  Host or device pointers.
 */
int ldpc_hip_bp_set_observables(ldpc_hip_bp *h, int32_t k, const int32_t *csr_row_ptr, const int32_t *csr_col_idx);
/* [batch][bits] one byte per bit <-> [batch][ceil(bits/8)] b8 rows, device pointers only, queued on the handle's
 * stream without synchronisation.  For packing decodings before they cross a link (the multi-GPU gather of
 * bench.py / ldpc_amd.sharding sends 1/8 of the bytes this way). */
int ldpc_hip_pack_b8(ldpc_hip_bp *h, const uint8_t *bytes, int64_t batch, int32_t bits, uint8_t *packed);
int ldpc_hip_unpack_b8(ldpc_hip_bp *h, const uint8_t *packed, int64_t batch, int32_t bits, uint8_t *bytes);
int ldpc_hip_bp_decode_b8(ldpc_hip_bp *h, const uint8_t *dets_b8, int64_t batch, int32_t with_osd, uint8_t *obs_b8,
                          uint8_t *decoding_b8, int32_t *iterations, uint8_t *converge);

/*
 * replaces: GF2Sparse::mulvec (gf2sparse.hpp:177-214) over a batch:
 * out[b][i] = XOR_{j in row i} in[b][j].  Used by received-vector mode (bp.hpp:162-180).
 */
int ldpc_hip_gf2_mulvec_batch(ldpc_hip_bp *h, const uint8_t *vectors, int64_t batch,
                              uint8_t *out_syndromes);

/*
 * Synthetic input generator for benchmarks: independent BSC errors
 * e[b][j] = ((splitmix64(seed, (shot0+b)*n + j) >> 11) < threshold)  (noise model of
 * noise_models/bsc.py:4-23 with a counter-based stream) and syndromes = H e mod 2
 * (mcs.py:124-126).  `errors` may be NULL.  Device or host output pointers.
 */
int ldpc_hip_gen_bsc_syndromes(ldpc_hip_bp *h, uint64_t seed, uint64_t threshold, int64_t shot0,
                               int64_t batch, uint8_t *syndromes, uint8_t *errors);

/* Duration in milliseconds of the BP kernel launch of the last decode call on this handle,
 * measured with HIP events on the launch stream (0 if none yet).  Also 0 after a call of at most four syndromes with host buffers
 * on a small code (a single decode()): there the two event packets would cost more than they tell -- debug switch
 * "TIME_SMALL_CALLS" = 1 keeps them.  After a large call with host buffers (pinned chunks, see ldpc_hip_bp_decode_batch) it
 * describes the LAST chunk only. */
int ldpc_hip_bp_last_kernel_ms(ldpc_hip_bp *h, float *ms);
/* Split of that time for the streaming path: the persistent kernel (bp_decode_kernel) and the per-pass launches
 * (bp_spread_*_kernel: small batches from the first iteration, stragglers of large ones).  Both 0 for other kernels. */
int ldpc_hip_bp_last_phase_ms(ldpc_hip_bp *h, float *persistent_ms, float *per_pass_ms);

/* Bytes of device workspace a decode of `batch` syndromes needs (message arrays dominate:
 * 2 * 8 * nnz bytes per syndrome). */
int64_t ldpc_hip_bp_workspace_bytes(const ldpc_hip_bp *h, int64_t batch);

/* Tuning knobs (0 = library default): wavefronts per workgroup of the BP kernel (1..16) and the
 * largest number of 64-syndrome tiles decoded per launch (bounds the workspace). */
int ldpc_hip_bp_set_tuning(ldpc_hip_bp *h, int32_t waves_per_workgroup, int32_t max_chunk_tiles);

/*
 * How the product-sum update evaluates std::tanh / std::log of bp.hpp:208-217 on the device:
 *   LDPC_HIP_MATH_LIBM_EXACT (default): re-implementations of the host glibc's tanh (fdlibm expm1
 *       based) and log (table driven, FMA build) that are bit-identical to them, so log_prob_ratios
 *       equal the reference's to the last bit (as min-sum's always do);
 *   LDPC_HIP_MATH_FAST: ~1-ulp routines, ~40 % fewer FP64 operations; hard decisions identical
 *       except on exact ties of the reference's posterior, LLRs within the 1e-5 relative tolerance.
 */
/* For matrices whose rows all have one weight and whose columns all have one weight the BP kernel
 * streams messages through a per-wavefront LDS ring filled by asynchronous global->LDS loads
 * (default: 2 slots per wavefront).  depth 0 forces the register-prefetch variant used for irregular
 * matrices, 2 or 3 select the ring depth (1 = default); results are identical in every case.  (Irregular matrices with rows of
 * <= 16 and columns of <= 8 entries have a ring of their own -- a queue of 1 KiB units, debug switch "VAR_RING" 1 -- which is built and
 * tested but not selected by default: see ldpc_hip_bp_set_handoff.) */
int ldpc_hip_bp_set_ring(ldpc_hip_bp *h, int32_t depth);

/* Straggler hand-off of the streaming BP kernel.  A 64-syndrome tile is decoded by one persistent workgroup,
 * i.e. on one compute unit; when no more than `threshold_tiles` tiles are still running (a few syndromes that
 * refuse to converge, a tiny batch, or the reference's default max_iter = n) those tiles park their state and
 * their remaining iterations run as per-pass launches (check pass, bit pass, syndrome test, bookkeeping) spread
 * over the whole chip; a batch of no more than `threshold_tiles` tiles runs that way from the first iteration.
 * -1 = automatic (default): 256 tiles -- except for product-sum on a matrix that has no fixed-degree ring variant (irregular codes; rows <= 16,
 * columns <= 8 entries), where every batch runs as per-pass launches from its first iteration (the persistent kernel's single register
 * allocation leaves three wavefronts per SIMD there, the per-pass kernels run 4 - 8: 0.45 -> 0.56 of HBM on the irregular n = 10 000 code);
 * 0 = off; any value set here is taken as given.  Results are identical.  The hand-off costs no host synchronisation. */
int ldpc_hip_bp_set_handoff(ldpc_hip_bp *h, int32_t threshold_tiles);

/* Codes whose two message arrays fit in a few KiB per syndrome (surface codes, bivariate-bicycle codes)
 * are decoded by on-chip kernels: messages live in LDS and a syndrome that converges is replaced at once.
 * Two variants: one wavefront per syndrome with no workgroup barrier in the loop (row and column weights <= 8),
 * and a workgroup that keeps up to four syndromes resident (any degrees).  mode -1 = automatic (default),
 * 0 = always use the streaming kernel, 1 = use an on-chip kernel whenever one syndrome fits in LDS,
 * 2 = as 1 but only the workgroup ("slot") variant, 3 = as 1 but only the lane = node wavefront variant (product-sum
 * otherwise prefers a lane = entry variant that keeps every lane busy with one transcendental).  That variant gives a syndrome
 * one wavefront, or -- where LDS leaves room for fewer than six syndromes per compute unit, e.g. a 768 x 1600 matrix -- all the
 * wavefronts of a workgroup (its "team" form); 4 = as 3, one wavefront per syndrome, 5 = as 3, a workgroup per syndrome.
 * Min-sum on matrices with rows of weight <= 4 and columns of weight 1 .. 2 (rotated / toric surface codes, ring codes; m <= 256)
 * takes a third variant in modes -1, 1 and 6: lane = EDGE, a row's entries in four neighbouring lanes, every message in a
 * register for the whole decode (bp_edge_kernel.h: bp_edge_kernel); with rows of weight <= 8 and columns of weight <= 4
 * (8 m <= 768: the bivariate-bicycle family) the same idea with a row in EIGHT neighbouring lanes (bp_edge8_kernel).
 * 6 = those variants where they apply, else as -1.  The on-chip kernels take batches below 2^30 syndromes (larger ones
 * are streamed).  Results are identical. */
int ldpc_hip_bp_set_small_code_kernel(ldpc_hip_bp *h, int32_t mode);
/* Measurement / test switches: kernel-shape choices that never change a result (profiles/README.md lists them: "PS_TEAM",
 * "OSD_UNBLOCKED", "OSD_PLANES", "TEAM_WAVES", "EDGE_STATIC_PCT", "EDGE_CHUNK", ...; for schedule = serial_relative: "REL_LDS" 0 = the
 * per-lane kernel with the state in HBM, 16 / 64 = lanes per syndrome of the on-chip kernel; "REL_LEVELS" 0 = sweep bit by bit instead of
 * level by level; "REL_SCRATCH_IN_L" 0 = scratch apart from the posterior array; "REL_PROF" 1 = print the kernel's cycle shares per phase
 * to stderr; for the streamed serial schedule: "SER_WAVES" / "SER_RING" wavefronts per tile and ring depth, "SER_LANE_MAX" rows at or below
 * which what a pass left finishes a workgroup per syndrome (0 = never), "SER_LANE_THREADS"; "NO_SPREAD_COMPACT" 1 = per-pass rounds never
 * compact their list of tiles; "VAR_RING" 1 = the streamed kernel's variable-degree LDS ring wherever it applies (rows <= 16, columns <= 8
 * entries), "VAR_RING_UNITS" its KiB of LDS per wavefront (8 .. 40, default 11); "SPREAD_NODES" / "SPREAD_NODES2" rows per wavefront of the
 * per-pass kernels from the start / after a hand-off; "SER_VAR" 0 = the streamed serial schedule's item form (any degree profile) never, 1 = also
 * on (6,3)-regular matrices, "SER_VAR_UNITS" its KiB of LDS per wavefront (default 8); "REL_EXT" 0 / 1 = serial_relative's on-chip kernel never / always with the messages and
 * per-entry records in global memory (default: where the all-in-LDS form leaves fewer than four wavefronts per compute unit); "REL_FIRST_ONCE" 0 = every syndrome re-enacts the first
 * iteration's sort itself instead of taking the call's (rel_first_order_kernel); "REPACK2" 1 .. 3 = the two-pass streamed decode compacts a second time, after that many
 * iterations of its second pass, 4 = where the iteration histogram suggests it (default: never -- measured no gain); "OSD_COLLECT_AFTER" 1 = BP + OSD lists the
 * rows BP left unconverged in a launch of its own after the BP kernel instead of inside the on-chip BP kernels; "OSD_NO_FLAT" 1 = OSD-0 on small matrices without the column permutation (osd0_reg_kernel instead of osd0_flat_kernel); "EDGE_CLAMP" 1 = the lane = edge min-sum kernel always
 * with its clamp to DBL_MAX (default: left out where it provably never bites)).  A handle reads the environment variables LDPC_HIP_<NAME> ONCE, when it is
 * created; afterwards only this call changes a switch (value < 0: back to "not set").  Unknown names are an error. */
int ldpc_hip_bp_set_debug_switch(ldpc_hip_bp *h, const char *name, int32_t value);

#define LDPC_HIP_MATH_LIBM_EXACT 0
#define LDPC_HIP_MATH_FAST 1
int ldpc_hip_bp_set_math(ldpc_hip_bp *h, int32_t math_mode);

/* The shader clock the BP kernels actually ran at (a measurement aid; no counterpart in the reference).  Every workgroup of the
 * long-running BP kernels (bp_decode_kernel, bp_wave_kernel, bp_wave_ps_kernel, bp_edge_kernel, bp_edge8_kernel) adds the shader cycles
 * (s_memtime) and the constant-rate ticks (s_memrealtime) of its lifetime to two 64-bit words of the handle; the words only grow.
 * This call waits for the handle's stream and returns them together with the tick rate (hipDeviceAttributeWallClockRate).
 * Clock over an interval = (cycles_after - cycles_before) / (ticks_after - ticks_before) * tick_hz, averaged over the kernels'
 * workgroups and weighted by their lifetime. */
int ldpc_hip_bp_clock_probe(ldpc_hip_bp *h, uint64_t *cycles, uint64_t *ticks, double *tick_hz);

/* The copy rate of THIS box, now (a measurement aid; no counterpart in the reference).  Copies `tiles` x `segments_per_tile` segments of
 * 512 bytes (64 syndromes x one double: the unit every streamed BP kernel moves) from one of the handle's message arrays to the other and
 * back, `passes` times, one workgroup per tile, non-temporal, on the handle's stream, timed with events on that stream.
 * gbytes_per_s = bytes read + bytes written per second.  bench.py prints it beside the roofline fraction so that one JSON line tells a slow
 * box from a slow build: boxes of the pool differ by ~10 % in what they give ANY kernel that moves these bytes once.  The handle's
 * message arrays are (re)allocated to the probe's size if they are smaller; their contents are scratch between decodes anyway. */
int ldpc_hip_bp_copy_probe(ldpc_hip_bp *h, int64_t tiles, int32_t segments_per_tile, int32_t passes, float *ms, double *gbytes_per_s);

/* Page-locked host memory for a caller's result arrays (no counterpart in the reference: its arrays never leave the host).  A host
 * pointer into such a block -- or into memory the caller registered with hipHostRegister -- handed to ldpc_hip_bp_decode_batch as `llr`
 * is written by the device-to-host copies themselves: the pipelined host path (above) skips its pinned staging buffer and the
 * host-side copy for that array.  At 65 536 x 10 000 that is 5.2 GB less to move per call.  NULL when the memory cannot be had
 * (page-locked memory is a limited resource): fall back to ordinary memory. */
void *ldpc_hip_host_alloc(size_t bytes);
void ldpc_hip_host_free(void *p);

const char *ldpc_hip_last_error(void);
const char *ldpc_hip_version(void);

/*
 * ---- several GPUs, one process --------------------------------------------------------------------------------
 * "device_ids[ndev] ... the library owns device memory, streams and the inter-GPU traffic
 * inside the handle".  ldpc_hip_bp_multi_create builds one ordinary handle per entry of device_ids (a device may be
 * listed more than once: two handles then share that GPU); ldpc_hip_bp_multi_decode_batch cuts the batch into
 * contiguous row ranges on 64-syndrome boundaries, decodes every range on its GPU concurrently (one host thread per
 * GPU for the duration of the call) and returns when all results are in the caller's buffers -- row b of every output
 * is what the single-GPU call gives for row b, bit for bit.  Buffers: all host memory (every GPU copies its own rows
 * over its own PCIe link), or all on ONE GPU (the others receive their rows by peer copy over xGMI and send their hard
 * decisions back bit-packed, 1/8 of the bytes, to be unpacked on that GPU when it is one of device_ids).
 * with_osd: -1 = ldpc_hip_bp_decode_batch, 0 = ldpc_hip_bposd0_decode_batch, 1 = ldpc_hip_bposd_decode_batch.
 * Settings (channel, parameters, schedule, OSD, math ...) are applied per GPU through ldpc_hip_bp_multi_handle(mh, i),
 * i < ldpc_hip_bp_multi_devices(mh): the setters above, unchanged.  The schedules that keep state in the decoder object
 * (serial_relative, the random serial schedule) have ONE state per multi handle: a call starts every row from the state of
 * handle 0 (copied to the others first -- which also settles random_schedule_seed = 0, the clock) and leaves the state of the
 * batch's last row on EVERY handle, so that sequences of calls equal the single-GPU sequences bit for bit.  Soft-syndrome
 * decoding (ldpc_hip_bp_soft_info_decode_batch) is a per-handle call: use ldpc_hip_bp_multi_handle(mh, 0).  Multi-PROCESS sharding (one rank per GPU, RCCL
 * gather of the packed rows) is ldpc_amd/sharding.py + bench.py; the reference has neither (bp.hpp:129-140).
 */
typedef struct ldpc_hip_bp_multi ldpc_hip_bp_multi;
int ldpc_hip_bp_multi_create(const ldpc_hip_bp_desc *desc /* desc->device is ignored */, const int32_t *device_ids,
                             int32_t ndev, ldpc_hip_bp_multi **out);
void ldpc_hip_bp_multi_destroy(ldpc_hip_bp_multi *mh);
int32_t ldpc_hip_bp_multi_devices(const ldpc_hip_bp_multi *mh);
ldpc_hip_bp *ldpc_hip_bp_multi_handle(ldpc_hip_bp_multi *mh, int32_t i);
int ldpc_hip_bp_multi_decode_batch(ldpc_hip_bp_multi *mh, int32_t with_osd, const uint8_t *syndromes, int64_t batch,
                                   uint8_t *decoding, double *llr, int32_t *iterations, uint8_t *converge);
/* BP kernel time of the last decode on every GPU (ldpc_hip_bp_last_kernel_ms per handle), ms_per_device[ndev] */
int ldpc_hip_bp_multi_last_kernel_ms(ldpc_hip_bp_multi *mh, float *ms_per_device);
/* Testing aid: force != 0 routes device buffers through the peer-copy / bit-packed path even on the GPU they live on
 * (so that the path can be exercised on a one-GPU machine, e.g. with device_ids = {0, 0}). */
int ldpc_hip_bp_multi_set_staging(ldpc_hip_bp_multi *mh, int32_t force);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H */
