/*
 * oracle/ref_harness.cpp -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A thin extern "C" shim around the REAL reference decoder: it #includes the reference's own
 * header-only sources from where they lie under /root/reference/src_cpp (nothing is copied into
 * this repository) and exposes ldpc::bp::BpDecoder::decode through plain pointers so Python can
 * call it with ctypes.  Built only in the build container by oracle/Makefile into
 * oracle/_ref/libref_bp.so (git-ignored; the GPU box only ever sees the prebuilt .so).
 *
 * Used to (i) prove oracle/bp_oracle.c bit-exact and (ii) generate tests/golden/*.npz
 * (tests/golden/make_golden.py).  `cstdint` must precede bp.hpp.
 */
#include <cstdint>
#include <vector>
#include <cstring>
#include "bp.hpp"

using ldpc::bp::BpDecoder;
using ldpc::bp::BpSparse;

struct ref_bp {
    BpSparse *pcm;
    BpDecoder *dec;
};

extern "C" {

/* rows/cols: coordinates of the nnz ones of H, any order (insert_entry keeps lists sorted). */
ref_bp *ref_bp_new(int m, int n, int nnz, const int32_t *rows, const int32_t *cols,
                   const double *channel_probs, int max_iter, int bp_method, int schedule,
                   double ms_scaling_factor, int input_type) {
    auto *r = new ref_bp;
    r->pcm = new BpSparse(m, n, nnz);
    for (int k = 0; k < nnz; k++) r->pcm->insert_entry(rows[k], cols[k]);
    std::vector<double> probs(channel_probs, channel_probs + n);
    r->dec = new BpDecoder(*r->pcm, probs, max_iter, static_cast<ldpc::bp::BpMethod>(bp_method),
                           static_cast<ldpc::bp::BpSchedule>(schedule), ms_scaling_factor, 1,
                           ldpc::bp::NULL_INT_VECTOR, 0, false,
                           static_cast<ldpc::bp::BpInputType>(input_type));
    return r;
}

void ref_bp_free(ref_bp *r) {
    if (!r) return;
    delete r->dec;
    delete r->pcm;
    delete r;
}

void ref_bp_set_channel(ref_bp *r, const double *channel_probs) {
    for (int j = 0; j < r->dec->bit_count; j++) r->dec->channel_probabilities[j] = channel_probs[j];
}

/* serial_schedule_order (bp.hpp:67; the Cython setter writes it element-wise, _bp_decoder.pyx:465-483) */
void ref_bp_set_serial_order(ref_bp *r, const int32_t *order) {
    r->dec->serial_schedule_order.resize((size_t)r->dec->bit_count);
    for (int j = 0; j < r->dec->bit_count; j++) r->dec->serial_schedule_order[(size_t)j] = order[j];
}

/* One BpDecoder::decode call (bp.hpp:159-190); `len` is m (syndrome) or n (received vector). */
void ref_bp_decode(ref_bp *r, const uint8_t *input, int len, uint8_t *decoding, double *llr,
                   int32_t *iterations, uint8_t *converge) {
    std::vector<uint8_t> in(input, input + len);
    r->dec->decode(in);
    const int n = r->dec->bit_count;
    std::memcpy(decoding, r->dec->decoding.data(), (size_t)n);
    if (llr) std::memcpy(llr, r->dec->log_prob_ratios.data(), sizeof(double) * (size_t)n);
    *iterations = r->dec->iterations;
    *converge = r->dec->converge ? 1 : 0;
}

void ref_bp_decode_batch(ref_bp *r, const uint8_t *inputs, int len, int64_t shots,
                         uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge) {
    const int n = r->dec->bit_count;
    for (int64_t b = 0; b < shots; b++)
        ref_bp_decode(r, inputs + b * len, len, decodings + b * n, llr ? llr + b * n : nullptr,
                      iterations + b, converge + b);
}

/* BpDecoder::soft_info_decode_serial (bp.hpp:547-660) as SoftInfoBpDecoder.decode drives it (_bp_decoder.pyx:761-785);
 * the decoder must have been created with schedule SERIAL (0) and bp_method MINIMUM_SUM (1), pyx:751-752 */
void ref_bp_soft_info_decode_batch(ref_bp *r, const double *soft_syndromes, int64_t shots, double cutoff, double sigma,
                                   uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge,
                                   double *soft_syndromes_out) {
    const int m = r->dec->check_count, n = r->dec->bit_count;
    for (int64_t b = 0; b < shots; b++) {
        std::vector<double> s(soft_syndromes + b * m, soft_syndromes + (b + 1) * m);
        r->dec->soft_info_decode_serial(s, cutoff, sigma);
        std::memcpy(decodings + b * n, r->dec->decoding.data(), (size_t)n);
        if (llr) std::memcpy(llr + b * n, r->dec->log_prob_ratios.data(), sizeof(double) * (size_t)n);
        if (soft_syndromes_out) std::memcpy(soft_syndromes_out + b * m, r->dec->soft_syndrome.data(), sizeof(double) * (size_t)m);
        iterations[b] = r->dec->iterations;
        converge[b] = r->dec->converge ? 1 : 0;
    }
}

/* A NEW decoder object per row (what "row b of a batch" means for the schedules that keep state in the object: the
 * serial_relative order, the random schedule's order and generator): bp.hpp:77-132 constructor with the given schedule
 * (0 serial, 2 serial_relative), random_serial_schedule and random_schedule_seed, then ONE decode. */
void ref_bp_decode_fresh_batch(int m, int n, int nnz, const int32_t *rows, const int32_t *cols, const double *channel_probs,
                               int max_iter, int bp_method, int schedule, double ms_scaling_factor, int random_serial,
                               int random_seed, const uint8_t *inputs, int64_t shots, uint8_t *decodings, double *llr,
                               int32_t *iterations, uint8_t *converge, int32_t *final_order) {
    BpSparse pcm(m, n, nnz);
    for (int k = 0; k < nnz; k++) pcm.insert_entry(rows[k], cols[k]);
    std::vector<double> probs(channel_probs, channel_probs + n);
    for (int64_t b = 0; b < shots; b++) {
        BpDecoder dec(pcm, probs, max_iter, static_cast<ldpc::bp::BpMethod>(bp_method), static_cast<ldpc::bp::BpSchedule>(schedule),
                      ms_scaling_factor, 1, ldpc::bp::NULL_INT_VECTOR, random_seed, random_serial != 0, ldpc::bp::SYNDROME);
        std::vector<uint8_t> in(inputs + b * m, inputs + (b + 1) * m);
        dec.decode(in);
        std::memcpy(decodings + b * n, dec.decoding.data(), (size_t)n);
        if (llr) std::memcpy(llr + b * n, dec.log_prob_ratios.data(), sizeof(double) * (size_t)n);
        iterations[b] = dec.iterations;
        converge[b] = dec.converge ? 1 : 0;
        if (final_order) for (int j = 0; j < n; j++) final_order[b * n + j] = dec.serial_schedule_order[(size_t)j];
    }
}

/* The same with a serial_schedule_order given to the constructor (bp.hpp:85, 110-111: any n bit numbers, taken as they are): a new
 * object per row (carried == 0) or one object for all rows. */
void ref_bp_decode_batch_order(int m, int n, int nnz, const int32_t *rows, const int32_t *cols, const double *channel_probs,
                               int max_iter, int bp_method, int schedule, double ms_scaling_factor, const int32_t *order0, int carried,
                               const uint8_t *inputs, int64_t shots, uint8_t *decodings, double *llr, int32_t *iterations,
                               uint8_t *converge, int32_t *final_order) {
    BpSparse pcm(m, n, nnz);
    for (int k = 0; k < nnz; k++) pcm.insert_entry(rows[k], cols[k]);
    std::vector<double> probs(channel_probs, channel_probs + n);
    std::vector<int> start(order0, order0 + n);
    BpDecoder *one = nullptr;
    if (carried)
        one = new BpDecoder(pcm, probs, max_iter, static_cast<ldpc::bp::BpMethod>(bp_method), static_cast<ldpc::bp::BpSchedule>(schedule),
                            ms_scaling_factor, 1, start, 0, false, ldpc::bp::SYNDROME);
    for (int64_t b = 0; b < shots; b++) {
        BpDecoder *dec = one ? one
                             : new BpDecoder(pcm, probs, max_iter, static_cast<ldpc::bp::BpMethod>(bp_method),
                                             static_cast<ldpc::bp::BpSchedule>(schedule), ms_scaling_factor, 1, start, 0, false, ldpc::bp::SYNDROME);
        std::vector<uint8_t> in(inputs + b * m, inputs + (b + 1) * m);
        dec->decode(in);
        std::memcpy(decodings + b * n, dec->decoding.data(), (size_t)n);
        if (llr) std::memcpy(llr + b * n, dec->log_prob_ratios.data(), sizeof(double) * (size_t)n);
        iterations[b] = dec->iterations;
        converge[b] = dec->converge ? 1 : 0;
        if (final_order) for (int j = 0; j < n; j++) final_order[b * n + j] = dec->serial_schedule_order[(size_t)j];
        if (!one) delete dec;
    }
    delete one;
}

/* ONE decoder object for all rows, as a loop of BpDecoder.decode calls on one Python object: the order (and generator) carry over */
void ref_bp_decode_carried_batch(int m, int n, int nnz, const int32_t *rows, const int32_t *cols, const double *channel_probs,
                                 int max_iter, int bp_method, int schedule, double ms_scaling_factor, int random_serial,
                                 int random_seed, const uint8_t *inputs, int64_t shots, uint8_t *decodings, double *llr,
                                 int32_t *iterations, uint8_t *converge, int32_t *final_order) {
    BpSparse pcm(m, n, nnz);
    for (int k = 0; k < nnz; k++) pcm.insert_entry(rows[k], cols[k]);
    std::vector<double> probs(channel_probs, channel_probs + n);
    BpDecoder dec(pcm, probs, max_iter, static_cast<ldpc::bp::BpMethod>(bp_method), static_cast<ldpc::bp::BpSchedule>(schedule),
                  ms_scaling_factor, 1, ldpc::bp::NULL_INT_VECTOR, random_seed, random_serial != 0, ldpc::bp::SYNDROME);
    for (int64_t b = 0; b < shots; b++) {
        std::vector<uint8_t> in(inputs + b * m, inputs + (b + 1) * m);
        dec.decode(in);
        std::memcpy(decodings + b * n, dec.decoding.data(), (size_t)n);
        if (llr) std::memcpy(llr + b * n, dec.log_prob_ratios.data(), sizeof(double) * (size_t)n);
        iterations[b] = dec.iterations;
        converge[b] = dec.converge ? 1 : 0;
        if (final_order) for (int j = 0; j < n; j++) final_order[b * n + j] = dec.serial_schedule_order[(size_t)j];
    }
}

/* soft_info_decode_serial with random_serial_schedule (bp.hpp:573-577): a new decoder object per row (`fresh`) or one object
 * for all rows (the order carries over).  Schedule SERIAL, MINIMUM_SUM as SoftInfoBpDecoder sets them (pyx:751-752). */
void ref_bp_soft_random_batch(int m, int n, int nnz, const int32_t *rows, const int32_t *cols, const double *channel_probs,
                              int max_iter, double ms_scaling_factor, int random_seed, int fresh, const double *soft_syndromes,
                              int64_t shots, double cutoff, double sigma, uint8_t *decodings, double *llr, int32_t *iterations,
                              uint8_t *converge, double *soft_out, int32_t *final_order) {
    BpSparse pcm(m, n, nnz);
    for (int k = 0; k < nnz; k++) pcm.insert_entry(rows[k], cols[k]);
    std::vector<double> probs(channel_probs, channel_probs + n);
    BpDecoder *dec = nullptr;
    for (int64_t b = 0; b < shots; b++) {
        if (fresh || !dec) {
            delete dec;
            dec = new BpDecoder(pcm, probs, max_iter, ldpc::bp::MINIMUM_SUM, ldpc::bp::SERIAL, ms_scaling_factor, 1,
                                ldpc::bp::NULL_INT_VECTOR, random_seed, true, ldpc::bp::SYNDROME);
        }
        std::vector<double> s(soft_syndromes + b * m, soft_syndromes + (b + 1) * m);
        dec->soft_info_decode_serial(s, cutoff, sigma);
        std::memcpy(decodings + b * n, dec->decoding.data(), (size_t)n);
        if (llr) std::memcpy(llr + b * n, dec->log_prob_ratios.data(), sizeof(double) * (size_t)n);
        if (soft_out) std::memcpy(soft_out + b * m, dec->soft_syndrome.data(), sizeof(double) * (size_t)m);
        iterations[b] = dec->iterations;
        converge[b] = dec->converge ? 1 : 0;
        if (final_order) for (int j = 0; j < n; j++) final_order[b * n + j] = dec->serial_schedule_order[(size_t)j];
    }
    delete dec;
}

/* GF2Sparse::mulvec (gf2sparse.hpp:177-214) */
void ref_bp_mulvec(ref_bp *r, const uint8_t *in, uint8_t *out) {
    std::vector<uint8_t> v(in, in + r->pcm->n);
    auto s = r->pcm->mulvec(v);
    std::memcpy(out, s.data(), (size_t)r->pcm->m);
}

} /* extern "C" */

/* ---- BP + OSD-0: the real ldpc::osd::OsdDecoder (osd.hpp:26-191) driven as BpOsdDecoder.decode does ---- */
#include "osd.hpp"

struct ref_bposd {
    ref_bp *bp;
    ldpc::osd::OsdDecoder *osd;
};

extern "C" {

ref_bposd *ref_bposd_new2(int m, int n, int nnz, const int32_t *rows, const int32_t *cols,
                          const double *channel_probs, int max_iter, int bp_method, double ms_scaling_factor, int schedule) {
    auto *r = new ref_bposd;
    r->bp = ref_bp_new(m, n, nnz, rows, cols, channel_probs, max_iter, bp_method, schedule, ms_scaling_factor, 0 /*SYNDROME*/);
    r->osd = new ldpc::osd::OsdDecoder(*r->bp->pcm, ldpc::osd::OSD_0, 0, r->bp->dec->channel_probabilities);
    return r;
}

ref_bposd *ref_bposd_new(int m, int n, int nnz, const int32_t *rows, const int32_t *cols,
                         const double *channel_probs, int max_iter, int bp_method, double ms_scaling_factor) {
    auto *r = new ref_bposd;
    r->bp = ref_bp_new(m, n, nnz, rows, cols, channel_probs, max_iter, bp_method, 1 /*PARALLEL*/, ms_scaling_factor, 0 /*SYNDROME*/);
    r->osd = new ldpc::osd::OsdDecoder(*r->bp->pcm, ldpc::osd::OSD_0, 0, r->bp->dec->channel_probabilities);
    return r;
}

void ref_bposd_free(ref_bposd *r) {
    if (!r) return;
    delete r->osd;
    ref_bp_free(r->bp);
    delete r;
}

/* _bposd_decoder.pyx:125-134 without the Python-side zero-syndrome shortcut */
void ref_bposd_decode_batch(ref_bposd *r, const uint8_t *syndromes, int64_t shots, uint8_t *decodings, double *llr,
                            int32_t *iterations, uint8_t *converge) {
    const int m = r->bp->dec->check_count, n = r->bp->dec->bit_count;
    for (int64_t b = 0; b < shots; b++) {
        std::vector<uint8_t> s(syndromes + b * m, syndromes + (b + 1) * m);
        r->bp->dec->decode(s);
        iterations[b] = r->bp->dec->iterations;
        converge[b] = r->bp->dec->converge ? 1 : 0;
        if (llr) std::memcpy(llr + b * n, r->bp->dec->log_prob_ratios.data(), sizeof(double) * (size_t)n);
        if (r->bp->dec->converge) {
            std::memcpy(decodings + b * n, r->bp->dec->decoding.data(), (size_t)n);
        } else {
            auto &x = r->osd->decode(s, r->bp->dec->log_prob_ratios);
            std::memcpy(decodings + b * n, x.data(), (size_t)n);
        }
    }
}

/* osd_method / osd_order as BpOsdDecoder's setters write them, followed by osd_setup() (_bposd_decoder.pyx:64-70):
 * 1 = OSD_0, 2 = EXHAUSTIVE (OSD_E), 3 = COMBINATION_SWEEP (OSD_CS); osd.hpp:18-23, 59-101 */
void ref_bposd_set_osd(ref_bposd *r, int osd_method, int osd_order) {
    delete r->osd->LuDecomposition;
    r->osd->LuDecomposition = nullptr;
    r->osd->osd_method = static_cast<ldpc::osd::OsdMethod>(osd_method);
    r->osd->osd_order = osd_order;
    r->osd->osd_setup();
}

int ref_bposd_k(ref_bposd *r) { return r->osd->k; }

/* OsdDecoder::decode alone (osd.hpp:110-191) on caller-supplied log-ratios */
void ref_osd0(ref_bposd *r, const uint8_t *syndrome, const double *llr, uint8_t *decoding) {
    const int m = r->bp->dec->check_count, n = r->bp->dec->bit_count;
    std::vector<uint8_t> s(syndrome, syndrome + m);
    std::vector<double> l(llr, llr + n);
    auto &x = r->osd->decode(s, l);
    std::memcpy(decoding, x.data(), (size_t)n);
}

} /* extern "C" */
