// ldpc_hip.hpp -- header-only C++ host class over the C ABI (include/ldpc_hip.h).
//
// `ldpc_hip::BpDecoder` carries the member surface the reference's Cython layer touches on
// `ldpc::bp::BpDecoder` (src_cpp/bp.hpp:51-76, declared to Cython in _bp_decoder.pxd:47-83): public,
// mutable `channel_probabilities`, `maximum_iterations`, `bp_method`, `ms_scaling_factor`, and the results
// `decoding`, `log_prob_ratios`, `iterations`, `converge`; `decode(std::vector<uint8_t>&)` for one syndrome and
// the additive `decode_batch`; also `schedule` / `serial_schedule_order` (fixed-order serial sweep, bp.hpp:451-545),
// `bp_input_type` (syndrome or received vector, bp.hpp:162-180) and `soft_info_decode_serial` with its `soft_syndrome`
// result (bp.hpp:547-660).  A binding written against the reference class therefore transliterates
// (INTEGRATION.md §2).  As in the reference, construction throws on bad input (`except +` in the pxd) and
// decode does not throw across the binding: it records `last_status` / `last_error` instead.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "ldpc_hip.h"

namespace ldpc_hip {

enum BpMethod { PRODUCT_SUM = LDPC_HIP_PRODUCT_SUM, MINIMUM_SUM = LDPC_HIP_MINIMUM_SUM };  // bp.hpp:23-26
enum BpSchedule { SERIAL = 0, PARALLEL = 1, SERIAL_RELATIVE = 2 };                        // bp.hpp:28-32
enum BpInputType { SYNDROME = 0, RECEIVED_VECTOR = 1, AUTO = 2 };                          // bp.hpp:34-38

class BpDecoder {
public:
    // ---- members named as in ldpc::bp::BpDecoder (bp.hpp:54-75) ----
    std::vector<double> channel_probabilities;
    int check_count = 0;
    int bit_count = 0;
    int maximum_iterations = 0;
    BpMethod bp_method = PRODUCT_SUM;
    BpSchedule schedule = PARALLEL;            // SERIAL_RELATIVE is refused by the device (LDPC_HIP_ERR_UNSUPPORTED)
    BpInputType bp_input_type = SYNDROME;
    double ms_scaling_factor = 1.0;
    std::vector<int> serial_schedule_order;    // empty: 0 .. n-1 (bp.hpp:120-124)
    std::vector<double> soft_syndrome;         // after soft_info_decode_serial (bp.hpp:65, 547-660)
    std::vector<uint8_t> decoding;
    std::vector<double> log_prob_ratios;
    int iterations = 0;
    bool converge = false;
    // ---- batch results (additive) ----
    std::vector<uint8_t> decoding_batch;       // [batch][n]
    std::vector<double> log_prob_ratios_batch; // [batch][n]
    std::vector<int32_t> iterations_batch;     // [batch]
    std::vector<uint8_t> converge_batch;       // [batch]
    std::vector<uint8_t> osd_status_batch;     // [batch] after decode_batch(..., osd = true): 0 BP converged, 1 OSD solved H x = s,
                                               // 2 syndrome outside the image of H (ldpc_hip_bposd_get_status, ldpc_hip.h)
    int last_status = LDPC_HIP_OK;
    std::string last_error;

    // H as CSR with strictly ascending column indices per row (what insert_entry maintains, sparse_matrix_base.hpp:423-482)
    BpDecoder(int m, int n, const std::vector<int32_t> &csr_row_ptr, const std::vector<int32_t> &csr_col_idx,
              std::vector<double> channel_probs, int max_iter = 0, BpMethod method = PRODUCT_SUM,
              double min_sum_scaling_factor = 0.625, int device = -1, const std::vector<int> &device_ids = {})
        : channel_probabilities(std::move(channel_probs)), check_count(m), bit_count(n),
          maximum_iterations(max_iter > 0 ? max_iter : n), bp_method(method), ms_scaling_factor(min_sum_scaling_factor) {
        if ((int)channel_probabilities.size() != n)  // bp.hpp:103-106
            throw std::runtime_error("Channel probabilities vector must have length equal to the number of bits");
        if ((int)csr_row_ptr.size() != m + 1) throw std::runtime_error("csr_row_ptr must have m + 1 entries");
        ldpc_hip_bp_desc d;
        d.m = m; d.n = n; d.nnz = (int32_t)csr_col_idx.size();
        d.csr_row_ptr = csr_row_ptr.data(); d.csr_col_idx = csr_col_idx.data();
        d.channel_probs = channel_probabilities.data();
        d.max_iter = maximum_iterations; d.bp_method = (int32_t)bp_method; d.ms_scaling_factor = ms_scaling_factor;
        d.device = device;
        if (!device_ids.empty()) {
            // several GPUs behind one object: decode_batch shards the rows over them (ldpc_hip_bp_multi, ldpc_hip.h);
            // single-vector calls and H r run on the first one
            std::vector<int32_t> ids(device_ids.begin(), device_ids.end());
            if (ldpc_hip_bp_multi_create(&d, ids.data(), (int32_t)ids.size(), &mh_) != LDPC_HIP_OK) throw std::runtime_error(ldpc_hip_last_error());
            h_ = ldpc_hip_bp_multi_handle(mh_, 0);
        } else if (ldpc_hip_bp_create(&d, &h_) != LDPC_HIP_OK) throw std::runtime_error(ldpc_hip_last_error());
        synced_probs_ = channel_probabilities;
        decoding.assign((size_t)n, 0);
        log_prob_ratios.assign((size_t)n, 0.0);
    }
    BpDecoder(const BpDecoder &) = delete;
    BpDecoder &operator=(const BpDecoder &) = delete;
    ~BpDecoder() { if (mh_) ldpc_hip_bp_multi_destroy(mh_); else ldpc_hip_bp_destroy(h_); }

    // one input vector: ldpc::bp::BpDecoder::decode (bp.hpp:159-190).  A received vector r (bp_input_type
    // RECEIVED_VECTOR, or AUTO and n entries) is turned into the syndrome H r on the device, decoded, and XORed back in.
    std::vector<uint8_t> &decode(std::vector<uint8_t> &input_vector) {
        const bool received = bp_input_type == RECEIVED_VECTOR || (bp_input_type == AUTO && (int)input_vector.size() == bit_count);
        if ((int)input_vector.size() != (received ? bit_count : check_count)) { fail_(LDPC_HIP_ERR_INVALID, "input vector has the wrong length"); return decoding; }
        if (!sync_()) return decoding;
        std::vector<uint8_t> syndrome_of_r;
        const uint8_t *synd = input_vector.data();
        if (received) {
            syndrome_of_r.assign((size_t)check_count, 0);
            last_status = ldpc_hip_gf2_mulvec_batch(h_, input_vector.data(), 1, syndrome_of_r.data());
            if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return decoding; }
            synd = syndrome_of_r.data();
        }
        int32_t it = 0;
        uint8_t cv = 0;
        last_status = ldpc_hip_bp_decode_batch(h_, synd, 1, decoding.data(), log_prob_ratios.data(), &it, &cv);
        if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return decoding; }
        if (received)
            for (int j = 0; j < bit_count; ++j) decoding[(size_t)j] ^= input_vector[(size_t)j];
        iterations = it;
        converge = cv != 0;
        return decoding;
    }

    // ldpc::bp::BpDecoder::soft_info_decode_serial (bp.hpp:547-660): analog syndrome readouts, serial minimum-sum with
    // `ms_scaling_factor` taken literally, virtual check nodes below `cutoff`.  Fills `soft_syndrome` as the reference does.
    std::vector<uint8_t> &soft_info_decode_serial(std::vector<double> &soft_info_syndrome, double cutoff, double sigma) {
        if ((int)soft_info_syndrome.size() != check_count) { fail_(LDPC_HIP_ERR_INVALID, "soft syndrome has the wrong length"); return decoding; }
        if (!sync_()) return decoding;
        soft_syndrome.assign((size_t)check_count, 0.0);
        int32_t it = 0;
        uint8_t cv = 0;
        last_status = ldpc_hip_bp_soft_info_decode_batch(h_, soft_info_syndrome.data(), 1, cutoff, sigma, decoding.data(),
                                                         log_prob_ratios.data(), &it, &cv, soft_syndrome.data());
        if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return decoding; }
        iterations = it;
        converge = cv != 0;
        return decoding;
    }

    // `batch` syndromes, row-major [batch][m]; osd = true post-processes unconverged rows with ordered-statistics
    // decoding (osd.hpp:103-187) of method `osd_method` (1 OSD_0, 2 OSD_E, 3 OSD_CS; osd.hpp:18-23) and `osd_order`
    int osd_method = 1, osd_order = 0;
    bool decode_batch(const uint8_t *syndromes, int64_t batch, bool want_llr = true, bool osd = false) {
        decoding_batch.resize((size_t)batch * bit_count);
        if (want_llr) log_prob_ratios_batch.resize((size_t)batch * bit_count);
        iterations_batch.resize((size_t)batch);
        converge_batch.resize((size_t)batch);
        return decode_batch_into(syndromes, batch, decoding_batch.data(), want_llr ? log_prob_ratios_batch.data() : nullptr,
                                 iterations_batch.data(), converge_batch.data(), osd);
    }
    // the same into the caller's arrays ([batch][n] decoding, [batch][n] log-ratios or null, [batch] iterations, [batch] converge):
    // what a binding that hands out NumPy arrays wants -- the results cross the PCIe link straight into them (large batches: in
    // pinned, double-buffered chunks that overlap the kernels, ldpc_hip.h)
    bool decode_batch_into(const uint8_t *syndromes, int64_t batch, uint8_t *dec_out, double *llr_out, int32_t *iters_out,
                           uint8_t *conv_out, bool osd = false) {
        if (!sync_()) return false;
        if (osd) {
            last_status = each_([&](ldpc_hip_bp *h) { return ldpc_hip_bp_set_osd(h, osd_method, osd_order); });
            if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return false; }
        }
        auto fn = osd ? ldpc_hip_bposd_decode_batch : ldpc_hip_bp_decode_batch;
        last_status = mh_ ? ldpc_hip_bp_multi_decode_batch(mh_, osd ? 1 : -1, syndromes, batch, dec_out, llr_out, iters_out, conv_out)
                          : fn(h_, syndromes, batch, dec_out, llr_out, iters_out, conv_out);
        if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return false; }
        osd_status_batch.clear();
        if (osd && !mh_) {  // (the sharded object keeps one status array per GPU; ask the handles for those)
            osd_status_batch.assign((size_t)batch, 0);
            last_status = ldpc_hip_bposd_get_status(h_, osd_status_batch.data(), batch);
            if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return false; }
        }
        return true;
    }

    ldpc_hip_bp *handle() { return h_; }
    ldpc_hip_bp_multi *multi_handle() { return mh_; }  // null unless constructed with device_ids
    int device_count() const { return mh_ ? (int)ldpc_hip_bp_multi_devices(mh_) : 1; }

private:
    // the reference lets callers write the public members between decodes; push them to the device handle
    bool sync_() {
        if (channel_probabilities != synced_probs_) {
            last_status = each_([&](ldpc_hip_bp *h) { return ldpc_hip_bp_set_channel(h, channel_probabilities.data(), (int32_t)channel_probabilities.size()); });
            if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return false; }
            synced_probs_ = channel_probabilities;
        }
        last_status = each_([&](ldpc_hip_bp *h) { return ldpc_hip_bp_set_params(h, maximum_iterations, (int32_t)bp_method, ms_scaling_factor); });
        if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return false; }
        if ((int)schedule != synced_schedule_ || serial_schedule_order != synced_order_) {
            if (!serial_schedule_order.empty() && (int)serial_schedule_order.size() != bit_count) {
                fail_(LDPC_HIP_ERR_INVALID, "serial_schedule_order must have n entries");
                return false;
            }
            std::vector<int32_t> order(serial_schedule_order.begin(), serial_schedule_order.end());
            last_status = each_([&](ldpc_hip_bp *h) { return ldpc_hip_bp_set_schedule(h, (int32_t)schedule, order.empty() ? nullptr : order.data()); });
            if (last_status != LDPC_HIP_OK) { last_error = ldpc_hip_last_error(); return false; }
            synced_schedule_ = (int)schedule;
            synced_order_ = serial_schedule_order;
        }
        return true;
    }
    void fail_(int code, const char *msg) { last_status = code; last_error = msg; }
    template <class F>
    int each_(F f) {  // a setter on every GPU's handle
        const int nd = device_count();
        for (int i = 0; i < nd; ++i) {
            const int rc = f(mh_ ? ldpc_hip_bp_multi_handle(mh_, i) : h_);
            if (rc != LDPC_HIP_OK) return rc;
        }
        return LDPC_HIP_OK;
    }
    ldpc_hip_bp *h_ = nullptr;
    ldpc_hip_bp_multi *mh_ = nullptr;
    std::vector<double> synced_probs_;
    int synced_schedule_ = (int)PARALLEL;
    std::vector<int> synced_order_;
};

}  // namespace ldpc_hip
