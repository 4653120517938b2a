// cpp_host_demo.cpp -- the C++ host side over the C ABI, no Python: decodes the reference's own
// known-answer cases (cpp_test/TestBPDecoder.cpp:166-231: repetition code 5, five syndromes, product-sum
// and min-sum) through ldpc_hip::BpDecoder and checks the expected hard decisions; then the serial schedule, a received
// vector and analog syndromes against outputs of the reference on the same inputs.
//   g++ -std=c++17 -Iinclude examples/cpp_host_demo.cpp -Lldpc_amd/lib -lldpc_hip -Wl,-rpath,'$ORIGIN/../ldpc_amd/lib' -o examples/cpp_host_demo
#include <cstdio>
#include "ldpc_hip.hpp"

int main() {
    const int n = 5, m = 4;
    std::vector<int32_t> rp{0, 2, 4, 6, 8}, ci{0, 1, 1, 2, 2, 3, 3, 4};  // rows i: bits i, i+1
    const std::vector<std::vector<uint8_t>> syndromes{{0, 0, 0, 0}, {0, 0, 0, 1}, {0, 1, 0, 1}, {1, 0, 1, 0}, {1, 1, 1, 1}};
    const std::vector<std::vector<uint8_t>> expected{{0, 0, 0, 0, 0}, {0, 0, 0, 0, 1}, {0, 0, 1, 1, 0}, {0, 1, 1, 0, 0}, {0, 1, 0, 1, 0}};
    int bad = 0;
    for (auto method : {ldpc_hip::PRODUCT_SUM, ldpc_hip::MINIMUM_SUM}) {
        ldpc_hip::BpDecoder dec(m, n, rp, ci, std::vector<double>(n, 0.1), n, method, 1.0);
        for (size_t k = 0; k < syndromes.size(); ++k) {
            auto s = syndromes[k];
            auto &d = dec.decode(s);
            if (dec.last_status != 0) { std::printf("decode failed: %s\n", dec.last_error.c_str()); return 2; }
            if (d != expected[k]) { ++bad; std::printf("method %d syndrome %zu: wrong decoding\n", (int)method, k); }
        }
        // batch form: all five at once
        std::vector<uint8_t> flat;
        for (auto &s : syndromes) flat.insert(flat.end(), s.begin(), s.end());
        if (!dec.decode_batch(flat.data(), (int64_t)syndromes.size())) { std::printf("batch failed: %s\n", dec.last_error.c_str()); return 2; }
        for (size_t k = 0; k < syndromes.size(); ++k)
            for (int j = 0; j < n; ++j)
                if (dec.decoding_batch[k * n + j] != expected[k][j]) ++bad;
        dec.maximum_iterations = 1;  // public members are live, as in the reference
        auto s = syndromes[4];
        dec.decode(s);
        if (dec.iterations != 1) { ++bad; std::printf("maximum_iterations member not honoured\n"); }
    }
    {   // serial schedule, received-vector input and analog syndromes: expected values from the reference itself
        // (bp.hpp:451-545, 162-180, 547-660 run through oracle/ref_harness.cpp on these inputs)
        ldpc_hip::BpDecoder dec(m, n, rp, ci, std::vector<double>(n, 0.1), n, ldpc_hip::PRODUCT_SUM, 1.0);
        dec.schedule = ldpc_hip::SERIAL;
        for (size_t k = 0; k < syndromes.size(); ++k) {
            auto s = syndromes[k];
            if (dec.decode(s) != expected[k] || dec.last_status != 0) { ++bad; std::printf("serial schedule, syndrome %zu: wrong decoding\n", k); }
        }
        dec.serial_schedule_order = {4, 3, 2, 1, 0};
        auto s1 = syndromes[1];
        if (dec.decode(s1) != expected[1]) { ++bad; std::printf("serial schedule with an order: wrong decoding\n"); }
        dec.schedule = ldpc_hip::PARALLEL;
        dec.bp_input_type = ldpc_hip::AUTO;
        std::vector<uint8_t> r{0, 0, 0, 0, 1};  // n entries: taken as a received vector; the single flip is removed
        if (dec.decode(r) != std::vector<uint8_t>{0, 0, 0, 0, 0}) { ++bad; std::printf("received-vector input: wrong decoding\n"); }
        dec.bp_input_type = ldpc_hip::SYNDROME;
        dec.bp_method = ldpc_hip::MINIMUM_SUM;
        dec.maximum_iterations = 5;
        const std::vector<std::vector<double>> soft{{2.0, 2.0, 2.0, -2.0}, {2.0, -0.3, 2.0, -2.0}, {0.2, -0.1, 0.3, -0.2}, {-2.0, 2.0, -2.0, 2.0}};
        const std::vector<std::vector<uint8_t>> want_soft{{0, 0, 0, 0, 1}, {0, 0, 0, 0, 1}, {0, 0, 0, 0, 0}, {0, 1, 1, 0, 0}};  // cutoff 2, sigma 0.7
        const int want_iters[4] = {1, 1, 1, 4};
        for (size_t k = 0; k < soft.size(); ++k) {
            auto a = soft[k];
            if (dec.soft_info_decode_serial(a, 2.0, 0.7) != want_soft[k] || dec.iterations != want_iters[k] || !dec.converge) {
                ++bad;
                std::printf("soft_info_decode_serial %zu: wrong result\n", k);
            }
        }
        if (dec.soft_syndrome.size() != 4 || dec.soft_syndrome[1] < 8.16 || dec.soft_syndrome[1] > 8.17) { ++bad; std::printf("soft_syndrome member not filled\n"); }
    }
    {   // one object over several GPUs (device_ids; here GPU 0 twice, so that a one-GPU machine runs the sharded path too,
        // and every visible GPU when there are more): a batch of 300 rows gives what the single-GPU object gives
        int ngpu = 0;
        {
            ldpc_hip_bp_multi *probe = nullptr;
            ldpc_hip_bp_desc d{m, n, (int32_t)ci.size(), rp.data(), ci.data(), nullptr, n, 0, 1.0, -1};
            std::vector<double> p(n, 0.1);
            d.channel_probs = p.data();
            int32_t ids[64];
            for (ngpu = 0; ngpu < 64; ++ngpu) {  // how many devices does the library see?
                ids[0] = ngpu;
                if (ldpc_hip_bp_multi_create(&d, ids, 1, &probe) != LDPC_HIP_OK) break;
                ldpc_hip_bp_multi_destroy(probe);
            }
        }
        std::vector<uint8_t> flat;
        for (int k = 0; k < 300; ++k) flat.insert(flat.end(), syndromes[(size_t)(k * 7 % 5)].begin(), syndromes[(size_t)(k * 7 % 5)].end());
        ldpc_hip::BpDecoder one(m, n, rp, ci, std::vector<double>(n, 0.1), n, ldpc_hip::PRODUCT_SUM, 1.0);
        if (!one.decode_batch(flat.data(), 300)) { std::printf("batch failed: %s\n", one.last_error.c_str()); return 2; }
        std::vector<std::vector<int>> sets{{0, 0}, {0, 0, 0}};
        if (ngpu > 1) { std::vector<int> all; for (int g = 0; g < ngpu; ++g) all.push_back(g); sets.push_back(all); }
        for (auto &ids : sets) {
            ldpc_hip::BpDecoder many(m, n, rp, ci, std::vector<double>(n, 0.1), n, ldpc_hip::PRODUCT_SUM, 1.0, -1, ids);
            if (many.device_count() != (int)ids.size()) { ++bad; std::printf("device_count\n"); }
            if (!many.decode_batch(flat.data(), 300)) { std::printf("multi-GPU batch failed: %s\n", many.last_error.c_str()); return 2; }
            if (many.decoding_batch != one.decoding_batch || many.log_prob_ratios_batch != one.log_prob_ratios_batch ||
                many.iterations_batch != one.iterations_batch || many.converge_batch != one.converge_batch) {
                ++bad;
                std::printf("device_ids of %zu entries: results differ from the single-GPU object\n", ids.size());
            }
        }
        std::printf("multi-GPU object: %d GPU(s) visible, %zu device lists checked\n", ngpu, sets.size());
    }
    std::printf(bad ? "FAIL (%d)\n" : "cpp_host_demo: all reference known answers reproduced (PASS)\n", bad);
    return bad ? 1 : 0;
}
