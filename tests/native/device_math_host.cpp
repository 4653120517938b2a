// Host build of the DEVICE math helpers (ldpc_amd/csrc/bp_math.h) for CPU-side accuracy tests.
// Test infrastructure: lets tests/test_device_math.py measure tanh_half / log_pos against the host
// libm and run the oracle with the device routines plugged in (oracle/bp_oracle.c: bp_oracle_set_math).
#include "../../ldpc_amd/csrc/bp_math.h"

extern "C" {
double dm_tanh_half(double b) { return ldpc_math::tanh_half(b); }
double dm_log_ratio(double x) { return ldpc_math::ps_log_ratio(x); }
double dm_log_pos(double q) { return ldpc_math::log_pos(q); }
void dm_tanh_half_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = ldpc_math::tanh_half(in[i]); }
void dm_log_pos_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = ldpc_math::log_pos(in[i]); }
void dm_log_ratio_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = ldpc_math::ps_log_ratio(in[i]); }
// long double references (x87 80-bit: 64-bit significand) for ulp measurements
void ref_tanh_half_v(long n, const double *in, long double *out) { for (long i = 0; i < n; i++) out[i] = tanhl((long double)in[i] / 2); }
void ref_log_v(long n, const double *in, long double *out) { for (long i = 0; i < n; i++) out[i] = logl((long double)in[i]); }
double dx_tanh_half(double b) { return ldpc_math::tanh_half_libm(b); }
double dx_log_ratio(double x) { return ldpc_math::ps_log_ratio_libm(x, ldpc_math::k_log_tab); }
void dx_tanh_half_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = ldpc_math::tanh_half_libm(in[i]); }
void dx_log_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = ldpc_math::log_libm(in[i], ldpc_math::k_log_tab); }
void dx_log_ratio_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = ldpc_math::ps_log_ratio_libm(in[i], ldpc_math::k_log_tab); }
// the split routines of the check pass's fast path, recombined: must equal log_libm wherever q is a normal number in [2^-54, 2^54]
void dx_log_split_v(long n, const double *in, double *out) {
    for (long i = 0; i < n; i++)
        out[i] = ldpc_math::log_near_one(in[i]) ? ldpc_math::log_libm_near_one(in[i]) : ldpc_math::log_libm_general(in[i], ldpc_math::k_log_tab);
}
void libm_log_ratio_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = std::log((1 + in[i]) / (1 - in[i])); }
void *dx_tanh_half_ptr() { return (void *)&dx_tanh_half; }
void *dx_log_ratio_ptr() { return (void *)&dx_log_ratio; }
void *dm_tanh_half_ptr() { return (void *)&dm_tanh_half; }
void *dm_log_ratio_ptr() { return (void *)&dm_log_ratio; }
}
// the host libm itself (what the reference calls), vectorised -- NumPy's own SIMD tanh/log are NOT glibc's
extern "C" {
void libm_tanh_half_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = std::tanh(in[i] / 2); }
void libm_log_v(long n, const double *in, double *out) { for (long i = 0; i < n; i++) out[i] = std::log(in[i]); }
}
