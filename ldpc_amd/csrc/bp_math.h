// bp_math.h -- FP64 elementary functions for the product-sum check update on gfx950.
//
// The reference evaluates, per edge and per iteration (src_cpp/bp.hpp:205-218),
//     t = std::tanh(b / 2)          and          c = std::log((1 + x) / (1 - x))
// with the host libm.  ocml's double-double tanh/log are several hundred instructions each; the two
// routines below are ~40 FP64 instructions each and are written so that they ROUND like a good libm
// does, which matters more here than raw ulp counts:
//
//   * tanh_half(b) follows the classical expm1 formulation libms use (|x| >= 1: 1 - 2/(expm1(2|x|)+2),
//     |x| < 1: -e/(e+2) with e = expm1(-2|x|)), because the reference's results depend on how
//     tanh rounds just below 1.0: x = prod tanh(...) feeds log((1+x)/(1-x)), so one ulp of t near 1
//     moves a message by ~1e-4.  "1 - q" with an accurately computed small q rounds the way the
//     exact tanh does; "(1-E)/(1+E)" does not.
//   * log_pos(q) is the classical s = f/(2+f), log(1+f) = 2 atanh(s) evaluation with the degree-7
//     minimax polynomial in s^2 (coefficients Lg1..Lg7 from Sun's freely redistributable fdlibm
//     e_log.c, the published algorithm most libms derive from), error < 1 ulp.
//
// Everything is (nearly) straight-line code (no per-lane divergence) and is host-compilable so
// tests/test_device_math.py can measure it against the host libm on the CPU.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#include "bp_libm_tables.h"

#if defined(__HIPCC__) || defined(__HIP_DEVICE_COMPILE__)
#define LDPC_HD __host__ __device__ __forceinline__
#else
#define LDPC_HD static inline
#endif

namespace ldpc_math {

LDPC_HD double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

// a * B + C for two CONSTANTS B and C.  A VOP3 instruction of gfx9 reads at most one scalar operand, so one constant sits in a
// VGPR pair; left to itself the compiler then copies that pair into the destination and issues a two-address v_fmac (one extra
// VALU instruction per Horner step whose addend is a constant).  Spelt out, the three-address form needs no copy.
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double fma_kk(double a, double B, double C) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(B), "v"(C));
    return d;
}
#else
LDPC_HD double fma_kk(double a, double B, double C) { return __builtin_fma(a, B, C); }
#endif

// 2^k as a double for k in [-1022, 1023] by exponent construction (no libm call)
LDPC_HD double pow2i(int k) {
    const uint64_t bits = (uint64_t)(int64_t)(k + 1023) << 52;
    double r;
    memcpy(&r, &bits, sizeof r);
    return r;
}

// z = k*ln2 + r, |r| <= ln2/2 (Cody-Waite, ln2 split so that k*ln2_hi is exact); returns expm1(r)
// (Taylor through r^13: truncation < 2^-56 relative on the reduced interval) and k.
LDPC_HD double expm1_reduced(double z, int &k) {
    const double inv_ln2 = 1.44269504088896338700e+00;
    const double ln2_hi = 6.93147180369123816490e-01;  // 0x3fe62e42fee00000
    const double ln2_lo = 1.90821492927058770002e-10;  // 0x3dea39ef35793c76
    const double kf = __builtin_rint(z * inv_ln2);
    k = (int)kf;
    double r = fma_(-kf, ln2_hi, z);
    r = fma_(-kf, ln2_lo, r);
    double p = 1.0 / 6227020800.0;            // 1/13!
    p = fma_(p, r, 1.0 / 479001600.0);        // 1/12!
    p = fma_(p, r, 1.0 / 39916800.0);
    p = fma_(p, r, 1.0 / 3628800.0);
    p = fma_(p, r, 1.0 / 362880.0);
    p = fma_(p, r, 1.0 / 40320.0);
    p = fma_(p, r, 1.0 / 5040.0);
    p = fma_(p, r, 1.0 / 720.0);
    p = fma_(p, r, 1.0 / 120.0);
    p = fma_(p, r, 1.0 / 24.0);
    p = fma_(p, r, 1.0 / 6.0);
    p = fma_(p, r, 0.5);
    return fma_(p * r, r, r);  // r + r^2 * P(r)
}

// tanh(b / 2), sign-symmetric, NaN -> NaN, +-inf -> +-1, |b| > ~38.2 -> +-1 exactly.
LDPC_HD double tanh_half(double b) {
    double a = __builtin_fabs(b);
    a = a > 40.0 ? 40.0 : a;       // tanh(20) == 1.0 already; keeps exp in range; NaN stays NaN
    const bool big = a >= 2.0;     // |b/2| >= 1
    int k;
    const double p = expm1_reduced(big ? a : -a, k);
    const double twok = pow2i(k);  // k in [-3, 58]
    // big:   e^a + 1        = 2^k p + (2^k + 1)         (one rounding)
    // small: expm1(-a)      = 2^k p + (2^k - 1)         (one rounding)
    const double em = fma_(twok, p, twok + (big ? 1.0 : -1.0));
    const double den = big ? em : em + 2.0;
    const double num = big ? 2.0 : -em;
    const double quo = num / den;
    const double t = big ? 1.0 - quo : quo;
    return __builtin_copysign(t, b);
}

// log(q) for q in [0, +inf] (the ratio (1+x)/(1-x) is never negative): 0 -> -inf, +inf -> +inf,
// NaN -> NaN.  q is never subnormal on this path (q >= 2^-54 when non-zero).
LDPC_HD double log_pos(double q) {
    const double ln2_hi = 6.93147180369123816490e-01;
    const double ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t bits;
    memcpy(&bits, &q, sizeof bits);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    uint64_t mb = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;  // mantissa in [1, 2)
    double mant;
    memcpy(&mant, &mb, sizeof mant);
    const bool up = mant > 1.41421356237309504880;
    mant = up ? mant * 0.5 : mant;  // [sqrt(1/2), sqrt(2)]
    e += up ? 1 : 0;
    const double f = mant - 1.0;    // exact
    const double s = f / (2.0 + f);
    const double z = s * s;
    const double w = z * z;
    const double t1 = w * fma_(w, fma_(w, Lg6, Lg4), Lg2);
    const double t2 = z * fma_(w, fma_(w, fma_(w, Lg7, Lg5), Lg3), Lg1);
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double dk = (double)e;
    double r = dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    r = q == 0.0 ? -INFINITY : r;
    r = q == INFINITY ? INFINITY : r;
    r = q != q ? q : r;
    return r;
}

// check -> bit product-sum message body (bp.hpp:211-216): log((1 + x) / (1 - x))
LDPC_HD double ps_log_ratio(double x) { return log_pos((1.0 + x) / (1.0 - x)); }

// =================================================================================================
// Bit-identical twins of the host libm the reference runs on (x86-64 glibc >= 2.28, FMA-capable CPU).
//
// std::tanh there is fdlibm's algorithm on top of fdlibm's expm1 (glibc sysdeps/ieee754/dbl-64/
// s_tanh.c, s_expm1.c: plain IEEE double operations, no FMA -- there is no multiarch variant), and
// std::log is the table-driven routine of ARM's optimized-routines (e_log.c, the FMA multiarch build
// contracts every mul+add).  Both are deterministic sequences of IEEE operations, so repeating the
// same operations in the same order -- with fma() exactly where the host build fuses -- yields the
// same bits on a GPU.  tests/test_device_math.py verifies bit-identity against the host libm on
// tens of millions of arguments; with these two routines the product-sum kernel reproduces the
// reference's log-probability ratios BIT FOR BIT, which also settles every knife-edge case (a
// posterior that is exactly 0.0 in the reference stays exactly 0.0 here).
// Everything is restricted to the arguments the BP update can produce (see each routine).
// =================================================================================================

// IEEE-754 correctly rounded n / d for finite, non-zero, normal d and results far from the overflow /
// underflow thresholds -- every division of the product-sum update qualifies once d == 0 is handled by
// the caller (|operands| in [2^-54, 2^65]).  On the device this is the Newton-Raphson + final-residual
// sequence the compiler itself emits for `/` (v_rcp_f64, four FMAs, q = n*r, rem = n - d*q,
// q + rem*r), minus the v_div_scale / v_div_fmas / v_div_fixup range handling that is a no-op for such
// operands: same bits, three instructions fewer per division.  On the host it is just `/`.
LDPC_HD double div_cr(double n, double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(d);
    r = fma_(fma_(-d, r, 1.0), r, r);
    r = fma_(fma_(-d, r, 1.0), r, r);
    const double q = n * r;
    return fma_(fma_(-d, q, n), r, q);
#else
    return n / d;
#endif
}

LDPC_HD uint64_t as_u64(double x) { uint64_t u; memcpy(&u, &x, sizeof u); return u; }
LDPC_HD double as_f64(uint64_t u) { double x; memcpy(&x, &u, sizeof x); return x; }
LDPC_HD double add_exponent(double y, int k) { return as_f64(as_u64(y) + ((uint64_t)(int64_t)k << 52)); }

// "does any lane of the wavefront need this rare fix-up?"  On the device the answer is wave-uniform, so a
// rarely-true condition costs one compare + one scalar branch instead of selects on every call; on the
// host it is just the condition.
#if defined(__HIP_DEVICE_COMPILE__)
#define LDPC_ANY(cond) (__builtin_amdgcn_ballot_w64(cond) != 0)
#else
#define LDPC_ANY(cond) (cond)
#endif

// std::tanh(b / 2) exactly as glibc evaluates it (sysdeps/ieee754/dbl-64/s_tanh.c on top of fdlibm's
// __expm1, s_expm1.c), any double b.
//
//   |x| >= 1 :  t = expm1( 2|x|),  z = 1 - 2/(t + 2)          |x| < 1 :  t = expm1(-2|x|),  z = -t/(t + 2)
//
// expm1 is inlined and restricted to those two argument ranges (w in [2, 44) with k >= 3, or w in (-2, 0)
// with k in {0,-1,-2,-3}); every k-dependent tail of the original is evaluated by the original's operations
// and selected, so lanes in different ranges do not serialise:
//   k == 0      : x - e                          (with c = 0 the general e reduces to x*e0 - hxs)
//   k == -1     : 0.5*(x - e) - 0.5              == -(0.5*(e - x) + 0.5)   (negation is exact)
//   k <= -2     : scalb(1 - (e - x), k) - 1
//   2 <= k < 20 : scalb((1 - 2^-k) - (e - x), k)
//   k >= 20     : scalb((x - (e + 2^-k)) + 1, k)      (rare: |b| > 13.5; evaluated only if some lane needs it.
//                 The original's separate k > 56 form only matters for |x| < 22 when it yields t > 2^56,
//                 for which 1 - 2/(t+2) == 1.0 exactly, and this form gives t > 2^56 as well.)
// Tiny (|x| < 2^-55), huge (|x| >= 22), infinite and NaN arguments are patched afterwards (rare).
LDPC_HD double tanh_half_libm(double b) {
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                 invln2 = 1.44269504088896338700e+00;
    const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
                 Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
                 Q5 = -2.01099218183624371326e-07;
    // x = b / 2 and 2|x| = |b| exactly for every normal b (a subnormal b ends in the "tiny" patch below, which does
    // its own b * 0.5), so the argument reduction works on |b| and never forms x.
    const uint64_t bb = as_u64(b);
    const uint32_t hb = (uint32_t)(bb >> 32) & 0x7fffffffu;  // high word of |b| = high word of 2|x|
    const bool big = hb >= 0x40000000u;                      // |x| >= 1 (a NaN lands here too and is patched at the end)
    const uint32_t sign_w = big ? 0u : 0x80000000u;          // w = 2|x| if |x| >= 1, else -2|x|
    const double w = as_f64((bb & 0x7fffffffffffffffull) | ((uint64_t)sign_w << 32));
    // ---- __expm1(w): argument reduction (s_expm1.c) ----
    //   |w| <= 0.5 ln2 (hw <= 0x3fd62e42): k = 0;  0.5 ln2 < |w| < 1.5 ln2: k = +-1;  else k = (int)(invln2 w +- 0.5).
    //   The middle case needs no test of its own: there invln2 |w| + 0.5 lies in [1.0000003, 1.9999997] (the bounds are
    //   the doubles next to the two high-word thresholds, 1.5 ln2 itself sits INSIDE the high word 0x3ff0a2b2 and so takes
    //   the general formula in the original as well), which truncates to 1.
    int k = (int)(invln2 * w + as_f64((uint64_t)(0x3fe00000u | sign_w) << 32));  // + copysign(0.5, w)
    k = hb > 0x3fd62e42u ? k : 0;
    const double t_k = (double)k;
    const double hi = w - t_k * ln2_hi;  // k = 0: hi = w, lo = 0, c = 0 (identical to the unreduced path)
    const double lo = t_k * ln2_lo;
    const double x = hi - lo;
    const double c = (hi - x) - lo;
    const double hfx = 0.5 * x;
    const double hxs = x * hfx;
    const double R1 = 1.0 + hxs * Q1, h2 = hxs * hxs;
    const double R2 = Q2 + hxs * Q3, h4 = h2 * h2;
    const double R3 = Q4 + hxs * Q5;
    const double r1 = R1 + h2 * R2 + h4 * R3;
    const double tt = 3.0 - r1 * hfx;
    const double e0 = hxs * div_cr(r1 - tt, 6.0 - x * tt);  // denominator in [5.6, 6.4]
    double e = (x * (e0 - c) - c);
    e -= hxs;
    const double emx = e - x;
    // ---- tails ----
    // a1 = 1 - 2^-k for 2 <= k < 20, 1 for k <= -2; not used otherwise (k is never 1; k in {0, -1} and k >= 20 have forms of
    // their own).  A shift count is taken modulo 32, which turns k = -2, -3 into shifts by 30, 29: 0x200000 >> that = 0.
    const double a1 = as_f64((uint64_t)(0x3ff00000u - (0x200000u >> ((unsigned)k & 31u))) << 32);
    const double y1 = add_exponent(a1 - emx, k);      // 2 <= k < 20: the result; k <= -2: result + 1
    double t = k >= 2 ? y1 : y1 - 1.0;
    t = k == -1 ? -fma_(0.5, emx, 0.5) : t;           // -(0.5 * emx + 0.5): the product is exact, so the fused form rounds alike
    t = k == 0 ? -emx : t;
    // everything rare hangs off ONE test: |b| >= 13 (superset of k >= 20), |x| < 2^-55, |x| >= 22, inf, NaN
    const bool rare = hb - 0x3c900000u >= 0x402a0000u - 0x3c900000u;
    const bool any_rare = LDPC_ANY(rare);
    if (any_rare) {
        const double two_mk = as_f64((uint64_t)((uint32_t)(0x3ff - k) << 20) << 32);  // 2^-k
        const double y2 = add_exponent((x - (e + two_mk)) + 1.0, k);
        t = (big && k >= 20) ? y2 : t;
    }
    // ---- tanh from expm1 ----
    const double quo = div_cr(big ? 2.0 : -t, t + 2.0);  // denominator in [1.1, 2^64]
    double z = big ? 1.0 - quo : quo;
    z = __builtin_copysign(z, b);
    if (any_rare) {
        const double xh = b * 0.5;
        const double ax = __builtin_fabs(xh);
        z = ax < 0x1p-55 ? xh * (1.0 + xh) : z;                      // tiny and +-0 (carries its own sign)
        z = ax >= 22.0 ? __builtin_copysign(1.0, xh) : z;            // also +-inf
        z = xh != xh ? xh + xh : z;                                   // NaN
    }
    return z;
}

// {1/c, log(c)} pairs of glibc's log (bp_libm_tables.h).  Device code never indexes this array from
// the hot loop: a per-lane global load in the middle of the arithmetic would queue behind the streaming
// message traffic (gfx9 completes vector-memory operations in issue order), so the kernel stages the
// 2 KiB table into LDS once per workgroup and hands log_libm the LDS pointer.
#if defined(__HIPCC__)
__device__ const double k_log_tab[256] = LDPC_LOG_TAB;
#else
static const double k_log_tab[256] = LDPC_LOG_TAB;
#endif

// std::log(q) exactly as glibc's FMA build evaluates it, for q in [0, +inf] or NaN, q not subnormal.
LDPC_HD double log_libm(double q, const double *tab) {
#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__)
    return q;  // host pass of hipcc: never called
#else
    const double A[5] = LDPC_LOG_POLY;
    const double B[11] = LDPC_LOG_POLY1;
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const uint64_t ix = as_u64(q);
    const uint64_t LO = 0x3fee000000000000ull;  // asuint64(1.0 - 0x1p-4)
    const uint64_t HI = 0x3ff1090000000000ull;  // asuint64(1.0 + 0x1.09p-4)
    double y;
    if (ix - LO < HI - LO) {
        const double r = q - 1.0, r2 = r * r, r3 = r * r2;
        const double p3 = fma_(r3, B[10], fma_(r2, B[9], fma_kk(r, B[8], B[7])));
        const double p2 = fma_(r3, p3, fma_(r2, B[6], fma_kk(r, B[5], B[4])));
        const double p1 = fma_(r3, p2, fma_(r2, B[3], fma_kk(r, B[2], B[1])));
        double w = r * 0x1p27;
        const double rhi = r + w - w;
        const double rlo = r - rhi;
        w = rhi * rhi * B[0];  // B[0] == -0.5
        const double hi = r + w;
        double lo = r - hi + w;
        lo = fma_(B[0] * rlo, rhi + r, lo);
        y = fma_(r3, p1, lo) + hi;
        y = ix == 0x3ff0000000000000ull ? 0.0 : y;  // log(1) = +0 exactly
    } else {
        const uint64_t tmp = ix - 0x3fe6000000000000ull;
        const int i = (int)((tmp >> (52 - LDPC_LOG_TABLE_BITS)) & 127);
        const int64_t k = (int64_t)tmp >> 52;
        const double z = as_f64(ix - (tmp & (0xfffull << 52)));
        const double invc = tab[2 * i], logc = tab[2 * i + 1];
        const double r = fma_(z, invc, -1.0);
        const double kd = (double)k;
        const double w = fma_(kd, Ln2hi, logc);
        const double hi = w + r;
        const double lo = fma_(kd, Ln2lo, w - hi + r);
        const double r2 = r * r;
        const double poly = fma_(r2, fma_kk(r, A[4], A[3]), fma_kk(r, A[2], A[1]));
        y = fma_(r * r2, poly, fma_(r2, A[0], lo)) + hi;
    }
    if (LDPC_ANY(ix - 0x0010000000000000ull >= 0x7ff0000000000000ull - 0x0010000000000000ull)) {  // 0, inf, NaN (never subnormal here)
        y = q == 0.0 ? -INFINITY : y;
        y = q < INFINITY ? y : q;  // +inf -> +inf, NaN -> NaN
    }
    return y;
#endif
}

// ---- the same log, split for the check pass's fast path ----------------------------------------------------------
// There q = (1 + x) / (1 - x) with 0 <= |x| < 1 (the row's tanh values all have magnitude below 1 and none is NaN:
// tested once per row), so q is a normal number in [2^-54, 2^54]: no zero, infinity or NaN tail, and the two
// evaluation branches can be driven separately (the near-1 branch on compacted lanes, bp_device_common.h).
LDPC_HD bool log_near_one(double q) {  // glibc's `ix - LO < HI - LO` on the high word (LO and HI have zero low words)
    return (uint32_t)(as_u64(q) >> 32) - 0x3fee0000u < 0x3ff10900u - 0x3fee0000u;
}

// the table branch, for q outside [1 - 2^-4, 1 + 0x1.09p-4)
LDPC_HD double log_libm_general(double q, const double *tab) {
#if defined(__HIPCC__) && !defined(__HIP_DEVICE_COMPILE__)
    return q;  // host pass of hipcc: never called
#else
    const double A[5] = LDPC_LOG_POLY;
    const double Ln2hi = 0x1.62e42fefa3800p-1, Ln2lo = 0x1.ef35793c76730p-45;
    const uint64_t ix = as_u64(q);
    const uint32_t hq = (uint32_t)(ix >> 32);
    const uint32_t th = hq - 0x3fe60000u;                      // high word of tmp = ix - 0x3fe6000000000000
    const uint32_t off = (th >> 9) & 0x7f0u;                   // ((tmp >> 45) & 127) * 16: byte offset of {1/c, log c}
    const int k = (int32_t)th >> 20;                           // (int64_t)tmp >> 52
    const double z = as_f64(((uint64_t)(hq - (th & 0xfff00000u)) << 32) | (uint32_t)ix);
    const double *pair = reinterpret_cast<const double *>(reinterpret_cast<const char *>(tab) + off);
    const double invc = pair[0], logc = pair[1];
    const double r = fma_(z, invc, -1.0);
    const double kd = (double)k;
    const double w = fma_(kd, Ln2hi, logc);
    const double hi = w + r;
    const double lo = fma_(kd, Ln2lo, w - hi + r);
    const double r2 = r * r;
    const double poly = fma_(r2, fma_kk(r, A[4], A[3]), fma_kk(r, A[2], A[1]));
    return fma_(r * r2, poly, fma_(r2, A[0], lo)) + hi;
#endif
}

// the near-1 branch, for q in [1 - 2^-4, 1 + 0x1.09p-4).  q == 1 needs no patch: r = 0 makes every term +0.
LDPC_HD double log_libm_near_one(double q) {
    const double B[11] = LDPC_LOG_POLY1;
    const double r = q - 1.0, r2 = r * r, r3 = r * r2;
    const double p3 = fma_(r3, B[10], fma_(r2, B[9], fma_kk(r, B[8], B[7])));
    const double p2 = fma_(r3, p3, fma_(r2, B[6], fma_kk(r, B[5], B[4])));
    const double p1 = fma_(r3, p2, fma_(r2, B[3], fma_kk(r, B[2], B[1])));
    double w = r * 0x1p27;
    const double rhi = r + w - w;
    const double rlo = r - rhi;
    w = rhi * rhi * B[0];  // B[0] == -0.5
    const double hi = r + w;
    double lo = r - hi + w;
    lo = fma_(B[0] * rlo, rhi + r, lo);
    return fma_(r3, p1, lo) + hi;
}

// std::log((1 + x) / (1 - x)) with the host libm's bits; |x| <= 1 or NaN.  1 - x == 0 (x == 1) gives
// 2 / 0 = +inf exactly as IEEE division does.
LDPC_HD double ps_log_ratio_libm(double x, const double *tab) {
    const double den = 1.0 - x;
    double q = div_cr(1.0 + x, den);
    if (LDPC_ANY(den == 0.0)) q = den == 0.0 ? INFINITY : q;  // x == 1: 2 / 0 (div_cr alone would give NaN)
    return log_libm(q, tab);
}

}  // namespace ldpc_math
