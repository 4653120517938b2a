// bp_small_kernel.h -- bp_small_kernel: messages resident in LDS for small codes
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// ---- on-chip variant for small codes (BASELINE configs 3 and 5) ------------------------------------
// When both message arrays of a syndrome fit in a few KiB (rotated surface d=21: 13 KiB, BB [[144,12,12]]:
// 7 KiB) nothing but the syndrome and the results needs to touch HBM.  A workgroup keeps SLOTS syndromes
// resident in LDS and iterates them together; work items are (slot, node) pairs so lanes stay busy when
// m or n is not a multiple of 64.  A slot whose syndrome converged (or hit max_iter) writes its outputs and
// immediately pulls the next syndrome from a device-wide counter, so the work done is proportional to the
// iterations each syndrome really needs (the streaming kernel's 64-lane tiles run until their slowest
// lane finishes).  Per node the edges are walked sequentially in the reference's order with the
// reference's two sweeps (bp.hpp:205-218, 278-281 + 313-316), so results are bit-identical to the
// streaming kernel's and to the reference's.
struct SmallArgs {
    int32_t m, n, nnz, max_iter, slots;
    double ms_scaling_factor;
    int64_t batch;
    const int32_t *row_ptr, *col_idx, *col_ptr, *csc_edge;
    const double *llr0;
    const uint8_t *synd;        // [batch][m]
    uint8_t *decoding;          // [batch][n]
    double *llr;                // [batch][n] or nullptr
    int32_t *iters;             // [batch] or nullptr
    uint8_t *conv;              // [batch] or nullptr
    unsigned long long *next;   // device-wide work counter (zeroed before launch)
};

template <int METHOD, int MATH>
__global__ void __launch_bounds__(256) bp_small_kernel(const SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int m = a.m, n = a.n, nnz = a.nnz, S = a.slots;
    // LDS carve-up: [log table 2 KiB][llr0 n][row_ptr m+1][col_idx nnz][col_ptr n+1][csc_edge nnz] then per slot
    // [A nnz f64][C nnz f64][L n f64][hard n u8][sy m u8]
    double *log_tab = reinterpret_cast<double *>(sm_lds);
    double *prior = log_tab + 256;
    int32_t *rp = reinterpret_cast<int32_t *>(prior + n);
    int32_t *ci = rp + (m + 1);
    int32_t *cp = ci + nnz;
    int32_t *ce = cp + (n + 1);
    size_t off = (size_t)(reinterpret_cast<unsigned char *>(ce + nnz) - sm_lds);
    off = (off + 15) & ~(size_t)15;
    const size_t slot_bytes = ((size_t)nnz * 16 + (size_t)n * 8 + (size_t)n + (size_t)m + 15) & ~(size_t)15;
    __shared__ long long slot_synd[16];  // syndrome index held by the slot, -1 = idle
    __shared__ int slot_iter[16];
    __shared__ int slot_unsat[16];
    __shared__ int slot_state[16];       // 0 running, 1 finished this iteration (write out + refill), 2 fresh (needs init)
    __shared__ int n_active;

    for (int q = tid; q < 256; q += T) log_tab[q] = ldpc_math::k_log_tab[q];
    for (int q = tid; q < n; q += T) prior[q] = a.llr0[q];
    for (int q = tid; q <= m; q += T) rp[q] = a.row_ptr[q];
    for (int q = tid; q < nnz; q += T) { ci[q] = a.col_idx[q]; ce[q] = a.csc_edge[q]; }
    for (int q = tid; q <= n; q += T) cp[q] = a.col_ptr[q];
    if (tid < S) {
        const unsigned long long idx = atomicAdd(a.next, 1ull);
        slot_synd[tid] = idx < (unsigned long long)a.batch ? (long long)idx : -1;
        slot_state[tid] = 2;
        slot_iter[tid] = 0;
        slot_unsat[tid] = 0;
    }
    __syncthreads();

    const float inv_m = m > 0 ? 1.0f / (float)m : 0.f, inv_n = n > 0 ? 1.0f / (float)n : 0.f, inv_e = nnz > 0 ? 1.0f / (float)nnz : 0.f;
    auto slot_base = [&](int s) { return sm_lds + off + (size_t)s * slot_bytes; };
    auto split = [](int w, int len, float inv, int &s, int &r) {  // w = s * len + r, exact for w < 2^22
        s = (int)(((float)w + 0.5f) * inv);
        r = w - s * len;
        if (r < 0) { --s; r += len; }
        if (r >= len) { ++s; r -= len; }
    };

    for (;;) {
        // ---- (re)initialise fresh slots: initialise_log_domain_bp (bp.hpp:147-157) + syndrome bytes ----
        for (int w = tid; w < S * nnz; w += T) {
            int s, e;
            split(w, nnz, inv_e, s, e);
            if (slot_state[s] == 2 && slot_synd[s] >= 0)
                reinterpret_cast<double *>(slot_base(s))[e] = edge_form<METHOD, MATH>(prior[ci[e]]);
        }
        for (int w = tid; w < S * m; w += T) {
            int s, i;
            split(w, m, inv_m, s, i);
            if (slot_state[s] == 2 && slot_synd[s] >= 0)
                (slot_base(s) + (size_t)nnz * 16 + (size_t)n * 9)[i] = a.synd[slot_synd[s] * m + i];
        }
        __syncthreads();
        if (tid < S && slot_state[tid] == 2) slot_state[tid] = 0;
        if (tid == 0) {
            int act = 0;
            for (int s = 0; s < S; ++s) act += slot_synd[s] >= 0;
            n_active = act;
        }
        __syncthreads();
        if (n_active == 0) break;

        // ---- check pass (bp.hpp:201-273): item = (slot, check) ----
        for (int w = tid; w < S * m; w += T) {
            int s, i;
            split(w, m, inv_m, s, i);
            if (slot_synd[s] < 0) continue;
            double *A = reinterpret_cast<double *>(slot_base(s));
            double *Cm = A + nnz;
            const uint8_t sb = (slot_base(s) + (size_t)nnz * 16 + (size_t)n * 9)[i];
            const int lo = rp[i], hi = rp[i + 1];
            if (METHOD == LDPC_HIP_PRODUCT_SUM) {
                const bool neg = sb != 0;
                double temp = 1.0;
                for (int e = lo; e < hi; ++e) { Cm[e] = temp; temp *= A[e]; }
                temp = 1.0;
                for (int e = hi - 1; e >= lo; --e) {
                    Cm[e] = ps_message<MATH>(Cm[e] * temp, neg, log_tab);
                    temp *= A[e];
                }
            } else {
                const int it = slot_iter[s] + 1;
                const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
                int parity = sb & 1;
                double temp = DBL_MAX;
                for (int e = lo; e < hi; ++e) {
                    const double bk = A[e];
                    if (bk <= 0) parity ^= 1;
                    Cm[e] = temp;
                    const double ab = fabs(bk);
                    if (ab < temp) temp = ab;
                }
                temp = DBL_MAX;
                for (int e = hi - 1; e >= lo; --e) {
                    const double bk = A[e];
                    const int sgn = parity ^ (bk <= 0 ? 1 : 0);
                    double mag = Cm[e];
                    if (temp < mag) mag = temp;
                    Cm[e] = mag * (sgn ? -alpha : alpha);
                    const double ab = fabs(bk);
                    if (ab < temp) temp = ab;
                }
            }
        }
        __syncthreads();

        // ---- bit pass (bp.hpp:276-298, 311-318): item = (slot, bit) ----
        for (int w = tid; w < S * n; w += T) {
            int s, j;
            split(w, n, inv_n, s, j);
            if (slot_synd[s] < 0) continue;
            double *A = reinterpret_cast<double *>(slot_base(s));
            double *Cm = A + nnz;
            double *L = Cm + nnz;
            uint8_t *hard = reinterpret_cast<uint8_t *>(L + n);
            const int lo = cp[j], hi = cp[j + 1];
            double temp = prior[j];
            for (int p = lo; p < hi; ++p) { const int e = ce[p]; A[e] = temp; temp += Cm[e]; }
            L[j] = temp;
            hard[j] = temp <= 0 ? 1 : 0;
            double sfx = 0.0;
            for (int p = hi - 1; p >= lo; --p) {
                const int e = ce[p];
                A[e] = edge_form<METHOD, MATH>(A[e] + sfx);
                sfx += Cm[e];
            }
        }
        __syncthreads();

        // ---- syndrome test (bp.hpp:292-294, 300-302): candidate parity of every check vs its syndrome BYTE ----
        for (int w = tid; w < S * m; w += T) {
            int s, i;
            split(w, m, inv_m, s, i);
            if (slot_synd[s] < 0) continue;
            const uint8_t *hard = slot_base(s) + (size_t)nnz * 16 + (size_t)n * 8;
            const uint8_t sb = (slot_base(s) + (size_t)nnz * 16 + (size_t)n * 9)[i];
            uint8_t par = 0;
            for (int e = rp[i]; e < rp[i + 1]; ++e) par ^= hard[ci[e]];
            if (par != sb) atomicOr(&slot_unsat[s], 1);
        }
        __syncthreads();
        if (tid < S && slot_synd[tid] >= 0) {
            const int it = ++slot_iter[tid];
            if (!slot_unsat[tid] || it >= a.max_iter) slot_state[tid] = 1;
        }
        __syncthreads();

        // ---- finished slots: outputs (bp.hpp:62,65,69,71), then pull the next syndrome ----
        for (int w = tid; w < S * n; w += T) {
            int s, j;
            split(w, n, inv_n, s, j);
            if (slot_state[s] != 1) continue;
            const double *L = reinterpret_cast<const double *>(slot_base(s)) + 2 * (size_t)nnz;
            const uint8_t *hard = reinterpret_cast<const uint8_t *>(L + n);
            const long long b = slot_synd[s];
            a.decoding[b * n + j] = hard[j];
            if (a.llr) a.llr[b * n + j] = L[j];
        }
        __syncthreads();
        if (tid < S) {
            if (slot_state[tid] == 1) {
                const long long b = slot_synd[tid];
                if (a.iters) a.iters[b] = slot_iter[tid];
                if (a.conv) a.conv[b] = slot_unsat[tid] ? 0 : 1;
                const unsigned long long idx = atomicAdd(a.next, 1ull);
                slot_synd[tid] = idx < (unsigned long long)a.batch ? (long long)idx : -1;
                slot_state[tid] = 2;
                slot_iter[tid] = 0;
            }
            slot_unsat[tid] = 0;
        }
        __syncthreads();
    }
}
