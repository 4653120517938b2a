// bp_wave_kernel.h -- bp_wave_kernel: one wavefront decodes one syndrome of a small code out of LDS
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// ---- wavefront-per-syndrome variant for small codes with bounded degrees (BASELINE configs 3 and 5) --------
// Lane = NODE.  A wavefront owns one syndrome: its two message arrays live in a wave-private LDS region, the check
// pass gives lane l the checks l, l+64, ..., the bit pass the bits l, l+64, ...  Nothing but wave-level ordering is
// needed between the passes (LDS operations of a wavefront complete in order), so there is NO workgroup barrier in
// the decode loop and a wavefront that finishes its syndrome pulls the next one from a device-wide counter at once:
// work is proportional to the iterations each syndrome really needs.
//
// Both arrays are kept "structure of arrays" with padded strides mp = roundup(m, 64), np = roundup(n, 64): the k-th
// entry of row i sits at A[k * mp + i], the k-th entry of column j at C[k * np + j], k < DR resp. DC (the template
// bounds).  Every READ of a pass is a unit-stride, conflict-free LDS access with no index lookup; the writes go
// through position tables (u16) built once per decoder.  Rows and columns lighter than the bound, and the padding
// nodes, own PHANTOM entries that hold the neutral element of the pass (min-sum: +DBL_MAX, product-sum: 1.0 for the
// check pass; +0.0 for the bit pass) and whose results are written to a dummy slot, so the min-sum arithmetic runs
// without a single per-entry branch; product-sum only guards its transcendentals.  Neutral elements do not change a
// single bit: min(x, DBL_MAX) = x, x * 1.0 = x, and a partial sum is never -0.0 (priors are log((1-p)/p), never
// -0.0), so x + 0.0 = x.  Per node the entries are walked in the reference's order with the reference's two sweeps
// (bp.hpp:205-218, 278-281 + 313-316): results are bit-identical to every other kernel here and to the reference.
struct WaveArgs {
    int32_t m, n, mp, np, max_iter;
    double ms_scaling_factor;
    int64_t batch;
    const uint8_t *rdeg, *cdeg;  // [mp], [np] node degrees (0 for padding nodes)
    const uint16_t *col;         // [DR * mp] column of the k-th entry of row i at [k * mp + i]; phantom: np
    const uint16_t *cpos;        // [DR * mp] position in C of that entry; phantom: DC * np (the dummy slot)
    const uint16_t *apos;        // [DC * np] position in A of the k-th entry of column j at [k * np + j]; phantom: DR * mp
    const double *llr0;          // [n]
    const uint8_t *synd;         // [batch][m]
    uint8_t *decoding;           // [batch][n]
    double *llr;                 // [batch][n] or nullptr
    int32_t *iters;              // [batch] or nullptr
    uint8_t *conv;               // [batch] or nullptr
    unsigned long long *next;    // device-wide work counter (zeroed before launch)
    int32_t lds_shared, lds_per_wave;  // bytes
};

// LDS bytes: shared tables of a workgroup / private region of one wavefront (host and device agree through these)
__host__ __device__ inline size_t wave_lds_shared(int mp, int np, int DR, int DC) {
    size_t b = 256 * 8 + (size_t)np * 8 + (size_t)(np + 2) * 8 + (size_t)DR * mp * 4 + (size_t)DC * np * 2 + (size_t)mp + (size_t)np;
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t wave_lds_private(int mp, int np, int DR, int DC) {
    size_t b = ((size_t)DR * mp + 2 + (size_t)DC * np + 2) * 8 + (size_t)(np / 64 + 1) * 8 + (size_t)mp;
    return (b + 15) & ~(size_t)15;
}

template <int METHOD, int MATH, int DR, int DC>
__global__ void __launch_bounds__(512) bp_wave_kernel(const WaveArgs a) {
    // nodes per lane in flight: min-sum has few live values per node, the transcendental chains of product-sum many
    constexpr int U = METHOD == LDPC_HIP_MINIMUM_SUM ? (DR <= 4 ? 4 : 2) : (DR <= 6 ? 2 : 1);
    constexpr bool PS = METHOD == LDPC_HIP_PRODUCT_SUM;
    extern __shared__ __attribute__((aligned(16))) unsigned char wv_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, n = a.n, mp = a.mp, np = a.np, rm = DR * mp, cn = DC * np;
    // Every LDS pointer is typed in the LDS address space from the start: generic ("flat") pointers into LDS make this
    // compiler emit null checks against the shared aperture that it then fails to select for some template variants.
    typedef __attribute__((address_space(3))) unsigned char lds_u8;
    typedef __attribute__((address_space(3))) double lds_f64;
    typedef __attribute__((address_space(3))) uint16_t lds_u16;
    typedef __attribute__((address_space(3))) uint64_t lds_u64;
    lds_u8 *base = (lds_u8 *)wv_lds;
    // shared, read-only after the barriers below:
    // [log table][llr0 np][edge form of llr0, np + 2: entry np = neutral][col][cpos][apos][rdeg][cdeg]
    lds_f64 *log_tab_l = (lds_f64 *)base;
    const double *log_tab = reinterpret_cast<const double *>(wv_lds);  // same place, for the math routines' signature
    lds_f64 *prior = log_tab_l + 256;
    lds_f64 *pform = prior + np;
    lds_u16 *col = (lds_u16 *)(pform + np + 2);
    lds_u16 *cpos = col + rm;
    lds_u16 *apos = cpos + rm;
    lds_u8 *rdeg = (lds_u8 *)(apos + cn);
    lds_u8 *cdeg = rdeg + mp;
    for (int q = tid; q < 256; q += T) log_tab_l[q] = ldpc_math::k_log_tab[q];
    for (int q = tid; q < np; q += T) { prior[q] = q < n ? a.llr0[q] : 1.0; cdeg[q] = a.cdeg[q]; }
    for (int q = tid; q < mp; q += T) rdeg[q] = a.rdeg[q];
    for (int q = tid; q < rm; q += T) { col[q] = a.col[q]; cpos[q] = a.cpos[q]; }
    for (int q = tid; q < cn; q += T) apos[q] = a.apos[q];
    __syncthreads();
    for (int q = tid; q < n; q += T) pform[q] = edge_form<METHOD, MATH>(prior[q]);
    if (tid == 0) pform[np] = PS ? 1.0 : DBL_MAX;  // what a phantom entry of A holds
    __syncthreads();

    // wave-private: [A DR*mp + dummy][C DC*np + dummy][hard decisions np/64 + 1 words, the last one zero][syndrome bytes mp]
    lds_u8 *mine = base + a.lds_shared + wave * a.lds_per_wave;
    lds_f64 *A = (lds_f64 *)mine;
    lds_f64 *C = A + rm + 2;
    volatile lds_u64 *hardw = (volatile lds_u64 *)(C + cn + 2);
    volatile lds_u8 *sy = (volatile lds_u8 *)(hardw + np / 64 + 1);
    for (int q = lane; q <= cn; q += 64) C[q] = 0.0;  // phantom entries of C stay +0.0 for good (cpos never points at them)
    for (int q = lane; q < mp; q += 64) sy[q] = 0;
    if (lane == 0) hardw[np / 64] = 0;
    __builtin_amdgcn_wave_barrier();

    for (;;) {
        unsigned long long pulled = 0;
        if (lane == 0) pulled = atomicAdd(a.next, 1ull);
        const int64_t b = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(pulled >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)(pulled & 0xffffffffu)));
        if (b >= a.batch) break;
        // initialise_log_domain_bp (bp.hpp:147-157) + this syndrome's bytes; phantom entries get the neutral element
        for (int i = lane; i < m; i += 64) sy[i] = a.synd[b * m + i];
        for (int q = lane; q < rm; q += 64) A[q] = pform[col[q]];
        __builtin_amdgcn_wave_barrier();

        int it = 0;
        bool unsat_any = true;
        do {
            ++it;
            const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
            // ---- check pass (bp.hpp:201-273), U rows per lane in flight: all loads, then the arithmetic, then the stores ----
            for (int i0 = 0; i0 < mp; i0 += 64 * U) {
                uint8_t sb[U];
                int d[U];
                double cur[U][DR], out[U][DR];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (i0 + u * 64 < mp) {  // wave-uniform
                        const int i = i0 + u * 64 + lane;
                        sb[u] = sy[i];
                        if (PS) d[u] = rdeg[i];
#pragma unroll
                        for (int k = 0; k < DR; ++k) cur[u][k] = A[k * mp + i];
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (i0 + u * 64 < mp) {
                        if (PS) {
                            const bool neg = sb[u] != 0;  // bp.hpp:213
                            double temp = 1.0;
#pragma unroll
                            for (int k = 0; k < DR; ++k) { out[u][k] = temp; temp *= cur[u][k]; }
                            temp = 1.0;
#pragma unroll
                            for (int k = DR - 1; k >= 0; --k) {
                                if (k < d[u]) out[u][k] = ps_message<MATH>(out[u][k] * temp, neg, log_tab);
                                temp *= cur[u][k];
                                LDPC_EDGE_FENCE();
                            }
                        } else {
                            int parity = sb[u] & 1;  // total_sgn = syndrome[i] + #{b2c <= 0}, parity only (bp.hpp:236-262)
                            double temp = DBL_MAX;
#pragma unroll
                            for (int k = 0; k < DR; ++k) {
                                if (cur[u][k] <= 0) parity ^= 1;
                                out[u][k] = temp;
                                const double ab = fabs(cur[u][k]);
                                if (ab < temp) temp = ab;
                            }
                            temp = DBL_MAX;
#pragma unroll
                            for (int k = DR - 1; k >= 0; --k) {
                                const int sgn = parity ^ (cur[u][k] <= 0 ? 1 : 0);
                                double mag = out[u][k];
                                if (temp < mag) mag = temp;
                                out[u][k] = mag * (sgn ? -alpha : alpha);
                                const double ab = fabs(cur[u][k]);
                                if (ab < temp) temp = ab;
                            }
                        }
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (i0 + u * 64 < mp) {
                        const int i = i0 + u * 64 + lane;
#pragma unroll
                        for (int k = 0; k < DR; ++k) C[cpos[k * mp + i]] = out[u][k];  // phantom entries land in the dummy slot
                    }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- bit pass (bp.hpp:276-298, 311-318), U bits per lane in flight ----
            for (int j0 = 0; j0 < np; j0 += 64 * U) {
                int d[U];
                double c[U][DC], pre[U][DC], pr[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j0 + u * 64 < np) {
                        const int j = j0 + u * 64 + lane;
                        pr[u] = prior[j];
                        if (PS) d[u] = cdeg[j];
#pragma unroll
                        for (int k = 0; k < DC; ++k) c[u][k] = C[k * np + j];
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j0 + u * 64 < np) {
                        double temp = pr[u];
#pragma unroll
                        for (int k = 0; k < DC; ++k) { pre[u][k] = temp; temp += c[u][k]; }
                        const uint64_t word = __ballot(temp <= 0);  // padding bits: prior 1.0, no entries -> 0
                        if (lane == 0) hardw[(j0 >> 6) + u] = word;
                        double sfx = 0.0;
#pragma unroll
                        for (int k = DC - 1; k >= 0; --k) {
                            if (PS) {
                                if (k < d[u]) pre[u][k] = edge_form<METHOD, MATH>(pre[u][k] + sfx);
                                LDPC_EDGE_FENCE();
                            } else {
                                pre[u][k] = pre[u][k] + sfx;
                            }
                            sfx += c[u][k];
                        }
                    }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (j0 + u * 64 < np) {
                        const int j = j0 + u * 64 + lane;
#pragma unroll
                        for (int k = 0; k < DC; ++k) A[apos[k * np + j]] = pre[u][k];
                    }
            }
            __builtin_amdgcn_wave_barrier();
            // ---- syndrome test (bp.hpp:292-294, 300-302): candidate parity of every check vs its syndrome BYTE ----
            bool unsat = false;
            for (int i0 = 0; i0 < mp; i0 += 64) {
                const int i = i0 + lane;
                unsigned par = 0;
#pragma unroll
                for (int k = 0; k < DR; ++k) {
                    const int cj = col[k * mp + i];  // phantom: bit np, always zero
                    par ^= (unsigned)(hardw[cj >> 6] >> (cj & 63)) & 1u;
                }
                unsat |= par != (unsigned)sy[i];
            }
            unsat_any = __ballot(unsat) != 0;
        } while (unsat_any && it < a.max_iter);

        // ---- outputs (bp.hpp:62,65,69,71): C still holds this iteration's check->bit messages ----
        for (int j = lane; j < n; j += 64) {
            a.decoding[b * n + j] = (uint8_t)((hardw[j >> 6] >> (j & 63)) & 1ull);
            if (a.llr) {
                double temp = prior[j];
#pragma unroll
                for (int k = 0; k < DC; ++k) temp += C[k * np + j];
                a.llr[b * n + j] = temp;
            }
        }
        if (lane == 0) {
            if (a.iters) a.iters[b] = it;
            if (a.conv) a.conv[b] = unsat_any ? 0 : 1;
        }
        __builtin_amdgcn_wave_barrier();
    }
}
