// bp_relative_lds_kernel.h -- schedule = serial_relative with the whole decode of a syndrome on chip: one WAVEFRONT per syndrome
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"
#include "bp_relative_kernel.h"

// serial_relative (bp.hpp:451-545 with the re-sort of :469-483) gives every syndrome its own, data-dependent bit order, re-sorted
// with std::sort at the start of every iteration.  bp_serial_relative_kernel (bp_relative_kernel.h) runs that loop per LANE on
// batch-minor arrays in HBM: every access of a wavefront is 64 separate cache lines and the sort is 64 divergent introsorts --
// correct, and 24 - 70 x slower than the fixed-order serial kernel (0.09 M syndromes/s on the d = 21 surface code).  For codes whose
// state fits in LDS (messages 8 nnz + posteriors 8 n + order 2 n + scratch 5 n bytes per syndrome) this kernel instead gives a
// syndrome ONE WAVEFRONT and keeps all of it in wave-private LDS:
//   * the sweep stays sequential over the bits of the lane's ... of the SYNDROME's order (that is what the schedule means); per bit,
//     lane p < column weight owns the bit's p-th check: it multiplies / minimises over the row's other entries in row order (the
//     reference's association, bp.hpp:491-523), the posterior and the two column sweeps (bp.hpp:501-503 / 520-522, 530-534) run as
//     scalar chains over v_readlane values, every lane redoing them, lane p keeping its own bit_to_check;
//   * std::sort is re-enacted IN PARALLEL, result for result: libstdc++'s introsort is (i) median-of-three + an unguarded Hoare
//     partition down to runs of 16, (ii) heapsort when the depth budget is spent, (iii) one final insertion sort.
//       (i)  A partition's outcome is a function of the ORIGINAL arrangement: with L = the positions, ascending, whose key does not
//            precede the pivot's (where the scan from the left stops) and R = the positions, descending, whose key the pivot's does
//            not precede (where the scan from the right stops), the loop swaps L[k] <-> R[k] while L[k] < R[k] -- both scans only
//            ever see untouched elements before they meet -- and returns min(L[K], R[K - 1]) after K swaps.  So: two ballots per 64
//            positions, ranks by mbcnt, K by a count, all swaps at once.
//       (iii) insertion sort is STABLE, and after (i) every run of <= 16 is ordered against its neighbours: the final pass is a stable
//            sort of each run by itself -- every element finds its place by counting inside its run.
//       (ii) (and any NaN key, where the reference's comparator is not a strict weak order) falls back to the sequential restatement
//            of bp_relative_kernel.h on one lane -- the same code the CPU checker pins to the host's real std::sort.
// Same operations on the same operands as the per-lane kernel: same bits (tests/golden/stateful_rel_*.npz, test_stateful_schedules.py).
struct RelLdsArgs {
    int32_t m, n, nnz, max_iter, dc;  // dc: stride of the per-column tables (the heaviest column)
    double ms_scaling_factor;
    int64_t batch;
    const int32_t *row_ptr, *col_idx;   // CSR
    const uint16_t *t_edge;             // [n][dc] CSR edge of the k-th entry of column j (rows ascending); beyond the column's weight: unused
    const uint16_t *t_chk;              // [n][dc] its check
    const uint8_t *t_cdeg;              // [n]
    const int32_t *order0;              // [n] the order every row starts from (the decoder object's serial_schedule_order) or nullptr
    const double *llr0;                 // [n]
    const uint8_t *synd;                // [batch][m]
    uint8_t *decoding;                  // [batch][n]
    double *llr;                        // [batch][n] or nullptr
    int32_t *iters;
    uint8_t *conv;
    int32_t *last_order;                // [n] the order the LAST row of the batch ended with (the object's state after the call)
    unsigned long long *next;           // work counter (zeroed before launch)
    int32_t lds_shared, lds_per_wave;
    unsigned long long *clk;            // shader-clock probe or nullptr
};

__host__ __device__ inline size_t rel_lds_shared(int m, int n, int nnz, int dc, bool product_sum) {
    size_t b = (size_t)n * 8 + (product_sum ? (size_t)n * 8 + 256 * 8 : 0);        // priors, their edge form, log table
    b += ((size_t)(m + 1) * 2 + (size_t)nnz * 2 + 2 * (size_t)n * dc * 2 + (size_t)n + 15) & ~(size_t)15;  // rstart, rcol, t_edge, t_chk, cdeg
    return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t rel_lds_private(int m, int n, int nnz) {
    // A [nnz] f64, L [n] f64, ord [n] u16, tmp [n] u16, posL [n] u16, posR [n] u16, dbit [n] u8, run starts [n] u8, syndrome bytes [m] u8, stack 64 x 3 u16
    size_t b = (size_t)nnz * 8 + (size_t)n * 8 + 4 * (((size_t)n * 2 + 7) & ~(size_t)7) + 2 * (((size_t)n + 7) & ~(size_t)7) + (((size_t)m + 7) & ~(size_t)7) + 64 * 3 * 2;
    return (b + 15) & ~(size_t)15;
}

namespace rel_lds {
typedef __attribute__((address_space(3))) unsigned char l_u8;
typedef __attribute__((address_space(3))) uint16_t l_u16;
typedef __attribute__((address_space(3))) double l_f64;

__device__ __forceinline__ void lds_sync() {  // LDS operations of one wavefront execute in order; this pins the compiler and re-converges the lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double readlane_f64(double x, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), l), __builtin_amdgcn_readlane(__double2loint(x), l));
}

// the sequential restatement (bp_relative_kernel.h: rel_sort) on wave-private LDS arrays, for ONE lane
struct SeqCtx {
    l_u16 *v;
    const l_f64 *key;
    __device__ __forceinline__ int get(long i) const { return v[i]; }
    __device__ __forceinline__ void set(long i, int x) const { v[i] = (uint16_t)x; }
    __device__ __forceinline__ bool gt(int a, int b) const { return key[a] > key[b]; }
    __device__ __forceinline__ void swap(long i, long j) const { const int t = get(i); set(i, get(j)); set(j, t); }
};
}  // namespace rel_lds

namespace rel_sort {
// (the routines of bp_relative_kernel.h are written against `const Ctx &`; the same text serves the LDS context)
template <class CTX>
__device__ inline void push_heap_t(const CTX &x, long first, long hole, long top, int value) {
    long parent = (hole - 1) / 2;
    while (hole > top && x.gt(x.get(first + parent), value)) {
        x.set(first + hole, x.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    x.set(first + hole, value);
}
template <class CTX>
__device__ inline void adjust_heap_t(const CTX &x, long first, long hole, long len, int value) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (x.gt(x.get(first + child), x.get(first + (child - 1)))) child--;
        x.set(first + hole, x.get(first + child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        x.set(first + hole, x.get(first + (child - 1)));
        hole = child - 1;
    }
    push_heap_t(x, first, hole, top, value);
}
template <class CTX>
__device__ inline void heapsort_t(const CTX &x, long first, long last) {  // __partial_sort(first, last, last)
    const long len = last - first;
    if (len >= 2)
        for (long parent = (len - 2) / 2;; parent--) {
            adjust_heap_t(x, first, parent, len, x.get(first + parent));
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        const int value = x.get(last);
        x.set(last, x.get(first));
        adjust_heap_t(x, first, 0, last - first, value);
    }
}
// the whole std::sort, sequentially (NaN keys: as bp_serial_relative_kernel does it, loops bounded by their range)
template <class CTX>
__device__ inline void sort_desc_seq(const CTX &x, long n) {
    if (n <= 0) return;
    int depth = 0;
    for (long q = n; q > 1; q >>= 1) depth++;
    depth *= 2;
    int stack_first[48], stack_last[48], stack_depth[48];
    int sp = 1;
    stack_first[0] = 0; stack_last[0] = (int)n; stack_depth[0] = depth;
    while (sp > 0) {
        --sp;
        long first = stack_first[sp], last = stack_last[sp];
        int d = stack_depth[sp];
        while (last - first > 16) {
            if (d == 0) { heapsort_t(x, first, last); break; }
            --d;
            const long mid = first + (last - first) / 2;
            {   // __move_median_to_first(first, first + 1, mid, last - 1)
                const long a = first + 1, b = mid, c = last - 1;
                const int va = x.get(a), vb = x.get(b), vc = x.get(c);
                if (x.gt(va, vb)) {
                    if (x.gt(vb, vc)) x.swap(first, b);
                    else if (x.gt(va, vc)) x.swap(first, c);
                    else x.swap(first, a);
                } else if (x.gt(va, vc)) x.swap(first, a);
                else if (x.gt(vb, vc)) x.swap(first, c);
                else x.swap(first, b);
            }
            long lo = first + 1, hi = last, f = first + 1, l = last;
            long cut;
            for (;;) {  // __unguarded_partition(first + 1, last, first), bounded as in bp_relative_kernel.h
                const int vp = x.get(first);
                while (f < hi - 1 && x.gt(x.get(f), vp)) ++f;
                --l;
                while (l > lo && x.gt(vp, x.get(l))) --l;
                if (!(f < l)) { cut = f; break; }
                x.swap(f, l);
                ++f;
            }
            stack_first[sp] = (int)cut; stack_last[sp] = (int)last; stack_depth[sp] = d; ++sp;
            last = cut;
        }
    }
    // __final_insertion_sort
    auto linear_insert = [&](long last) {
        const int val = x.get(last);
        long next = last - 1;
        while (next >= 0 && x.gt(val, x.get(next))) {
            x.set(last, x.get(next));
            last = next;
            --next;
        }
        x.set(last, val);
    };
    const long head = n > 16 ? 16 : n;
    for (long i = 1; i < head; ++i) {
        if (x.gt(x.get(i), x.get(0))) {
            const int val = x.get(i);
            for (long q = i; q > 0; --q) x.set(q, x.get(q - 1));
            x.set(0, val);
        } else linear_insert(i);
    }
    for (long i = 16; i < n; ++i) linear_insert(i);
}
}  // namespace rel_sort

namespace rel_lds {
// std::sort(ord, ord + n, [](a, b) { return key[a] > key[b]; }) by one wavefront.  posL / posR / tmp: u16 [n] scratch each;
// runs: u8 [n] (start-of-run marks); stack: u16 [64 * 3].  All wave-private LDS.
__device__ inline void sort_desc_wave(l_u16 *ord, const l_f64 *key, int n, int lane, l_u16 *posL, l_u16 *posR, l_u16 *tmp, l_u8 *runs, l_u16 *stack) {
    if (n <= 1) return;
    // any NaN key: the comparator is no strict weak order and the reference's own result is whatever its loops happen to do -- take the
    // sequential restatement (what the per-lane kernel and the CPU checker run)
    bool nan = false;
    for (int p = lane; p < n; p += 64) { const double k = key[ord[p]]; nan = nan || k != k; }
    if (__builtin_amdgcn_ballot_w64(nan)) {
        if (lane == 0) { SeqCtx cx{ord, key}; rel_sort::sort_desc_seq(cx, n); }
        lds_sync();
        return;
    }
    for (int p = lane; p < n; p += 64) runs[p] = p == 0 ? 1 : 0;
    int depth = 0;
    for (int q = n; q > 1; q >>= 1) depth++;
    depth *= 2;
    int sp = 1;  // (wave-uniform; the entries live in LDS, written by lane 0)
    if (lane == 0) { stack[0] = 0; stack[1] = (uint16_t)n; stack[2] = (uint16_t)depth; }
    lds_sync();
    while (sp > 0) {
        --sp;
        int first = __builtin_amdgcn_readfirstlane((int)stack[sp * 3]), last = __builtin_amdgcn_readfirstlane((int)stack[sp * 3 + 1]), d = __builtin_amdgcn_readfirstlane((int)stack[sp * 3 + 2]);
        while (last - first > 16) {
            if (d == 0) {  // depth budget spent: heapsort of this range (sequential, rare: median-of-three on real posteriors stays balanced)
                if (lane == 0) { SeqCtx cx{ord, key}; rel_sort::heapsort_t(cx, first, last); }
                lds_sync();
                break;
            }
            --d;
            const int mid = first + (last - first) / 2;
            {   // __move_median_to_first(first, first + 1, mid, last - 1): every lane works it out, lane 0 swaps
                const int pa = first + 1, pb = mid, pc = last - 1;
                const int va = ord[pa], vb = ord[pb], vc = ord[pc];
                const double ka = key[va], kb = key[vb], kc = key[vc];
                int pick;
                if (ka > kb) pick = kb > kc ? pb : (ka > kc ? pc : pa);
                else pick = ka > kc ? pa : (kb > kc ? pc : pb);
                pick = __builtin_amdgcn_readfirstlane(pick);
                lds_sync();
                if (lane == 0) { const uint16_t t = ord[first]; ord[first] = ord[pick]; ord[pick] = t; }
                lds_sync();
            }
            // __unguarded_partition(first + 1, last, pivot = first), all at once (header comment)
            const double pk = key[ord[first]];
            int nL = 0, nR = 0;
            for (int c = first + 1; c < last; c += 64) {
                const int p = c + lane;
                const bool valid = p < last;
                const double kv = valid ? key[ord[p]] : 0.0;
                const bool isL = valid && !(kv > pk), isR = valid && !(pk > kv);
                const uint64_t mL = __builtin_amdgcn_ballot_w64(isL), mR = __builtin_amdgcn_ballot_w64(isR);
                if (isL) posL[nL + lane_rank(mL)] = (uint16_t)p;
                if (isR) posR[nR + lane_rank(mR)] = (uint16_t)p;  // ascending here; R[k] = posR[nR - 1 - k]
                nL += __builtin_popcountll(mL);
                nR += __builtin_popcountll(mR);
            }
            lds_sync();
            const int nmin = nL < nR ? nL : nR;
            int K = 0;
            for (int k0 = 0; k0 < nmin; k0 += 64) {
                const int k = k0 + lane;
                const bool sw = k < nmin && posL[k] < posR[nR - 1 - k];
                const uint64_t ms = __builtin_amdgcn_ballot_w64(sw);
                K += __builtin_popcountll(ms);
                if (ms != ~0ull) break;  // L ascends, R descends: once a pair has crossed all later ones have
            }
            for (int k = lane; k < K; k += 64) {
                const int pl = posL[k], pr = posR[nR - 1 - k];
                const uint16_t t = ord[pl];
                ord[pl] = ord[pr];
                ord[pr] = t;
            }
            int cut = K > 0 ? (int)posR[nR - K] : last;          // R[K - 1]: it now holds an element the scan from the left stops at
            if (K < nL && (int)posL[K] < cut) cut = posL[K];
            if (cut > last - 1) cut = last - 1;                  // (cannot happen with ordered keys: the median-of-three leaves a stopper)
            if (cut < first + 1) cut = first + 1;
            cut = __builtin_amdgcn_readfirstlane(cut);
            lds_sync();
            // __introsort_loop(cut, last, d) later; carry on with [first, cut)
            if (lane == 0) { stack[sp * 3] = (uint16_t)cut; stack[sp * 3 + 1] = (uint16_t)last; stack[sp * 3 + 2] = (uint16_t)d; runs[cut] = 1; }
            ++sp;
            last = cut;
            lds_sync();
        }
    }
    // __final_insertion_sort = a stable sort of every run by itself (header comment): count inside the run
    for (int p = lane; p < n; p += 64) {
        int a = p;
        while (!runs[a]) --a;
        int b = p + 1;
        while (b < n && !runs[b]) ++b;
        const int v = ord[p];
        const double kp = key[v];
        int r = a;
        for (int q = a; q < b; ++q) {
            const double kq = key[ord[q]];
            r += (kq > kp || (q < p && !(kp > kq))) ? 1 : 0;  // q comes first: strictly greater key, or an equal key that stood before p
        }
        tmp[r] = (uint16_t)v;
    }
    lds_sync();
    for (int p = lane; p < n; p += 64) ord[p] = tmp[p];
    lds_sync();
}
}  // namespace rel_lds

// DRT: bound of the row loop (heaviest row <= DRT)
template <int METHOD, int MATH, int DRT>
__global__ void __launch_bounds__(1024) bp_relative_lds_kernel(const RelLdsArgs a) {
    using namespace rel_lds;
    extern __shared__ __attribute__((aligned(16))) unsigned char rl_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = a.m, n = a.n, nnz = a.nnz, dc = a.dc;
    constexpr bool PS = METHOD == LDPC_HIP_PRODUCT_SUM;
    __shared__ unsigned long long clk_stamp[2];
    if (tid == 0) clock_probe_begin(clk_stamp);
    // shared: [prior n][pform n][log table 256] (the last two: product-sum) [rstart m + 1][rcol nnz][t_edge n dc][t_chk n dc][cdeg n]
    l_u8 *base = (l_u8 *)rl_lds;
    l_f64 *prior = (l_f64 *)base;
    l_f64 *pform = prior + n;
    l_f64 *log_tab_l = pform + (PS ? n : 0);
    const double *log_tab = reinterpret_cast<const double *>(rl_lds) + (size_t)n + (PS ? (size_t)n : 0);
    l_u16 *rstart = (l_u16 *)(log_tab_l + (PS ? 256 : 0));
    l_u16 *rcol = rstart + (m + 1);
    l_u16 *t_edge = rcol + nnz;
    l_u16 *t_chk = t_edge + (size_t)n * dc;
    l_u8 *cdeg = (l_u8 *)(t_chk + (size_t)n * dc);
    for (int q = tid; q < n; q += T) { prior[q] = a.llr0[q]; cdeg[q] = a.t_cdeg[q]; }
    for (int q = tid; q <= m; q += T) rstart[q] = (uint16_t)a.row_ptr[q];
    for (int q = tid; q < nnz; q += T) rcol[q] = (uint16_t)a.col_idx[q];
    for (int q = tid; q < n * dc; q += T) { t_edge[q] = a.t_edge[q]; t_chk[q] = a.t_chk[q]; }
    if (PS && MATH == 0)
        for (int q = tid; q < 256; q += T) log_tab_l[q] = ldpc_math::k_log_tab[q];
    __syncthreads();
    if (PS)
        for (int q = tid; q < n; q += T) pform[q] = edge_form<METHOD, MATH>(prior[q]);
    __syncthreads();

    // wave-private: [A nnz f64][L n f64][ord][tmp][posL][posR] (u16 n each, padded to 8 bytes) [dbit n u8][runs n u8][sy m u8][stack 64 x 3 u16]
    l_u8 *mine = base + a.lds_shared + (size_t)wave * a.lds_per_wave;
    const int n2 = (n * 2 + 7) & ~7, n1 = (n + 7) & ~7, m1 = (m + 7) & ~7;
    l_f64 *A = (l_f64 *)mine;
    l_f64 *L = A + nnz;
    l_u16 *ord = (l_u16 *)(L + n);
    l_u16 *tmp = (l_u16 *)((l_u8 *)ord + n2);
    l_u16 *posL = (l_u16 *)((l_u8 *)tmp + n2);
    l_u16 *posR = (l_u16 *)((l_u8 *)posL + n2);
    l_u8 *dbit = (l_u8 *)posR + n2;   // hard decisions (a bit the order never visits keeps its 0)
    l_u8 *runs = dbit + n1;            // the sort's run marks
    l_u8 *sy = runs + n1;
    l_u16 *stack = (l_u16 *)(sy + m1);

    for (;;) {
        unsigned long long pulled = 0;
        if (lane == 0) pulled = __hip_atomic_fetch_add(a.next, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int64_t b = ((int64_t)__builtin_amdgcn_readfirstlane((int)(pulled >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)pulled);
        if (b >= a.batch) break;
        bool big = false;
        for (int i = lane; i < m; i += 64) { const uint8_t v = a.synd[b * m + i]; sy[i] = v; big = big || v > 1; }
        const bool never = __builtin_amdgcn_ballot_w64(big) != 0;  // a syndrome byte > 1: cannot converge (bp.hpp:539)
        // initialise_log_domain_bp (bp.hpp:147-157), the starting order, no decision yet
        for (int e = lane; e < nnz; e += 64) A[e] = PS ? pform[rcol[e]] : prior[rcol[e]];
        for (int t = lane; t < n; t += 64) { ord[t] = (uint16_t)(a.order0 ? a.order0[t] : t); L[t] = 0.0; dbit[t] = 0; }
        lds_sync();
        int it = 0;
        bool converged = false;
        while (it < a.max_iter && !converged) {
            ++it;
            const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
            // bp.hpp:469-483: most reliable bits first -- by prior in the first iteration, by the previous posterior afterwards
            sort_desc_wave(ord, it != 1 ? L : prior, n, lane, posL, posR, tmp, runs, stack);
            for (int t = 0; t < n; ++t) {
                const int bit = __builtin_amdgcn_readfirstlane((int)ord[t]);  // (the same address in every lane: a broadcast read)
                const int cd = __builtin_amdgcn_readfirstlane((int)cdeg[bit]);
                const bool mine_p = lane < cd;
                int e = 0;
                double c = 0.0;
                if (mine_p) {
                    e = t_edge[bit * dc + lane];
                    const int chk = t_chk[bit * dc + lane];
                    const int rs = rstart[chk], rd = rstart[chk + 1] - rs;
                    const bool odd = (sy[chk] & 1) != 0;  // pow(-1, syndrome[check]) / sgn = syndrome[check] (bp.hpp:499, 506)
                    if (PS) {  // bp.hpp:491-503
                        double x = 1.0;
#pragma unroll
                        for (int k = 0; k < DRT; ++k)
                            if (k < rd && rs + k != e) x *= A[rs + k];
                        c = ps_message<MATH>(x, odd, log_tab);
                    } else {   // bp.hpp:504-523
                        int sgn = odd ? 1 : 0;
                        double temp = DBL_MAX;
#pragma unroll
                        for (int k = 0; k < DRT; ++k)
                            if (k < rd && rs + k != e) {
                                const double v = A[rs + k];
                                const double ab = fabs(v);
                                if (ab < temp) temp = ab;
                                if (v <= 0) sgn ^= 1;
                            }
                        c = alpha * (sgn ? -1.0 : 1.0) * temp;
                    }
                }
                // the column, top down (bp.hpp:488, 501-503 / 520-522): entry p keeps the running sum before its own message joins it
                double llr = prior[bit], part = 0.0;
                for (int p = 0; p < cd; ++p) {
                    const double cp = readlane_f64(c, p);
                    if (lane == p) part = llr;
                    llr += cp;
                }
                // ... and bottom up (bp.hpp:530-534)
                double sfx = 0.0, b2c = 0.0;
                for (int p = cd - 1; p >= 0; --p) {
                    const double cp = readlane_f64(c, p);
                    if (lane == p) b2c = part + sfx;
                    sfx += cp;
                }
                if (mine_p) A[e] = edge_form<METHOD, MATH>(b2c);
                if (lane == 0) { L[bit] = llr; dbit[bit] = llr <= 0 ? 1 : 0; }  // bp.hpp:525-529
                lds_sync();
            }
            // candidate syndrome of the current hard decision vs the syndrome (bp.hpp:537-543)
            bool differ = false;
            for (int i = lane; i < m; i += 64) {
                unsigned s = 0;
                for (int g = rstart[i]; g < rstart[i + 1]; ++g) s ^= dbit[rcol[g]];
                differ = differ || s != (unsigned)sy[i];
            }
            converged = !never && __builtin_amdgcn_ballot_w64(differ) == 0;
            lds_sync();
        }
        for (int j = lane; j < n; j += 64) {
            a.decoding[b * n + j] = dbit[j];
            if (a.llr) a.llr[b * n + j] = L[j];
        }
        if (lane == 0) {
            if (a.iters) a.iters[b] = it;
            if (a.conv) a.conv[b] = converged ? 1 : 0;
        }
        if (b == a.batch - 1 && a.last_order)
            for (int t = lane; t < n; t += 64) a.last_order[t] = ord[t];
        lds_sync();
    }
    if (tid == 0) clock_probe_end(a.clk, clk_stamp);
}
