// bp_relative_kernel.h -- bp_serial_relative_kernel: schedule = serial_relative (bp.hpp:451-545 with the re-sort of :469-483)
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "bp_device_common.h"

// serial_relative re-sorts the bit order at the start of EVERY iteration -- by descending prior in iteration 1, by descending
// posterior of the previous iteration afterwards -- with std::sort, and keeps the arrangement in the decoder object.  Every
// syndrome therefore walks its own, data-dependent order: the tile-wide kernels (lane = syndrome, ONE bit at a time for all
// 64 lanes) do not apply, and the order is part of the result (std::sort is not stable; with tied keys -- uniform priors make
// ALL keys of iteration 1 equal -- the arrangement is whatever libstdc++'s exact sequence of swaps leaves, and the serial
// sweep's messages depend on it).  So here a LANE runs the reference's loop for its own syndrome with its own order:
//   * messages stay in the batch-minor arrays A / C [tile][edge][64] (lane l reads and writes column l), posteriors in
//     llr_t [tile][n][64], the order in ord [tile][n][64]; when lanes sit at different bits their accesses are gathers
//     (8 bytes of a 512-byte row each), which is what a per-syndrome order costs;
//   * the sort is libstdc++'s std::sort restated operation for operation (introsort: median-of-three quicksort down to runs
//     of 16, heapsort when the recursion budget 2 floor(log2 n) is spent, a final insertion sort) on the lane's column of
//     ord, with the comparator of bp.hpp:472-482.  The CPU checker holds its own restatement, which
//     tests/test_std_sort_port.py pins to the host's real std::sort; tests/golden/stateful_*.npz pin this one to outputs
//     of the real reference (a new decoder object per row, and one object carried over a sequence of decodes).
// Throughput is not the point of this path (the sort alone diverges across the wavefront); it exists so that
// schedule='serial_relative' gives the reference's bits instead of an error.
struct RelArgs {
    int32_t m, n, nnz, max_iter;
    double ms_scaling_factor;
    int64_t batch;
    const int32_t *row_ptr, *col_idx, *col_ptr, *csc_edge, *csc_row;
    const int32_t *order0;        // [n] the order every row starts from (the decoder object's serial_schedule_order)
    const double *llr0;           // [n]
    double *A, *C;                // [tiles][nnz][64]
    double *llr_t;                // [tiles][n][64]
    int32_t *ord;                 // [tiles][n][64]
    uint8_t *dbit;                // [tiles][n][64] hard decision of bit j in lane l
    const uint64_t *par, *invalid;
    uint8_t *decoding;            // [batch][n]
    int32_t *iters;
    uint8_t *conv;
};

namespace rel_sort {
// v(i): element i of this lane's order; key(b): sort key of bit b in this lane.  Comparator comp(a, b) = key(a) > key(b).
struct Ctx {
    int32_t *v;         // lane's column: element i at v[i * 64]
    const double *key;  // lane's column of posteriors (stride 64), or the shared priors (stride 1)
    int kstride;
    __device__ __forceinline__ int get(long i) const { return v[(size_t)i * LDPC_WAVE]; }
    __device__ __forceinline__ void set(long i, int x) const { v[(size_t)i * LDPC_WAVE] = x; }
    __device__ __forceinline__ bool gt(int a, int b) const { return key[(size_t)a * kstride] > key[(size_t)b * kstride]; }
    __device__ __forceinline__ void swap(long i, long j) const { const int t = get(i); set(i, get(j)); set(j, t); }
};

__device__ inline void move_median_to_first(const Ctx &x, long result, long a, long b, long c) {
    const int va = x.get(a), vb = x.get(b), vc = x.get(c);
    if (x.gt(va, vb)) {
        if (x.gt(vb, vc)) x.swap(result, b);
        else if (x.gt(va, vc)) x.swap(result, c);
        else x.swap(result, a);
    } else if (x.gt(va, vc)) x.swap(result, a);
    else if (x.gt(vb, vc)) x.swap(result, c);
    else x.swap(result, b);
}
// The "unguarded" loops of libstdc++ rely on a strict weak order (a sentinel stops them).  With NaN keys (product-sum with
// priors 0 or 1: inf - inf) the comparison is none and the reference's own behaviour is undefined; on the device a walk off
// the lane's range would read another lane's order entry and use it as an index -- so the loops are also bounded by the
// range [lo, hi) they work in.  For keys that ARE ordered the bounds are never reached: nothing observable changes.
__device__ inline long unguarded_partition(const Ctx &x, long first, long last, long pivot) {
    const long lo = first, hi = last;
    for (;;) {
        const int vp = x.get(pivot);  // (the pivot position holds the same element throughout: swaps happen in (pivot, last))
        while (first < hi - 1 && x.gt(x.get(first), vp)) ++first;
        --last;
        while (last > lo && x.gt(vp, x.get(last))) --last;
        if (!(first < last)) return first;
        x.swap(first, last);
        ++first;
    }
}
__device__ inline void push_heap(const Ctx &x, long first, long hole, long top, int value) {
    long parent = (hole - 1) / 2;
    while (hole > top && x.gt(x.get(first + parent), value)) {
        x.set(first + hole, x.get(first + parent));
        hole = parent;
        parent = (hole - 1) / 2;
    }
    x.set(first + hole, value);
}
__device__ inline void adjust_heap(const Ctx &x, long first, long hole, long len, int value) {
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
    push_heap(x, first, hole, top, value);
}
__device__ inline void heapsort(const Ctx &x, long first, long last) {  // __partial_sort(first, last, last)
    const long len = last - first;
    if (len >= 2)
        for (long parent = (len - 2) / 2;; parent--) {
            adjust_heap(x, first, parent, len, x.get(first + parent));
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        const int value = x.get(last);
        x.set(last, x.get(first));
        adjust_heap(x, first, 0, last - first, value);
    }
}
__device__ inline void unguarded_linear_insert(const Ctx &x, long last) {
    const int val = x.get(last);
    long next = last - 1;
    while (next >= 0 && x.gt(val, x.get(next))) {  // (next >= 0: see unguarded_partition)
        x.set(last, x.get(next));
        last = next;
        --next;
    }
    x.set(last, val);
}
__device__ inline void insertion_sort(const Ctx &x, long first, long last) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (x.gt(x.get(i), x.get(first))) {
            const int val = x.get(i);
            for (long q = i; q > first; --q) x.set(q, x.get(q - 1));  // std::move_backward(first, i, i + 1)
            x.set(first, val);
        } else unguarded_linear_insert(x, i);
    }
}
// std::sort(v, v + n, comp) of libstdc++ (bits/stl_algo.h)
__device__ inline void sort_desc(const Ctx &x, long n) {
    if (n <= 0) return;
    int depth = 0;
    for (long q = n; q > 1; q >>= 1) depth++;
    depth *= 2;
    // __introsort_loop recurses into the right part and iterates on the left; the parts are disjoint ranges, so keeping the
    // right parts on an explicit stack changes nothing observable.  Depth budget 2 log2 n <= 46 for n < 2^23.
    int stack_first[48], stack_last[48], stack_depth[48];
    int sp = 1;
    stack_first[0] = 0; stack_last[0] = (int)n; stack_depth[0] = depth;
    while (sp > 0) {
        --sp;
        long first = stack_first[sp], last = stack_last[sp];
        int d = stack_depth[sp];
        while (last - first > 16) {
            if (d == 0) { heapsort(x, first, last); break; }
            --d;
            const long mid = first + (last - first) / 2;
            move_median_to_first(x, first, first + 1, mid, last - 1);
            const long cut = unguarded_partition(x, first + 1, last, first);
            stack_first[sp] = (int)cut; stack_last[sp] = (int)last; stack_depth[sp] = d; ++sp;
            last = cut;
        }
    }
    if (n > 16) {  // __final_insertion_sort
        insertion_sort(x, 0, 16);
        for (long i = 16; i != n; ++i) unguarded_linear_insert(x, i);
    } else insertion_sort(x, 0, n);
}
}  // namespace rel_sort

template <int METHOD, int MATH>
__global__ void __launch_bounds__(64) bp_serial_relative_kernel(const RelArgs a) {
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    const int m = a.m, n = a.n, nnz = a.nnz;
    double *A = a.A + (size_t)tile * (size_t)nnz * LDPC_WAVE + lane;    // this lane's column: entry e at A[e * 64]
    double *Cm = a.C + (size_t)tile * (size_t)nnz * LDPC_WAVE + lane;
    double *L = a.llr_t + (size_t)tile * (size_t)n * LDPC_WAVE + lane;
    int32_t *ord = a.ord + (size_t)tile * (size_t)n * LDPC_WAVE + lane;
    uint8_t *db = a.dbit + (size_t)tile * (size_t)n * LDPC_WAVE + lane;
    const uint64_t *par = a.par + tile * m;
    __shared__ __attribute__((aligned(16))) double log_tab[256];
    if (METHOD == LDPC_HIP_PRODUCT_SUM && MATH == 0)
        for (int q = lane; q < 256; q += 64) log_tab[q] = ldpc_math::k_log_tab[q];
    __builtin_amdgcn_wave_barrier();

    const int64_t b = tile * LDPC_WAVE + lane;
    bool active = b < a.batch;
    const bool never = (a.invalid[tile] >> lane) & 1ull;  // a syndrome byte > 1: cannot converge (bp.hpp:539)
    // initialise_log_domain_bp (bp.hpp:147-157), the starting order, no decision yet
    for (int e = 0; e < nnz; ++e) A[(size_t)e * LDPC_WAVE] = edge_form<METHOD, MATH>(a.llr0[a.col_idx[e]]);
    for (int t = 0; t < n; ++t) {
        ord[(size_t)t * LDPC_WAVE] = a.order0 ? a.order0[t] : t;
        db[(size_t)t * LDPC_WAVE] = 0;
        L[(size_t)t * LDPC_WAVE] = 0.0;
    }
    int my_iter = 0;
    bool converged = false;
    for (int it = 1; it <= a.max_iter; ++it) {
        if (!__builtin_amdgcn_ballot_w64(active)) break;
        const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
        if (active) {
            // bp.hpp:469-483: most reliable bits first
            rel_sort::Ctx cx;
            cx.v = ord;
            cx.key = it != 1 ? L : a.llr0;
            cx.kstride = it != 1 ? LDPC_WAVE : 1;
            rel_sort::sort_desc(cx, n);
            for (int t = 0; t < n; ++t) {
                const int bit = ord[(size_t)t * LDPC_WAVE];
                double llr = a.llr0[bit];  // bp.hpp:488
                const int cs = a.col_ptr[bit], ce = a.col_ptr[bit + 1];
                for (int p = cs; p < ce; ++p) {
                    const int e = a.csc_edge[p], chk = a.csc_row[p];
                    const bool odd = (par[chk] >> lane) & 1ull;
                    double c;
                    if (METHOD == LDPC_HIP_PRODUCT_SUM) {  // bp.hpp:491-503
                        c = 1.0;
                        for (int g = a.row_ptr[chk]; g < a.row_ptr[chk + 1]; ++g)
                            if (g != e) c *= A[(size_t)g * LDPC_WAVE];
                        c = ps_message<MATH>(c, odd, log_tab);
                    } else {  // bp.hpp:504-523
                        int sgn = odd ? 1 : 0;
                        double temp = DBL_MAX;
                        for (int g = a.row_ptr[chk]; g < a.row_ptr[chk + 1]; ++g)
                            if (g != e) {
                                const double v = A[(size_t)g * LDPC_WAVE];
                                const double ab = fabs(v);
                                if (ab < temp) temp = ab;
                                if (v <= 0) sgn ^= 1;
                            }
                        c = alpha * (sgn ? -1.0 : 1.0) * temp;
                    }
                    Cm[(size_t)e * LDPC_WAVE] = c;
                    A[(size_t)e * LDPC_WAVE] = llr;  // partial sum (bp.hpp:501 / 520); completed below before anybody reads it
                    llr += c;
                }
                L[(size_t)bit * LDPC_WAVE] = llr;
                db[(size_t)bit * LDPC_WAVE] = llr <= 0 ? 1 : 0;  // bp.hpp:525-529
                double temp = 0.0;
                for (int p = ce - 1; p >= cs; --p) {  // bp.hpp:530-534
                    const int e = a.csc_edge[p];
                    A[(size_t)e * LDPC_WAVE] = edge_form<METHOD, MATH>(A[(size_t)e * LDPC_WAVE] + temp);
                    temp += Cm[(size_t)e * LDPC_WAVE];
                }
            }
            // candidate syndrome of the current hard decision vs the syndrome (bp.hpp:537-543)
            bool equal = !never;
            for (int i = 0; i < m && equal; ++i) {
                unsigned s = 0;
                for (int g = a.row_ptr[i]; g < a.row_ptr[i + 1]; ++g) s ^= db[(size_t)a.col_idx[g] * LDPC_WAVE];
                if (s != (unsigned)((par[i] >> lane) & 1ull)) equal = false;
            }
            my_iter = it;
            if (equal) { converged = true; active = false; }
        }
    }
    if (b < a.batch) {
        for (int j = 0; j < n; ++j) a.decoding[(size_t)b * n + j] = db[(size_t)j * LDPC_WAVE];
        if (a.iters) a.iters[b] = my_iter;
        if (a.conv) a.conv[b] = converged ? 1 : 0;
    }
}
