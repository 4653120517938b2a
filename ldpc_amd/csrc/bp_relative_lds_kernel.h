// bp_relative_lds_kernel.h -- schedule = serial_relative with the whole decode of a syndrome on chip: state in LDS, a wavefront (or a quarter of one) per syndrome
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include <type_traits>

#include "bp_device_common.h"
#include "bp_relative_kernel.h"

// serial_relative (bp.hpp:451-545 with the re-sort of :469-483) gives every syndrome its own, data-dependent bit order, re-sorted
// with std::sort at the start of every iteration.  bp_serial_relative_kernel (bp_relative_kernel.h) runs that loop per LANE on
// batch-minor arrays in HBM: every access of a wavefront is 64 separate cache lines and the sort is 64 divergent introsorts --
// correct, and 24 - 70 x slower than the fixed-order serial kernel (0.09 M syndromes/s on the d = 21 surface code).  For codes whose
// state fits in LDS this kernel keeps a syndrome's messages, posteriors and order in LDS and gives it a GROUP of GS lanes -- the
// whole wavefront (GS = 64, the form host_serial.h picks), or a quarter of it (GS = 16, bit-by-bit product-sum only):
//   * THE SWEEP, LEVEL BY LEVEL (GS = 64, the order a permutation of the bits).  Two bits that share no check commute: side by side
//     each gets exactly the operands the one-after-the-other walk gives it.  So per syndrome and iteration the order is cut into
//     levels -- level(position) = 1 + the highest level among the earlier positions whose bit shares a check with this one -- and a
//     level's bits are updated together, DCT lanes to a bit: lane p owns the p-th entry of the bit's column, forms that check's
//     message from the row's other entries in row order (the reference's association, bp.hpp:491-523), and the posterior and the
//     two column sweeps (bp.hpp:501-503 / 520-522, 530-534) run over the DCT lanes' values by quad moves.  25 levels stand for the 441
//     bit steps of the d = 21 surface code, 33 for BB144's 144;
//   * the sweep, bit by bit (a caller's order with repeated bits, GS = 16, LDPC_HIP_REL_LEVELS=0): per bit, lane p < column weight of
//     the group owns the bit's p-th check; the chains run over the group's values, every lane redoing them;
//   * std::sort is re-enacted IN PARALLEL by the whole wavefront, result for result.  The keys are first replaced by their dense
//     ranks (rank = how many keys are greater: equal keys, equal ranks -- the comparator sees exactly what it saw; a bitonic network in
//     registers up to 512 keys) and packed with the bit number into one word per position, so no step chases a pointer.  libstdc++'s
//     introsort is (i) median-of-three + an unguarded Hoare partition down to runs of 16, (ii) heapsort when the depth budget is
//     spent, (iii) one final insertion sort.
//       (i)  A partition's outcome is a function of the ORIGINAL arrangement: with L = the positions, ascending, whose key does not
//            precede the pivot's (where the scan from the left stops) and R = the positions, descending, whose key the pivot's does
//            not precede (where the scan from the right stops), the loop swaps L[k] <-> R[k] while L[k] < R[k] -- both scans only
//            ever see untouched elements before they meet -- and returns min(L[K], R[K - 1]) after K swaps.  So: two ballots per 64
//            positions, ranks by mbcnt, K by a count, all swaps at once -- through LDS lists for a range of more than 64 places, in
//            registers (a lane per place, stoppers sent to their ranks by lane permutations) below that.
//       (iii) insertion sort is STABLE, and after (i) every run of <= 16 is ordered against its neighbours: the final pass is a stable
//            sort of each run by itself -- a lane takes a run, holds it in registers and places every element by counting.
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
    int32_t levels;                     // 1: the level-parallel sweep (the starting order is a permutation of the bits); 0: bit by bit
    int32_t scratch_in_l;               // 1: most of the sort's and the levels' scratch lives in the syndrome's posterior array (rel_lds_scratch)
    int32_t lds_shared, lds_per_syn, lds_scratch;  // bytes: shared tables of the workgroup / one syndrome's state / one wavefront's sort scratch
    // EXT = 1 (state beyond LDS, round 6): a syndrome's messages A[nnz] and posteriors L[n] live in global memory -- in a slot of its
    // wavefront (A_g, L2 / Infinity Cache resident: written and read by that one wavefront) -- and so do the read-only tables that were the
    // bulk of the shared LDS: the per-entry records (rec_g: [n][dc], the words the kernel otherwise builds in LDS), the priors (llr0) and their
    // edge forms (pform_g).  LDS keeps order, decisions, syndrome, the CSR copy, the log table and the sort's scratch: 23 KB a wavefront on the
    // [[1600,64]] hypergraph-product code -- five to six wavefronts per compute unit where everything in LDS left one (min-sum) or none (product-sum)
    double *A_g;                        // [workgroups][wavefronts][nnz + n]: a wavefront's messages, then its posteriors
    const unsigned long long *rec_g;
    const double *pform_g;              // [n] edge form of the priors (product-sum: tanh(llr0 / 2)); the priors themselves are llr0
    // [n] the order after the FIRST iteration's sort, or nullptr: every row of a call starts from the same order and the first sort's keys are
    // the priors, so its outcome is the same for every row -- worked out once per call (rel_first_order_kernel) instead of once per syndrome
    // (with uniform priors all its keys are equal: the sort whose re-enactment costs most)
    const int32_t *first_order;
    unsigned long long *clk;            // shader-clock probe or nullptr
    unsigned long long *prof;           // nullptr, or 14 words (LDPC_HIP_REL_PROF=1): shader cycles per phase, summed over the wavefronts --
                                        // {refill, sort, levels, sweep, syndrome test, results out}, wavefront-iterations, levels, cycles, wavefronts,
                                        // the sort's {ranks, partitions, final pass}, partitions (those of more than 64 places << 32)
};

// shared by the workgroup: [prior n f64][edge form of the priors n f64, log table 256 f64: product-sum][rec n dc u64][rstart m + 1 u16][rcol nnz u16][cdeg n u8]
__host__ __device__ inline size_t rel_lds_shared(int m, int n, int nnz, int dc, bool product_sum, bool ext = false) {
    size_t b = (ext ? 0 : (size_t)n * 8 + (product_sum ? (size_t)n * 8 : 0) + (size_t)n * dc * 8) + (product_sum ? 256 * 8 : 0);
    b += ((size_t)(m + 1) * 2 + (size_t)nnz * 2 + (size_t)n + 15) & ~(size_t)15;
    return (b + 15) & ~(size_t)15;
}
// one syndrome: [A nnz f64][L n f64][ord n u16][dbit n u8][sy m u8]
__host__ __device__ inline size_t rel_lds_per_syndrome(int m, int n, int nnz, int dc, bool ext = false) {
    size_t b = (ext ? 0 : (size_t)nnz * 8 + (size_t)n * 8) + (((size_t)n * 2 + 7) & ~(size_t)7) + (((size_t)n + 7) & ~(size_t)7) + (((size_t)m + 7) & ~(size_t)7);
    (void)dc;
    return (b + 15) & ~(size_t)15;
}
// one wavefront's sort scratch: [v n u32][posL n u16, posR n u16 -- later tmp n u32 in the same room][rank n u16 -- later the run list n + 1 u16][runs n u8]
// After the sort the same room holds the sweep's levels: [pos n u16][level n u16][list n u16][count / start n + 2 u16].
// in_l: between the moment the sort has the keys' ranks and the sweep, the syndrome's posteriors L[n] (8 n bytes) are dead -- the keys are read,
// and the level-parallel sweep rewrites every bit's posterior -- so everything but v (later: the level lists and counts, which the sweep
// reads while it writes L) can live THERE: posL / posR / tmp, rank / list, runs during the sort, pos and level afterwards.  (n <= 512: the
// ranks of a longer code are counted in passes that read the keys again; host_serial.h decides.)
__host__ __device__ inline size_t rel_lds_scratch(int n, int dc, bool in_l) {
    (void)dc;
    if (in_l) return ((size_t)n * 4 + 8 + 15) & ~(size_t)15;
    const size_t sort_b = 2 * (size_t)n * 4 + (((size_t)(n + 1) * 2 + 7) & ~(size_t)7) + (((size_t)n + 7) & ~(size_t)7);
    const size_t level_b = (size_t)n * 2 * 3 + 4 + (size_t)(n + 2) * 2;
    const size_t b = sort_b > level_b ? sort_b : level_b;
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
template <class KP = const l_f64 *>   // KP: where the keys live -- LDS, or (EXT) global memory
struct SeqCtxT {
    l_u16 *v;
    KP key;
    __device__ __forceinline__ int get(long i) const { return v[i]; }
    __device__ __forceinline__ void set(long i, int x) const { v[i] = (uint16_t)x; }
    __device__ __forceinline__ bool gt(int a, int b) const { return key[a] > key[b]; }
    __device__ __forceinline__ void swap(long i, long j) const { const int t = get(i); set(i, get(j)); set(j, t); }
};
typedef SeqCtxT<> SeqCtx;
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
// the whole std::sort, sequentially (NaN keys: as bp_serial_relative_kernel does it, loops bounded by their range).  `stk`: room for the
// ranges still to do, three 16-bit words each, never more than 2 log2 n + 2 of them (the depth budget) -- in LDS: as a private array
// (3 x 48 ints, indexed by the stack pointer) it was 576 of the kernel's 640 - 800 bytes of scratch per lane, in every instantiation,
// for a path that runs when a key is NaN.
template <class CTX>
__device__ inline void sort_desc_seq(const CTX &x, long n, rel_lds::l_u16 *stk) {
    if (n <= 1) return;  // (nothing to sort -- and a caller's scratch of 4 n bytes would not hold the first range's three words at n = 1)
    int depth = 0;
    for (long q = n; q > 1; q >>= 1) depth++;
    depth *= 2;
    int sp = 1;
    stk[0] = 0; stk[1] = (uint16_t)n; stk[2] = (uint16_t)depth;
    while (sp > 0) {
        --sp;
        long first = stk[3 * sp], last = stk[3 * sp + 1];
        int d = stk[3 * sp + 2];
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
            stk[3 * sp] = (uint16_t)cut; stk[3 * sp + 1] = (uint16_t)last; stk[3 * sp + 2] = (uint16_t)d; ++sp;
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
typedef __attribute__((address_space(3))) uint32_t l_u32;
typedef __attribute__((address_space(3))) unsigned long long l_u64;

// the value of lane ^ LX: data-parallel-primitive moves inside a row of 16 lanes, the LDS crossbar beyond
template <int LX>
__device__ __forceinline__ int lane_xor(int x) {
    if constexpr (LX == 1) return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xf, 0xf, true);       // quad_perm [1, 0, 3, 2]
    else if constexpr (LX == 2) return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xf, 0xf, true);  // quad_perm [2, 3, 0, 1]
    else if constexpr (LX == 4) {
        const int lo = __builtin_amdgcn_update_dpp(0, x, 0x104, 0xf, 0x5, false);           // row_shl:4 -> banks 0, 2 (lanes 0-3, 8-11 of a row)
        return __builtin_amdgcn_update_dpp(lo, x, 0x114, 0xf, 0xa, false);                  // row_shr:4 -> banks 1, 3
    } else if constexpr (LX == 8) return __builtin_amdgcn_mov_dpp(x, 0x128, 0xf, 0xf, true);  // row_ror:8
    else return __shfl_xor(x, LX, 64);
}
// compile-time loop (the permutation controls above are instruction fields: the stage numbers must be constants in the source)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
// stage t of the bitonic network: stretches of k = 2, 4, 8, ... places, strides j = k / 2, k / 4, ..., 1 inside each
constexpr int bitonic_k(int t) { int k = 2, steps = 1; while (t >= steps) { t -= steps; ++steps; k <<= 1; } return k; }
constexpr int bitonic_j(int t) { int k = 2, steps = 1; while (t >= steps) { t -= steps; ++steps; k <<= 1; } return (k >> 1) >> t; }
constexpr int bitonic_stages(int places) { int m = 0; while ((1 << m) < places) ++m; return m * (m + 1) / 2; }

// rank[b] = number of keys greater than key[b] (equal keys, equal ranks), n <= 64 E: a bitonic network over (key, bit) pairs, E consecutive
// places of the sequence per lane (strides < E stay in the lane's registers, the others exchange with lane ^ stride / E), descending; then a
// key's rank is the place where its run of equal keys begins.  Which of two equal keys the network puts first does not matter: they get
// the same rank.  Returns true (in every lane) and leaves `rank` alone if a key is NaN.
// With base0 / sorted: the same for the keys [base0, min(n, base0 + 64 E)) alone -- ranks within that chunk -- and the chunk's keys, descending,
// written to sorted[0 .. its size) (dense_ranks_chunked below).
template <int E, class KP = const l_f64 *>
__device__ __forceinline__ bool dense_ranks(KP key, int n, int lane, l_u16 *rank, int base0 = 0, l_f64 *sorted = nullptr) {
    double kk[E];
    int bb[E];
    bool nan = false;
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const int b = base0 + lane * E + i;
        kk[i] = b < n ? key[b] : -__builtin_inf();  // (padding sorts last; it shares a rank with real -inf keys, after all others either way)
        bb[i] = b < n ? b : 0xffff;
        nan = nan || kk[i] != kk[i];
    }
    if (__builtin_amdgcn_ballot_w64(nan) != 0) return true;
    static_for<0, bitonic_stages(64 * E)>([&](auto tc) {
        constexpr int k = bitonic_k(decltype(tc)::value), j = bitonic_j(decltype(tc)::value);
        if constexpr (j < E) {
#pragma unroll
            for (int i = 0; i < E; ++i) {
                if ((i & j) == 0) {
                    const bool desc = ((lane * E + i) & k) == 0;
                    // the lower place takes the greater key where this stretch runs downwards
                    const bool sw = desc ? !(kk[i] > kk[i | j]) : !(kk[i] < kk[i | j]);
                    const double tk = kk[i];
                    const int tb = bb[i];
                    kk[i] = sw ? kk[i | j] : tk;
                    bb[i] = sw ? bb[i | j] : tb;
                    kk[i | j] = sw ? tk : kk[i | j];
                    bb[i | j] = sw ? tb : bb[i | j];
                }
            }
        } else {
            constexpr int lx = j / E;
            const bool low = (lane & lx) == 0;
#pragma unroll
            for (int i = 0; i < E; ++i) {
                const double ok = __hiloint2double(lane_xor<lx>(__double2hiint(kk[i])), lane_xor<lx>(__double2loint(kk[i])));
                const int ob = lane_xor<lx>(bb[i]);
                const bool desc = ((lane * E + i) & k) == 0;
                const bool want_greater = low == desc;
                const bool take = want_greater ? !(kk[i] > ok) : !(kk[i] < ok);  // (equal keys: both sides take the other's -- a swap)
                kk[i] = take ? ok : kk[i];
                bb[i] = take ? ob : bb[i];
            }
        }
    });
    // rank of place s = s if its key is smaller than the key before it, else the rank of the place before: a running maximum
    const double prev_last = __hiloint2double(__shfl_up(__double2hiint(kk[E - 1]), 1, 64), __shfl_up(__double2loint(kk[E - 1]), 1, 64));
    int rk[E];
#pragma unroll
    for (int i = 0; i < E; ++i) {
        const bool head = i == 0 ? (lane == 0 || prev_last > kk[0]) : kk[i > 0 ? i - 1 : 0] > kk[i];
        const int own = head ? lane * E + i : 0;
        rk[i] = i == 0 ? own : (own > rk[i > 0 ? i - 1 : 0] ? own : rk[i > 0 ? i - 1 : 0]);
    }
    int run = rk[E - 1];
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(run, off, 64); if (lane >= off && o > run) run = o; }
    int carry = __shfl_up(run, 1, 64);
    if (lane == 0) carry = 0;
#pragma unroll
    for (int i = 0; i < E; ++i)
        if (bb[i] != 0xffff) rank[bb[i]] = (uint16_t)(rk[i] > carry ? rk[i] : carry);
    if (sorted) {
        const int real = n - base0 < 64 * E ? n - base0 : 64 * E;  // (the padding sorts behind every real key, equal -inf keys included: the first `real` places)
#pragma unroll
        for (int i = 0; i < E; ++i)
            if (lane * E + i < real) sorted[lane * E + i] = kk[i];
    }
    return false;
}

// the same ranks for any n, in chunks of 512 keys: every chunk goes through the network above (ranks within the chunk, its keys descending
// into `sorted` -- n doubles of scratch), then a key's rank is its rank at home plus, for every other chunk, the number of that chunk's keys
// that are greater: the place a binary search for the key stops in the chunk's descending list.  (Counting against all keys one by one --
// dense_ranks_counting below, what ran until round 6 -- is n^2 / 64 comparisons per lane: 58 % of a wavefront-iteration at n = 1600.)
template <class KP>
__device__ inline bool dense_ranks_chunked(KP key, int n, int lane, l_u16 *rank, l_f64 *sorted) {
    bool nan = false;
    for (int j = lane; j < n; j += 64) nan = nan || key[j] != key[j];
    if (__builtin_amdgcn_ballot_w64(nan) != 0) return true;
    const int chunks = (n + 511) / 512;
    for (int c = 0; c < chunks; ++c) (void)dense_ranks<8, KP>(key, n, lane, rank, c * 512, sorted + c * 512);
    lds_sync();
    for (int b0 = 0; b0 < n; b0 += 256) {  // four keys per lane at a time: their searches are independent chains
        double x[4];
        int home[4], add[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * 64 + lane;
            x[u] = key[b < n ? b : n - 1];
            home[u] = b >> 9;
            add[u] = 0;
        }
        for (int c = 0; c < chunks; ++c) {
            const l_f64 *sc = sorted + c * 512;
            const int size = n - c * 512 < 512 ? n - c * 512 : 512;
            int lo[4], hi[4];  // the answer lies in [lo, hi]: places below lo hold greater keys, places from hi on do not
#pragma unroll
            for (int u = 0; u < 4; ++u) { lo[u] = 0; hi[u] = size; }
            for (int step = 0; step < 10; ++step) {  // (2^10 > 512: every interval has closed)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int mid = (lo[u] + hi[u]) >> 1;
                    const bool open = lo[u] < hi[u];
                    const double km = sc[open ? mid : 0];
                    const bool greater = km > x[u];
                    lo[u] = open && greater ? mid + 1 : lo[u];
                    hi[u] = open && !greater ? mid : hi[u];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) add[u] += c != home[u] ? lo[u] : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int b = b0 + u * 64 + lane;
            if (b < n) rank[b] = (uint16_t)(rank[b] + add[u]);
        }
    }
    return false;
}

// the same ranks by counting, any n: 64 keys compared per broadcast read, eight keys per lane in registers and every key read once per pass
template <class KP>
__device__ inline bool dense_ranks_counting(KP key, int n, int lane, l_u16 *rank) {
    bool nan = false;
    for (int j = lane; j < n; j += 64) nan = nan || key[j] != key[j];
    if (__builtin_amdgcn_ballot_w64(nan) != 0) return true;
    for (int base0 = 0; base0 < n; base0 += 512) {
        double kb[8];
        int cnt[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = base0 + i * 64 + lane;
            kb[i] = b < n ? key[b] : 0.0;
            cnt[i] = 0;
        }
#pragma unroll 2
        for (int j = 0; j < n; ++j) {
            const double kj = key[j];  // (the same address in every lane: a broadcast read)
#pragma unroll
            for (int i = 0; i < 8; ++i) cnt[i] += kj > kb[i] ? 1 : 0;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int b = base0 + i * 64 + lane;
            if (b < n) rank[b] = (uint16_t)cnt[i];
        }
    }
    return false;
}

struct PackedCtx {  // rel_sort's routines on packed words
    l_u32 *w;
    __device__ __forceinline__ int get(long i) const { return (int)w[i]; }
    __device__ __forceinline__ void set(long i, int x) const { w[i] = (uint32_t)x; }
    __device__ __forceinline__ bool gt(int a, int b) const { return ((uint32_t)a >> 16) < ((uint32_t)b >> 16); }
};

// std::sort(ord, ord + n, [](a, b) { return key[a] > key[b]; }) by one wavefront.  Scratch (wave-private LDS): v / tmp u32 [n],
// posL / posR / rank u16 [n], list u16 [n + 1], runs u8 [n] (tmp may share the room of posL + posR; list that of rank).  A word
// of v: (rank of the bit's key << 16) | bit; "x comes before y" (key x > key y) is "rank x < rank y".
template <class KP>
__device__ inline void sort_desc_wave(l_u16 *ord, KP key, int n, int lane, l_u32 *v, l_u32 *tmp, l_u16 *posL, l_u16 *posR, l_u16 *rank,
                                      l_u16 *list, l_u8 *runs, unsigned long long *pfs) {
    if (n <= 1) return;
    unsigned long long pfs_t = pfs ? __builtin_readcyclecounter() : 0;
#define RL_SMARK(k) do { if (pfs) { const unsigned long long now_ = __builtin_readcyclecounter(); pfs[k] += now_ - pfs_t; pfs_t = now_; } } while (0)
    // any NaN key -- the comparator is no strict weak order and the reference's own result is whatever its loops happen to do: the
    // sequential restatement (what the per-lane kernel and the CPU checker run)
    // (n > 512: v and tmp -- 8 n bytes next to each other when the scratch is not housed in the posterior array, which it never is beyond 512
    // bits -- are not yet in use and take the chunks' sorted keys)
    const bool chunked = n > 512 && (l_u8 *)tmp == (l_u8 *)v + (size_t)n * 4 && (n & 1) == 0;
    const bool nan = n <= 256 ? dense_ranks<4, KP>(key, n, lane, rank) : n <= 512 ? dense_ranks<8, KP>(key, n, lane, rank)
                     : chunked ? dense_ranks_chunked<KP>(key, n, lane, rank, (l_f64 *)v) : dense_ranks_counting<KP>(key, n, lane, rank);
    if (nan) {
        if (lane == 0) { SeqCtxT<KP> cx{ord, key}; rel_sort::sort_desc_seq(cx, n, reinterpret_cast<l_u16 *>(v)); }  // (v: 4 n bytes, not yet in use; the stack needs 6 (2 log2 n + 2) for n > 16, 6 below)
        lds_sync();
        return;
    }
    lds_sync();
    // No two elements of the order with equal keys (the usual case for product-sum posteriors from the second iteration on): the comparator
    // is a strict total order on them and ANY correct sort ends with the element of rank r in place r -- no partition needs re-enacting.
    // (Equal keys -- min-sum posteriors, uniform priors, a caller's order with a bit twice -- leave where std::sort's own sequence of swaps
    // leaves them: the re-enactment below.)  Counted per rank in tmp (not yet in use, 4 n bytes).
    {
        for (int p = lane; p < n; p += 64) tmp[p] = 0;
        lds_sync();
        bool tie = false;
        for (int p = lane; p < n; p += 64)
            tie = tie || __hip_atomic_fetch_add(&tmp[rank[ord[p]]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) != 0u;
        if (__builtin_amdgcn_ballot_w64(tie) == 0) {
            lds_sync();
            for (int p = lane; p < n; p += 64) v[p] = ord[p];  // (all of the order read before any of it is rewritten)
            lds_sync();
            for (int p = lane; p < n; p += 64) { const uint32_t bq = v[p]; ord[rank[bq]] = (uint16_t)bq; }
            lds_sync();
            RL_SMARK(0);
            return;
        }
        lds_sync();
    }
    RL_SMARK(0);
    for (int p = lane; p < n; p += 64) { const uint32_t b = ord[p]; v[p] = ((uint32_t)rank[b] << 16) | b; runs[p] = p == 0 ? 1 : 0; }
    int depth = 0;
    for (int q = n; q > 1; q >>= 1) depth++;
    depth *= 2;
    // the ranges still to do: entry i of the stack sits in lane i of two registers (the stack is wave-uniform and never deeper than `depth` < 64)
    int sp = 1, stk_range = n << 16, stk_depth = depth;
    bool in_regs = false;
    uint32_t w_reg = 0, wf_reg = 0;
    lds_sync();
    while (sp > 0) {
        --sp;
        in_regs = false;  // (a range from the stack: its words are in LDS)
        const int range = __builtin_amdgcn_readlane(stk_range, sp);
        int first = range & 0xffff, last = (int)((unsigned)range >> 16), d = __builtin_amdgcn_readlane(stk_depth, sp);
        while (last - first > 16) {
            if (d == 0) {  // depth budget spent: heapsort of this range (sequential, rare: median-of-three on real posteriors stays balanced)
                lds_sync();
                if (lane == 0) {
                    PackedCtx cx{v};
                    rel_sort::heapsort_t(cx, first, last);
                }
                lds_sync();
                break;
            }
            --d;
            const int mid = first + (last - first) / 2;
            const int np = last - first - 1;  // the places of the partition: first + 1 ... last - 1
            int cut;
            if (pfs) { ++pfs[3]; if (np > 64) pfs[3] += 1ull << 32; }
            if (np <= 64) {
                // The whole partition inside one 64-place window: lane i stands for place first + 1 + i and the words stay in registers.  The
                // loop carries on with [first, cut) -- the same window, the same lanes -- so a chain of partitions reads LDS once, at its start.
                if (!in_regs) {
                    lds_sync();
                    w_reg = v[first + 1 + (lane < np ? lane : np - 1)];
                    wf_reg = v[first];
                    in_regs = true;
                }
                // __move_median_to_first(first, first + 1, mid, last - 1)
                const uint32_t wa = (uint32_t)__builtin_amdgcn_readlane((int)w_reg, 0), wb = (uint32_t)__builtin_amdgcn_readlane((int)w_reg, mid - first - 1),
                               wc = (uint32_t)__builtin_amdgcn_readlane((int)w_reg, np - 1), wf = (uint32_t)__builtin_amdgcn_readfirstlane((int)wf_reg);
                const uint32_t ka = wa >> 16, kb = wb >> 16, kc = wc >> 16;  // "key a > key b" = ka < kb
                int pick_l;  // (as a lane of the window)
                if (ka < kb) pick_l = kb < kc ? mid - first - 1 : (ka < kc ? np - 1 : 0);
                else pick_l = ka < kc ? 0 : (kb < kc ? np - 1 : mid - first - 1);
                const uint32_t wp = pick_l == 0 ? wa : pick_l == np - 1 ? wc : wb;
                const uint32_t pk = wp >> 16;
                const uint32_t w_here = lane == pick_l ? wf : w_reg;  // place `pick` holds the old v[first] from here on
                // __unguarded_partition: the stoppers of the scan from the left (L, ascending) and from the right (R, descending) by rank --
                // lane k learns L[k] and R[k] through two lane permutations (stoppers go to their rank, the other lanes fill up behind)
                const bool valid = lane < np;
                const uint32_t kv = w_here >> 16;
                const bool isL = valid && !(kv < pk), isR = valid && !(pk < kv);
                const uint64_t mL = __builtin_amdgcn_ballot_w64(isL), mR = __builtin_amdgcn_ballot_w64(isR);
                const int nL = __builtin_popcountll(mL), nR = __builtin_popcountll(mR);
                const int rL = lane_rank(mL), aboveR = nR - lane_rank(mR) - (isR ? 1 : 0);
                const int Lk = __builtin_amdgcn_ds_permute((isL ? rL : nL + lane - rL) << 2, lane);
                const int Rk = __builtin_amdgcn_ds_permute((isR ? aboveR : nR + (63 - lane) - aboveR) << 2, lane);
                const int nmin = nL < nR ? nL : nR;
                const int K = __builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < nmin && Lk < Rk));  // swap L[k] <-> R[k] while L[k] < R[k]
                const int with_l = __builtin_amdgcn_ds_bpermute(rL << 2, Rk), with_r = __builtin_amdgcn_ds_bpermute(aboveR << 2, Lk);
                const bool swl = isL && rL < K, swr = isR && aboveR < K;  // (never both: the swapped L's all lie left of the swapped R's)
                const uint32_t w_far = (uint32_t)__builtin_amdgcn_ds_bpermute((swl ? with_l : with_r) << 2, (int)w_here);
                const uint32_t w_after = (swl || swr) ? w_far : w_here;
                if (valid) v[first + 1 + lane] = w_after;
                if (lane == 0) v[first] = wp;
                w_reg = w_after;
                wf_reg = wp;
                cut = K > 0 ? first + 1 + __builtin_amdgcn_readlane(Rk, K > 0 ? K - 1 : 0) : last;  // R[K - 1]: it now holds an element the scan from the left stops at
                if (K < nL) { const int c2 = first + 1 + __builtin_amdgcn_readlane(Lk, K < 63 ? K : 63); cut = c2 < cut ? c2 : cut; }
                if (cut > last - 1) cut = last - 1;
                if (cut < first + 1) cut = first + 1;
            } else {
                // __move_median_to_first(first, first + 1, mid, last - 1) and the pivot it leaves at `first`
                const int pa = first + 1, pb = mid, pc = last - 1;
                const uint32_t wa = v[pa], wb = v[pb], wc = v[pc], wf = v[first];
                const uint32_t w_near = v[first + 1 + lane < last ? first + 1 + lane : last - 1];  // (the first 64 places of the partition, on their way with the median's reads)
                const uint32_t ka = wa >> 16, kb = wb >> 16, kc = wc >> 16;  // "key a > key b" = ka < kb
                int pick;
                if (ka < kb) pick = kb < kc ? pb : (ka < kc ? pc : pa);
                else pick = ka < kc ? pa : (kb < kc ? pc : pb);
                pick = __builtin_amdgcn_readfirstlane(pick);
                const uint32_t wp = pick == pa ? wa : pick == pb ? wb : wc;
                const uint32_t pk = __builtin_amdgcn_readfirstlane((int)(wp >> 16));
                // __unguarded_partition(first + 1, last, pivot = first), all at once (header comment); the median swap rides along:
                // position `pick` holds the old v[first] from here on
                int nL = 0, nR = 0;
                for (int c = first + 1; c < last; c += 64) {
                    const int p = c + lane;
                    const bool valid = p < last;
                    const uint32_t kv = (valid ? (p == pick ? wf : c == first + 1 ? w_near : v[p]) : 0u) >> 16;
                    const bool isL = valid && !(kv < pk), isR = valid && !(pk < kv);
                    const uint64_t mL = __builtin_amdgcn_ballot_w64(isL), mR = __builtin_amdgcn_ballot_w64(isR);
                    if (isL) posL[nL + lane_rank(mL)] = (uint16_t)p;
                    if (isR) posR[nR + lane_rank(mR)] = (uint16_t)p;  // ascending here; R[k] = posR[nR - 1 - k]
                    nL += __builtin_popcountll(mL);
                    nR += __builtin_popcountll(mR);
                }
                if (lane == 0) { v[first] = wp; v[pick] = wf; }
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
                    const uint32_t t = v[pl];
                    v[pl] = v[pr];
                    v[pr] = t;
                }
                cut = K > 0 ? (int)posR[nR - K] : last;          // R[K - 1]: it now holds an element the scan from the left stops at
                if (K < nL && (int)posL[K] < cut) cut = posL[K];
                if (cut > last - 1) cut = last - 1;                  // (cannot happen with ordered keys: the median-of-three leaves a stopper)
                if (cut < first + 1) cut = first + 1;
                cut = __builtin_amdgcn_readfirstlane(cut);
                lds_sync();
            }
            // __introsort_loop(cut, last, d) later; carry on with [first, cut)
            if (lane == sp) { stk_range = cut | (last << 16); stk_depth = d; }
            if (lane == 0) runs[cut] = 1;
            ++sp;
            last = cut;
        }
    }
    lds_sync();
    RL_SMARK(1);
    // __final_insertion_sort = a stable sort of every run by itself (header comment).  The runs' starts, listed; then a lane per run:
    // its <= 16 words in registers, every one placed by counting the words that come before it.
    int nruns = 0;
    for (int c = 0; c < n; c += 64) {
        const int p = c + lane;
        const bool st = p < n && runs[p] != 0;
        const uint64_t mk = __builtin_amdgcn_ballot_w64(st);
        if (st) list[nruns + lane_rank(mk)] = (uint16_t)p;
        nruns += __builtin_popcountll(mk);
    }
    if (lane == 0) list[nruns] = (uint16_t)n;
    lds_sync();
    for (int r = lane; r < nruns; r += 64) {
        const int a = list[r], len = (int)list[r + 1] - a;
        if (len > 16) {  // (a heapsorted range: in order already, and insertion sort leaves equal keys where they are)
            for (int i = 0; i < len; ++i) tmp[a + i] = v[a + i];
            continue;
        }
        uint32_t w[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) w[i] = i < len ? v[a + i] : 0xffffffffu;  // (padding: rank 65535, after everything)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            int before = 0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const uint32_t kj = w[j] >> 16, ki = w[i] >> 16;
                before += (kj < ki || (j < i && kj == ki)) ? 1 : 0;  // j comes first: a greater key, or an equal key that stood before i
            }
            if (i < len) tmp[a + before] = w[i];
        }
    }
    lds_sync();
    for (int p = lane; p < n; p += 64) ord[p] = (uint16_t)(tmp[p] & 0xffffu);
    lds_sync();
    RL_SMARK(2);
#undef RL_SMARK
}

template <int CTRL>
__device__ __forceinline__ double quad_move(double x) {  // (two 32-bit DPP moves: 64-bit DPP allows row_newbcast only)
    return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xf, 0xf, true), __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xf, 0xf, true));
}
// out[q] = the value of lane q among the W consecutive lanes this lane belongs to (W = 2, 4: quad permutations; 8: LDS-crossbar reads)
template <int W>
__device__ __forceinline__ void bits_lanes(double x, int lane, double *out) {
    if constexpr (W == 2) {
        out[0] = quad_move<0xA0>(x);  // quad_perm [0, 0, 2, 2]
        out[1] = quad_move<0xF5>(x);  // quad_perm [1, 1, 3, 3]
    } else if constexpr (W == 4) {
        out[0] = quad_move<0x00>(x);
        out[1] = quad_move<0x55>(x);
        out[2] = quad_move<0xAA>(x);
        out[3] = quad_move<0xFF>(x);
    } else {
#pragma unroll
        for (int q = 0; q < W; ++q) {
            const int src = (lane & ~(W - 1)) + q;
            out[q] = __hiloint2double(__shfl(__double2hiint(x), src, 64), __shfl(__double2loint(x), src, 64));
        }
    }
}

template <int GS>
__device__ __forceinline__ double group_lane(double x, int p, int lane) {  // the value of lane p of this lane's group
    if (GS == 64) return readlane_f64(x, p);
    const int src = (lane & ~(GS - 1)) + p;
    return __hiloint2double(__shfl(__double2hiint(x), src, 64), __shfl(__double2loint(x), src, 64));
}
}  // namespace rel_lds

// GS: lanes of a syndrome (64: one per wavefront; 16: four per wavefront).  DRT: bound of the row loop (heaviest row <= DRT).  DCT: bound of
// the per-lane column arrays of the level-parallel sweep (heaviest column <= DCT; GS = 64 only).
#ifndef LDPC_REL_LB
#define LDPC_REL_LB 1024
#endif
// (EXT: at most 8 wavefronts per workgroup -- the codes it is for leave room for 3 to 8 -- so the compiler may use 256 VGPRs: no spills)
template <int METHOD, int MATH, int DRT, int GS, int DCT, int EXT = 0>
__global__ void __launch_bounds__(EXT ? 512 : LDPC_REL_LB) bp_relative_lds_kernel(const RelLdsArgs a) {
    using namespace rel_lds;
    constexpr int G = 64 / GS;
    static_assert(!EXT || GS == 64, "state in global memory: a wavefront per syndrome");
    typedef typename std::conditional<EXT != 0, double, l_f64>::type a_f64;                            // where the messages and posteriors live
    typedef typename std::conditional<EXT != 0, const double, l_f64>::type p_f64;                      // ... and the priors / their edge forms
    typedef typename std::conditional<EXT != 0, const unsigned long long, l_u64>::type rec_u64;       // ... and the records
    // (EXT: a wavefront's stores to A must have completed before its next level's loads -- same wavefront, same L1: a wait, no invalidate)
    auto state_sync = [&]() {
        if (EXT) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        lds_sync();
        if (EXT) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    extern __shared__ __attribute__((aligned(16))) unsigned char rl_lds[];
    const int tid = threadIdx.x, T = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gl = lane & (GS - 1), g = lane / GS;
    const int m = a.m, n = a.n, nnz = a.nnz, dc = a.dc;
    constexpr bool PS = METHOD == LDPC_HIP_PRODUCT_SUM;
    __shared__ unsigned long long clk_stamp[2];
    if (tid == 0) clock_probe_begin(clk_stamp);
    // shared (rel_lds_shared)
    l_u8 *base = (l_u8 *)rl_lds;
    l_f64 *prior_l = (l_f64 *)base;
    l_f64 *pform_l = prior_l + (EXT ? 0 : n);
    l_f64 *log_tab_l = pform_l + (PS && !EXT ? n : 0);
    const double *log_tab = reinterpret_cast<const double *>(rl_lds) + (EXT ? (size_t)0 : (size_t)n + (PS ? (size_t)n : 0));
    p_f64 *prior, *pform;
    if constexpr (EXT != 0) { prior = a.llr0; pform = a.pform_g; } else { prior = prior_l; pform = pform_l; }
    l_u64 *rec_l = (l_u64 *)(log_tab_l + (PS ? 256 : 0));   // per (bit, entry of its column): CSR edge | row start << 16 | row weight << 32 | check << 48
    rec_u64 *rec;
    if constexpr (EXT != 0) rec = a.rec_g; else rec = rec_l;
    l_u16 *rstart = (l_u16 *)(rec_l + (EXT ? 0 : (size_t)n * dc));
    l_u16 *rcol = rstart + (m + 1);
    l_u8 *cdeg = (l_u8 *)(rcol + nnz);
    for (int q = tid; q < n; q += T) { if constexpr (EXT == 0) prior_l[q] = a.llr0[q]; cdeg[q] = a.t_cdeg[q]; }
    for (int q = tid; q <= m; q += T) rstart[q] = (uint16_t)a.row_ptr[q];
    for (int q = tid; q < nnz; q += T) rcol[q] = (uint16_t)a.col_idx[q];
    if constexpr (EXT == 0)
        for (int q = tid; q < n * dc; q += T) {
            const int chk = a.t_chk[q];
            const unsigned long long rs = (unsigned long long)a.row_ptr[chk], rd = (unsigned long long)(a.row_ptr[chk + 1] - a.row_ptr[chk]);
            rec_l[q] = (unsigned long long)a.t_edge[q] | (rs << 16) | (rd << 32) | ((unsigned long long)chk << 48);
        }
    if (PS && MATH == 0)
        for (int q = tid; q < 256; q += T) log_tab_l[q] = ldpc_math::k_log_tab[q];
    __syncthreads();
    if constexpr (EXT == 0)
        if (PS)
            for (int q = tid; q < n; q += T) pform_l[q] = edge_form<METHOD, MATH>(prior_l[q]);
    __syncthreads();

    // this lane's syndrome (rel_lds_per_syndrome) and this wavefront's sort scratch (rel_lds_scratch)
    const int n2 = (n * 2 + 7) & ~7, n1 = (n + 7) & ~7;
    l_u8 *wave_base = base + a.lds_shared + (size_t)wave * ((size_t)G * a.lds_per_syn + a.lds_scratch);
    auto syn_base = [&](int gg) { return wave_base + (size_t)gg * a.lds_per_syn; };
    l_u8 *mine = syn_base(g);
    a_f64 *A, *L;
    l_u16 *ord;
    if constexpr (EXT != 0) {
        A = a.A_g + ((size_t)blockIdx.x * (size_t)(T >> 6) + (size_t)wave) * ((size_t)nnz + (size_t)n);
        L = A + nnz;
        ord = (l_u16 *)mine;
    } else {
        A = (l_f64 *)mine;
        L = A + nnz;
        ord = (l_u16 *)(L + n);
    }
    l_u8 *dbit = (l_u8 *)ord + n2;     // hard decisions (a bit the order never visits keeps its 0)
    l_u8 *sy = dbit + n1;
    l_u8 *scr = wave_base + (size_t)G * a.lds_per_syn;
    const bool in_l = a.scratch_in_l != 0;  // (GS = 64 only: `L` is the wavefront's one syndrome's)
    l_u32 *s_v = (l_u32 *)scr;
    l_u8 *room = scr + (size_t)n * 4;
    if constexpr (EXT == 0) { if (in_l) room = (l_u8 *)L; }
    l_u32 *s_tmp = (l_u32 *)room;                 // (the partitions' position lists are dead when the final pass fills tmp)
    l_u16 *s_posL = (l_u16 *)s_tmp;
    l_u16 *s_posR = s_posL + n;
    l_u16 *s_rank = (l_u16 *)(s_tmp + n);         // (the ranks are dead once v is packed: the run list takes their room)
    l_u16 *s_list = s_rank;
    l_u8 *s_runs = (l_u8 *)s_rank + (((n + 1) * 2 + 7) & ~7);
    (void)n2;

    // Every GROUP draws its own syndromes from the work counter and starts the next one as soon as its current one is done -- a group
    // never waits for the slowest syndrome of its wavefront (at BB144's operating point 4 % of the syndromes run all 50 iterations, the
    // rest ~5).  The wavefront's groups move through iterations in lockstep: sort (the groups that run, one after the other), sweep,
    // syndrome test; a group that has just taken a new syndrome simply is at iteration 1 while its neighbours are further on.
    int64_t b = 0;
    bool have = false, need = true, exhausted = false, never = false, running = false, converged = false;
    int it = 0;
    unsigned long long pfs[4] = {0, 0, 0, 0};
    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pf_t = a.prof ? __builtin_readcyclecounter() : 0;
    const unsigned long long pf_t0 = pf_t;
#define RL_MARK(k) do { if (a.prof) { const unsigned long long now_ = __builtin_readcyclecounter(); pf[k] += now_ - pf_t; pf_t = now_; } } while (0)
    for (int t = gl; t < n; t += GS) ord[t] = 0;  // (the sweep's look-ahead reads the order unconditionally, also in a group that has no syndrome yet)
    lds_sync();
    for (;;) {
        if (need && !exhausted) {  // (whole groups: `need` is the same in every lane of a group)
            unsigned long long pulled = 0;
            if (gl == 0) pulled = __hip_atomic_fetch_add(a.next, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int lead = lane & ~(GS - 1);
            b = ((int64_t)__shfl((int)(pulled >> 32), lead, 64) << 32) | (unsigned)__shfl((int)pulled, lead, 64);
            have = b < a.batch;
            exhausted = !have;
            need = false;
            it = 0;
            converged = false;
            running = false;
            if (have) {
                bool big = false;
                for (int i = gl; i < m; i += GS) { const uint8_t v = a.synd[b * m + i]; sy[i] = v; big = big || v > 1; }
                // a syndrome byte > 1: cannot converge (bp.hpp:539) -- the ballot covers the lanes that are here; a group is here as a whole
                const uint64_t bigm = __builtin_amdgcn_ballot_w64(big);
                never = GS == 64 ? bigm != 0 : ((bigm >> (g * GS)) & ((1ull << (GS & 63)) - 1ull)) != 0;
                // initialise_log_domain_bp (bp.hpp:147-157), the starting order, no decision yet
                for (int e = gl; e < nnz; e += GS) A[e] = PS ? pform[rcol[e]] : prior[rcol[e]];
                for (int t = gl; t < n; t += GS) { ord[t] = (uint16_t)(a.order0 ? a.order0[t] : t); L[t] = 0.0; dbit[t] = 0; }
                running = a.max_iter > 0;
            }
        }
        state_sync();
        RL_MARK(0);
        const uint64_t runm = __builtin_amdgcn_ballot_w64(running);
        if (runm != 0) {
            ++pf[6];
            if (running) ++it;
            const double alpha = (a.ms_scaling_factor == 0.0) ? 1.0 - ldexp(1.0, -it) : a.ms_scaling_factor;
            // bp.hpp:469-483: most reliable bits first -- by prior in the first iteration, by the previous posterior afterwards; the
            // wavefront's running syndromes one after the other, all 64 lanes on each
            for (int gg = 0; gg < G; ++gg) {
                if (!((runm >> (gg * GS)) & 1ull)) continue;
                l_u8 *sb = syn_base(gg);
                a_f64 *Lg;
                l_u16 *og;
                if constexpr (EXT != 0) { Lg = L; og = ord; } else { Lg = (l_f64 *)sb + nnz; og = (l_u16 *)(Lg + n); }
                const int itg = __builtin_amdgcn_readlane(it, gg * GS);
                if (itg == 1 && a.first_order) {
                    for (int t = lane; t < n; t += 64) og[t] = (uint16_t)a.first_order[t];
                    lds_sync();
                    continue;
                }
                sort_desc_wave<const a_f64 *>(og, itg != 1 ? (const a_f64 *)Lg : (const a_f64 *)prior, n, lane, s_v, s_tmp, s_posL, s_posR, s_rank, s_list, s_runs, a.prof ? pfs : nullptr);
            }
            RL_MARK(1);
            if (GS == 64 && a.levels) {
                // ---- the sweep, level by level ------------------------------------------------------------------------------------------
                // Two bits that share no check commute: processing them side by side gives each exactly the operands the bit-by-bit walk
                // gives it.  So the order is cut into LEVELS -- level(t) = 1 + the highest level among the EARLIER positions whose bit
                // shares a check with the bit at t (none: 1) -- and a level's bits go through the update side by side.  The levels depend on
                // the order, i.e. on this syndrome and this iteration: worked out here, in LDS (the sort's scratch is free again).  On the
                // d = 21 surface code 25 levels stand for 441 bit steps; the arithmetic per bit is the bit-by-bit walk's.
                l_u16 *pos = (l_u16 *)scr + n;
                if constexpr (EXT == 0) { if (in_l) pos = (l_u16 *)L; }
                l_u16 *level = pos + n, *llist = (l_u16 *)scr;
                l_u16 *lcnt = in_l ? llist + n + (n & 1) : level + n + (n & 1);  // [n + 2], on a 4-byte boundary: counted with 32-bit atomics, two counts to a word
                l_u32 *lcnt_w = (l_u32 *)lcnt;
                auto count_up = [&](int lv) {  // ++lcnt[lv], returns the old count (counts stay below 65536: no carry into the neighbour)
                    const unsigned old = __hip_atomic_fetch_add(&lcnt_w[lv >> 1], (lv & 1) ? 0x10000u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    return (lv & 1) ? old >> 16 : old & 0xffffu;
                };
                for (int t = lane; t < n; t += 64) pos[ord[t]] = (uint16_t)t;
                for (int q = lane; q < (n + 3) / 2; q += 64) lcnt_w[q] = 0;
                lds_sync();
                // level(t), 64 positions at a time in the order's direction: what a position depends on lies before it, so the earlier
                // chunks' levels are final; chains INSIDE a chunk settle by a few rounds of relaxation among its 64 lanes (lane to lane, no LDS).
                // pr[p]: per entry of the bit's column, the latest earlier position in that entry's row (-1: none) -- all loads of a chunk go out
                // together (rows read to their bound DRT, clamped), and the next chunk's are on their way while this chunk's levels settle.
                auto latest_before = [&](int c, int (&pr)[DCT]) {
                    const int t = c + lane;
                    const int bq = ord[t < n ? t : n - 1], cdq = t < n ? (int)cdeg[bq] : 0;
#pragma unroll
                    for (int p = 0; p < DCT; ++p) {
                        const unsigned long long rq = rec[bq * dc + (p < cdq ? p : 0)];
                        const int rsq = (int)((rq >> 16) & 0xffffu), rdq = (int)((rq >> 32) & 0xffffu), last_k = rdq > 0 ? rdq - 1 : 0;
                        int best = -1;
#pragma unroll
                        for (int k = 0; k < DRT; ++k) {
                            const int q = pos[rcol[rsq + (k < last_k ? k : last_k)]];
                            best = (p < cdq && k < rdq && q < t && q > best) ? q : best;
                        }
                        pr[p] = best;
                    }
                };
                int nlev = 0;
                int pr_next[DCT];
                latest_before(0, pr_next);
                for (int c = 0; c < n; c += 64) {
                    const int t = c + lane;
                    const bool valid = t < n;
                    int pr[DCT];
#pragma unroll
                    for (int p = 0; p < DCT; ++p) pr[p] = pr_next[p];
                    if (c + 64 < n) latest_before(c + 64, pr_next);
                    int lv = 1;
#pragma unroll
                    for (int p = 0; p < DCT; ++p)
                        if (pr[p] >= 0 && pr[p] < c) { const int lq = (int)level[pr[p]] + 1; lv = lq > lv ? lq : lv; }
                    bool inside = false;
#pragma unroll
                    for (int p = 0; p < DCT; ++p) inside = inside || pr[p] >= c;
                    if (__builtin_amdgcn_ballot_w64(inside) != 0)
                        for (;;) {
                            int nl = lv;
#pragma unroll
                            for (int p = 0; p < DCT; ++p) {
                                const int lq = __builtin_amdgcn_ds_bpermute((pr[p] >= c ? pr[p] - c : lane) << 2, lv) + 1;
                                nl = (pr[p] >= c && lq > nl) ? lq : nl;
                            }
                            const bool changed = nl != lv;
                            lv = nl;
                            if (__builtin_amdgcn_ballot_w64(changed) == 0) break;
                        }
                    if (valid) {
                        level[t] = (uint16_t)lv;
                        (void)count_up(lv);
                        nlev = lv > nlev ? lv : nlev;
                    }
                    lds_sync();
                }
                for (int off = 32; off > 0; off >>= 1) { const int o = __shfl_xor(nlev, off, 64); nlev = o > nlev ? o : nlev; }
                nlev = __builtin_amdgcn_readfirstlane(nlev);
                lds_sync();
                // the bits level after level: counts -> where each level starts -> every position takes the next free place of its level
                // (which of a level's bits stands where is of no consequence: they touch disjoint messages)
                {
                    unsigned carry = 0;
                    for (int base_l = 1; base_l <= nlev; base_l += 64) {
                        const int lq = base_l + lane;
                        const unsigned x = lq <= nlev ? lcnt[lq] : 0u;
                        unsigned inc = x;
                        for (int off = 1; off < 64; off <<= 1) { const unsigned o = (unsigned)__shfl_up((int)inc, off, 64); if (lane >= off) inc += o; }
                        if (lq <= nlev) lcnt[lq] = (uint16_t)(carry + inc - x);
                        carry += (unsigned)__shfl((int)inc, 63, 64);
                    }
                }
                lds_sync();
#pragma unroll 4
                for (int t = lane; t < n; t += 64) {
                    const unsigned at = count_up(level[t]);
                    llist[at] = ord[t];
                }
                lds_sync();  // now lcnt[lv] = where level lv ends = where level lv + 1 starts; lcnt[0] = 0
                RL_MARK(2);
                pf[7] += (unsigned)nlev;
                // a level's bits, DCT lanes to a bit: lane p of the bit's lanes owns the p-th entry of its column -- the row product / minimum of
                // that check (bp.hpp:488-503 / 504-522) -- and the two column sweeps run over the lanes' values (quad moves)
                constexpr int BPP = 64 / DCT;
                const int slot = lane / DCT, pl = lane % DCT;
                for (int lv = 1; lv <= nlev; ++lv) {
                    const int i0 = (int)lcnt[lv - 1], i1 = (int)lcnt[lv];
                    for (int i = i0 + slot; i < i1; i += BPP) {  // (a bit's lanes enter and leave together)
                        const int bq = llist[i], cdq = cdeg[bq];
                        const bool mine_p = pl < cdq;
                        const int pc = mine_p ? pl : 0;  // (a lane without an entry reads entry 0's places -- its own table rows are zeroes -- and drops the result)
                        const unsigned long long rq = rec[bq * dc + pc];
                        const int e = (int)(rq & 0xffffu), rs = (int)((rq >> 16) & 0xffffu), rd = (int)((rq >> 32) & 0xffffu);
                        const int odd = sy[rq >> 48] & 1;  // pow(-1, syndrome[check]) / sgn = syndrome[check] (bp.hpp:499, 506)
                        double av[DRT];
                        const int last_k = rd > 0 ? rd - 1 : 0;
#pragma unroll
                        for (int k = 0; k < DRT; ++k) av[k] = A[rs + (k < last_k ? k : last_k)];
                        double c;
                        if (PS) {
                            double x = 1.0;
#pragma unroll
                            for (int k = 0; k < DRT; ++k) x *= (k < rd && rs + k != e) ? av[k] : 1.0;
                            c = ps_message<MATH>(x, odd != 0, log_tab);
                        } else {
                            int sgn = odd;
                            double temp = DBL_MAX;
#pragma unroll
                            for (int k = 0; k < DRT; ++k) {
                                const bool use = k < rd && rs + k != e;
                                const double ab = fabs(av[k]);
                                temp = (use && ab < temp) ? ab : temp;
                                sgn ^= (use && av[k] <= 0) ? 1 : 0;
                            }
                            c = alpha * (sgn ? -1.0 : 1.0) * temp;
                        }
                        double cq[DCT];
                        bits_lanes<DCT>(c, lane, cq);
                        double llr = prior[bq], part = 0.0;
#pragma unroll
                        for (int q = 0; q < DCT; ++q) {  // the column, top down: entry q keeps the running sum before its own message joins it
                            if (pl == q) part = llr;
                            if (q < cdq) llr += cq[q];
                        }
                        double sfx = 0.0, b2c = 0.0;
#pragma unroll
                        for (int q = DCT - 1; q >= 0; --q) {  // ... and bottom up (bp.hpp:530-534)
                            if (pl == q) b2c = part + sfx;
                            if (q < cdq) sfx += cq[q];
                        }
                        if (mine_p) A[e] = edge_form<METHOD, MATH>(b2c);
                        if (pl == 0) { L[bq] = llr; dbit[bq] = llr <= 0 ? 1 : 0; }  // bp.hpp:525-529
                    }
                    state_sync();
                }
            } else {
            // the sweep: every running group walks its own order
            int bit = running ? (int)ord[0] : 0;
            unsigned long long rc = 0;
            int cd = 0, odd = 0;
            if (running) { cd = cdeg[bit]; if (gl < cd) { rc = rec[bit * dc + gl]; odd = sy[rc >> 48] & 1; } }
            for (int t = 0; t < n; ++t) {
                const int bit_next = (int)ord[t + 1 < n ? t + 1 : n - 1];
                const bool mine_p = running && gl < cd;
                const int e = (int)(rc & 0xffffu), rs = (int)((rc >> 16) & 0xffffu), rd = (int)((rc >> 32) & 0xffffu);
                // the row's entries: all DRT loads go out together (clamped to the row: an entry beyond its weight, or the bit's own, is
                // read and then replaced by the neutral element -- no branch, one LDS round trip)
                double c = 0.0;
                {
                    double av[DRT];
                    const int last_k = rd > 0 ? rd - 1 : 0;
#pragma unroll
                    for (int k = 0; k < DRT; ++k) av[k] = A[rs + (k < last_k ? k : last_k)];
                    if (PS) {  // bp.hpp:491-503 (x * 1.0 == x bit for bit, NaN and signed zero included)
                        double x = 1.0;
#pragma unroll
                        for (int k = 0; k < DRT; ++k) x *= (k < rd && rs + k != e) ? av[k] : 1.0;
                        if (mine_p) c = ps_message<MATH>(x, odd != 0, log_tab);
                    } else {   // bp.hpp:504-523
                        int sgn = odd;
                        double temp = DBL_MAX;
#pragma unroll
                        for (int k = 0; k < DRT; ++k) {
                            const bool use = k < rd && rs + k != e;
                            const double ab = fabs(av[k]);
                            temp = (use && ab < temp) ? ab : temp;
                            sgn ^= (use && av[k] <= 0) ? 1 : 0;
                        }
                        c = mine_p ? alpha * (sgn ? -1.0 : 1.0) * temp : 0.0;
                    }
                }
                // what the next bit needs and this bit's messages do not change: on its way while the column is summed up (unconditional
                // reads at clamped places: what a lane without an entry reads is never used)
                const int dcl = gl < dc ? gl : dc - 1;
                const int cd_next = cdeg[bit_next];
                const unsigned long long rc_next = rec[bit_next * dc + dcl];
                const int odd_next = sy[rc_next >> 48] & 1;
                // the column, top down (bp.hpp:488, 501-503 / 520-522): entry p keeps the running sum before its own message joins it
                double llr = running ? prior[bit] : 0.0, part = 0.0;
                for (int p = 0; p < dc; ++p) {
                    const double cp = group_lane<GS>(c, p, lane);
                    if (gl == p) part = llr;
                    if (p < cd) llr += cp;
                }
                // ... and bottom up (bp.hpp:530-534)
                double sfx = 0.0, b2c = 0.0;
                for (int p = dc - 1; p >= 0; --p) {
                    const double cp = group_lane<GS>(c, p, lane);
                    if (gl == p) b2c = part + sfx;
                    if (p < cd) sfx += cp;
                }
                if (mine_p) A[e] = edge_form<METHOD, MATH>(b2c);
                if (running && gl == 0) { L[bit] = llr; dbit[bit] = llr <= 0 ? 1 : 0; }  // bp.hpp:525-529
                state_sync();
                bit = bit_next; rc = rc_next; cd = cd_next; odd = odd_next;
            }
            }
            RL_MARK(3);
            // candidate syndrome of the current hard decision vs the syndrome (bp.hpp:537-543)
            bool differ = false;
            if (running)
                for (int i = gl; i < m; i += GS) {
                    unsigned s = 0;
                    for (int q = rstart[i]; q < rstart[i + 1]; ++q) s ^= dbit[rcol[q]];
                    differ = differ || s != (unsigned)sy[i];
                }
            const uint64_t dm = __builtin_amdgcn_ballot_w64(differ);
            const bool gdiffer = GS == 64 ? dm != 0 : ((dm >> (g * GS)) & ((1ull << (GS & 63)) - 1ull)) != 0;
            if (running) {
                converged = !never && !gdiffer;
                running = !converged && it < a.max_iter;
            }
            RL_MARK(4);
        }
        // groups whose syndrome is done (or never ran: max_iter = 0): results out, next syndrome in
        if (have && !running) {
            for (int j = gl; j < n; j += GS) {
                a.decoding[b * n + j] = dbit[j];
                if (a.llr) a.llr[b * n + j] = L[j];
            }
            if (gl == 0) {
                if (a.iters) a.iters[b] = it;
                if (a.conv) a.conv[b] = converged ? 1 : 0;
            }
            if (b == a.batch - 1 && a.last_order)
                for (int t = gl; t < n; t += GS) a.last_order[t] = ord[t];
            have = false;
            need = true;
        }
        lds_sync();
        RL_MARK(5);
        if (__builtin_amdgcn_ballot_w64(!exhausted) == 0 && __builtin_amdgcn_ballot_w64(have) == 0) break;
    }
#undef RL_MARK
    if (a.prof && lane == 0) {
        for (int k = 0; k < 8; ++k) __hip_atomic_fetch_add(a.prof + k, pf[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.prof + 8, __builtin_readcyclecounter() - pf_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(a.prof + 9, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int k = 0; k < 4; ++k) __hip_atomic_fetch_add(a.prof + 10 + k, pfs[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) clock_probe_end(a.clk, clk_stamp);
}


// The first iteration's sort of a call, once (RelLdsArgs::first_order): order0 sorted by descending prior with the re-enactment above.
// One wavefront; dynamic LDS: [prior n f64][ord n u16 (padded to 8)][the sort's scratch, rel_lds_scratch(n, ., false)].
__global__ void __launch_bounds__(64) rel_first_order_kernel(const double *__restrict__ llr0, const int32_t *__restrict__ order0, int n, int32_t *__restrict__ out) {
    using namespace rel_lds;
    extern __shared__ __attribute__((aligned(16))) unsigned char rl_lds[];
    const int lane = threadIdx.x;
    l_u8 *base = (l_u8 *)rl_lds;
    l_f64 *prior = (l_f64 *)base;
    l_u16 *ord = (l_u16 *)(prior + n);
    l_u8 *scr = (l_u8 *)ord + (((size_t)n * 2 + 15) & ~(size_t)15);
    l_u32 *s_v = (l_u32 *)scr;
    l_u32 *s_tmp = s_v + n;
    l_u16 *s_posL = (l_u16 *)s_tmp, *s_posR = s_posL + n;
    l_u16 *s_rank = (l_u16 *)(s_tmp + n);
    l_u8 *s_runs = (l_u8 *)s_rank + (((n + 1) * 2 + 7) & ~7);
    for (int t = lane; t < n; t += 64) { prior[t] = llr0[t]; ord[t] = (uint16_t)(order0 ? order0[t] : t); }
    lds_sync();
    sort_desc_wave<const l_f64 *>(ord, prior, n, lane, s_v, s_tmp, s_posL, s_posR, s_rank, s_rank, s_runs, nullptr);
    lds_sync();
    for (int t = lane; t < n; t += 64) out[t] = (int32_t)ord[t];
}
