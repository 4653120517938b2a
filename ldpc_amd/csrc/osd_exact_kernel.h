// osd_exact_kernel.h -- OSD on syndromes OUTSIDE the image of H: which rows the reference makes its pivot rows
// Part of libldpc_hip.so (one translation unit: bp_hip.hip includes every kernel header).
#pragma once

#include "osd_kernels.h"

// A rank-deficient H (toric-code checks, a matrix with redundant rows) and a syndrome with a faulty bit: no x solves
// H x = s.  The reference returns the solution of the subsystem of ITS pivot rows, and which rows those are is decided by
// the sparsity heuristic of its linked-list elimination -- among the unpivoted rows with an entry in the pivot column, the
// FIRST IN THE COLUMN'S LINKED LIST of minimal weight(U row) + weight(L row) (gf2sparse_linalg.hpp:149-163 in rref,
// :318-333 in fast_solve).  The list order is history: swap_rows relabels rows without moving their entries
// (sparse_matrix_base.hpp:284-299), insert_entry walks a column from the bottom to the first entry with a smaller row
// label (:449-460), add_rows inserts and removes entry by entry (gf2sparse.hpp:277-305).  The bit-packed eliminations of
// osd_kernels.h do not keep such lists (for a syndrome inside the image the choice of rows cannot change the solution), so
// the rows they flag (status 2) come here: one workgroup re-enacts the reference's elimination for one such syndrome --
// U as bit rows per row OBJECT, labels, per-column object lists with the reference's insert / remove, L as row degrees --
// over the whole column order (an out-of-image syndrome never triggers fast_solve's early stop, so fast_solve and
// rref + lu_solve make the same choices) and writes the syndrome s' that agrees with s on the chosen pivot rows and lies in
// the image of H: s'_i = s_i ^ (what the elimination leaves of s in row i).  H x = s' has the solutions of
// (pivot rows of H) x = (pivot rows of s), so the ordinary OSD kernels, run again on s', return the reference's vector --
// OSD-0 and the higher orders alike (osd.hpp:119-187 solves every candidate on those rows).  Pinned to the real reference,
// directly and through the CPU checker's twin of this routine (tests/test_osd_outside_image.py).
struct OsdExactArgs {
    OsdArgs o;                 // m, n, CSR, synd, llr, list / counters of the first pass
    const uint8_t *status;     // [batch] of the first pass: 2 = outside the image
    uint8_t *corrected;        // [batch][m] s' of the rows handled here
    int32_t *list2;            // the rows handled here, for the second pass
    unsigned *counters2;       // [0] their number
    uint64_t *scratch;         // per workgroup: U [m][hw] words, then the column lists [n][m] u16, the column order [n] i32, list lengths [n] i32
    int64_t slot_words;        // 64-bit words per workgroup
    int32_t hw;                // ceil(n / 64)
};

__host__ __device__ inline size_t osd_exact_slot_words(int m, int n) {
    const size_t hw = ((size_t)n + 63) / 64;
    return (size_t)m * hw + ((size_t)n * (size_t)m * 2 + 7) / 8 + ((size_t)n * 4 + 7) / 8 + ((size_t)n * 4 + 7) / 8;
}
__host__ __device__ inline size_t osd_exact_lds_bytes(int m) { return (size_t)m * 16 + 64; }

__global__ void __launch_bounds__(256) osd_exact_kernel(const OsdExactArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char osd_lds[];
    const OsdArgs &a = A.o;
    const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wave = tid >> 6;
    const int m = a.m, n = a.n, hw = A.hw;
    // LDS: per row object: degree of its U row, of its L row, label, object at a label, what is left of the syndrome, the step's targets
    int *deg = reinterpret_cast<int *>(osd_lds);
    int *ldeg = deg + m;
    uint16_t *label = reinterpret_cast<uint16_t *>(ldeg + m);
    uint16_t *obj_at = label + m;
    uint16_t *targets = obj_at + m;
    uint8_t *yb = reinterpret_cast<uint8_t *>(targets + m);
    __shared__ unsigned long long best_keys[2];  // (by parity of the step: a thread that runs ahead resets the OTHER one)
    __shared__ int n_targets, s_best;
    uint64_t *U = A.scratch + (size_t)blockIdx.x * (size_t)A.slot_words;
    uint16_t *lst = reinterpret_cast<uint16_t *>(U + (size_t)m * hw);
    int32_t *order = reinterpret_cast<int32_t *>(reinterpret_cast<uint64_t *>(lst) + ((size_t)n * (size_t)m * 2 + 7) / 8);
    int32_t *len = reinterpret_cast<int32_t *>(reinterpret_cast<uint64_t *>(order) + ((size_t)n * 4 + 7) / 8);

    const unsigned count = a.counters[0];
    for (unsigned r = blockIdx.x; r < count; r += gridDim.x) {
        const int64_t b = a.list[r];
        if (A.status[b] != 2) continue;  // (block-uniform)
        const double *llr = a.llr + b * n;
        // ---- the reference's objects in their initial state (initialise_LU, gf2sparse_linalg.hpp:72-87) ----
        for (int q = tid; q < m * hw; q += T) U[q] = 0;
        for (int c = tid; c < n; c += T) len[c] = 0;
        __syncthreads();
        for (int i = tid; i < m; i += T) {
            label[i] = obj_at[i] = (uint16_t)i;
            ldeg[i] = 0;
            yb[i] = a.synd[b * m + i] ? 1 : 0;
            int d = 0;
            for (int e = a.row_ptr[i]; e < a.row_ptr[i + 1]; ++e) {
                const int c = a.col_idx[e];
                if ((U[(size_t)i * hw + (c >> 6)] >> (c & 63)) & 1ull) continue;
                U[(size_t)i * hw + (c >> 6)] |= 1ull << (c & 63);
                ++d;
            }
            deg[i] = d;
        }
        // column order: ascending log-ratio, ties by index (soft_decision_col_sort, sort.hpp:35-62) -- a rank sort
        for (int i = tid; i < n; i += T) {
            const double v = llr[i];
            int rk = 0;
            for (int j = 0; j < n; ++j) rk += osd_less(llr[j], j, v, i) ? 1 : 0;
            order[rk] = i;
        }
        __syncthreads();
        // the initial column lists: rows ascend (each thread fills the lists of its columns)
        for (int c = tid; c < n; c += T) {
            int l = 0;
            for (int i = 0; i < m; ++i)
                if ((U[(size_t)i * hw + (c >> 6)] >> (c & 63)) & 1ull) lst[(size_t)c * m + l++] = (uint16_t)i;
            len[c] = l;
        }
        __syncthreads();

        int rank = 0, step = 0;
        const int max_rank = m < n ? m : n;
        for (int t = 0; t < n && rank < max_rank; ++t) {
            const int pc = order[t];
            const int L = len[pc];
            if (L == 0) continue;  // (uniform)
            uint16_t *plist = lst + (size_t)pc * m;
            unsigned long long &best_key = best_keys[step & 1];
            ++step;
            // ---- the pivot row: first in list order of minimal weight among the unpivoted (gf2sparse_linalg.hpp:149-163) ----
            if (tid == 0) best_key = ~0ull;
            __syncthreads();
            unsigned long long mine = ~0ull;
            for (int q = tid; q < L; q += T) {
                const int o = plist[q];
                if ((int)label[o] < rank) continue;
                const unsigned long long key = ((unsigned long long)(unsigned)(deg[o] + ldeg[o]) << 32) | (unsigned)q;
                if (key < mine) mine = key;
            }
            if (mine != ~0ull) atomicMin(&best_key, mine);
            __syncthreads();
            const unsigned long long key = best_key;
            if (key == ~0ull) continue;  // no pivot in this column (uniform; best_key is rewritten only after the next barrier)
            if (tid == 0) {
                const int best = plist[(unsigned)key];
                const int sw = label[best];
                if (sw != rank) {  // swap_rows (:165-172): labels only, the entries stay where they are
                    const int other = obj_at[rank];
                    label[best] = (uint16_t)rank; label[other] = (uint16_t)sw;
                    obj_at[rank] = (uint16_t)best; obj_at[sw] = (uint16_t)other;
                }
                ldeg[best] += 1;  // L.insert_entry(rank, rank) (:174-176)
                s_best = best;
            }
            __syncthreads();
            const int best = s_best;
            // ---- the rows the pivot row is added to, in list order (:178-184) ----
            if (wave == 0) {
                int nt = 0;
                for (int q0 = 0; q0 < L; q0 += 64) {
                    const int q = q0 + lane;
                    const int o = q < L ? (int)plist[q] : 0;
                    const bool take = q < L && (int)label[o] > rank;
                    const uint64_t mask = __ballot(take);
                    if (take) targets[nt + lane_rank(mask)] = (uint16_t)o;
                    nt += __builtin_popcountll(mask);
                }
                if (lane == 0) n_targets = nt;
            }
            __syncthreads();
            const int nt = n_targets;
            // ---- add_rows (gf2sparse.hpp:277-305), column by column of the pivot row: lists of different columns are independent, the
            //      targets of one column are served in order; a thread owns a column ----
            const uint64_t *ub = U + (size_t)best * hw;
            for (int c = tid; c < n; c += T) {
                if (!((ub[c >> 6] >> (c & 63)) & 1ull)) continue;
                uint16_t *Lc = lst + (size_t)c * m;
                int l = len[c];
                for (int k = 0; k < nt; ++k) {
                    const int tg = targets[k];
                    if ((U[(size_t)tg * hw + (c >> 6)] >> (c & 63)) & 1ull) {  // both have it: the entry goes
                        int q = 0;
                        while (Lc[q] != (uint16_t)tg) ++q;
                        for (; q + 1 < l; ++q) Lc[q] = Lc[q + 1];
                        --l;
                        atomicAdd(&deg[tg], -1);
                    } else {  // insert_entry (sparse_matrix_base.hpp:449-474): below the first entry, from the bottom, with a smaller label
                        const int tl = label[tg];
                        int pos = 0;
                        for (int q = l - 1; q >= 0; --q)
                            if ((int)label[Lc[q]] < tl) { pos = q + 1; break; }
                        for (int q = l; q > pos; --q) Lc[q] = Lc[q - 1];
                        Lc[pos] = (uint16_t)tg;
                        ++l;
                        atomicAdd(&deg[tg], 1);
                    }
                }
                len[c] = l;
            }
            __syncthreads();
            // ---- the rows themselves; L.insert_entry(row, rank); the syndrome column ----
            for (int q = tid; q < nt * hw; q += T) {
                const int k = q / hw, w = q - k * hw;
                U[(size_t)targets[k] * hw + w] ^= ub[w];
            }
            for (int k = tid; k < nt; k += T) {
                const int tg = targets[k];
                ldeg[tg] += 1;
                yb[tg] ^= yb[best];
            }
            ++rank;
            __syncthreads();
        }
        // ---- s': s on the pivot rows, the combination of them that the elimination found on the others ----
        for (int i = tid; i < m; i += T)
            A.corrected[b * m + i] = (uint8_t)((a.synd[b * m + i] ? 1 : 0) ^ ((int)label[i] >= rank ? yb[i] : 0));
        if (tid == 0) A.list2[atomicAdd(&A.counters2[0], 1u)] = (int32_t)b;
        __syncthreads();
    }
}
