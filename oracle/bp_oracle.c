/*
 * oracle/bp_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement of the reference's flooding-schedule belief propagation
 * (ldpc::bp::BpDecoder::bp_decode_parallel, /root/reference/src_cpp/bp.hpp:192-325) on flat
 * CSR/CSC arrays instead of the reference's doubly-linked-list matrix.  It performs the SAME
 * floating-point operations in the SAME order as the reference (rows walked in ascending column
 * order, columns in ascending row order -- sparse_matrix_base.hpp:423-482 keeps both sorted), so on
 * one host/libm it must agree with the reference bit for bit, log-probability ratios included.
 * tests/test_oracle_vs_ref.py checks exactly that against oracle/_ref (the real reference headers
 * compiled here); tests/golden/ holds reference outputs the oracle is pinned to everywhere else.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file.
 * The product path (ldpc_amd/csrc) never links or calls it.
 *
 * Build: see oracle/Makefile (gcc -O3 -std=c11 -ffp-contract=off: the reference is built without
 * FMA contraction -- setup.py:66-67 passes only -std=c++2a -O3).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BP_PRODUCT_SUM 0 /* bp.hpp:23-26 */
#define BP_MINIMUM_SUM 1

/* ---- SplitMix64 counter PRNG (twin of ldpc_amd/prng.py) -------------------------------------- */
static inline uint64_t sm64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1u) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

uint64_t oracle_sm64(uint64_t seed, uint64_t idx) { return sm64(seed, idx); }

/*
 * Synthetic BSC batch: error bit (shot, j) = ((sm64(seed, shot*n + j) >> 11) < threshold), and
 * syndrome = H e mod 2 (gf2sparse.hpp:177-214 mulvec; noise model of noise_models/bsc.py:4-23 with
 * a counter-based stream).  errors may be NULL.
 */
void oracle_gen_bsc_syndromes(int m, int n, const int32_t *row_ptr, const int32_t *col_idx,
                              uint64_t seed, uint64_t threshold, int64_t shot0, int64_t shots,
                              uint8_t *syndromes, uint8_t *errors) {
    for (int64_t b = 0; b < shots; b++) {
        uint64_t base = (uint64_t)(shot0 + b) * (uint64_t)n;
        for (int i = 0; i < m; i++) {
            uint8_t s = 0;
            for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++)
                s ^= (uint8_t)((sm64(seed, base + (uint64_t)col_idx[e]) >> 11) < threshold);
            syndromes[b * m + i] = s;
        }
        if (errors)
            for (int j = 0; j < n; j++)
                errors[b * n + j] = (uint8_t)((sm64(seed, base + (uint64_t)j) >> 11) < threshold);
    }
}

/* ---- decoder state --------------------------------------------------------------------------- */
typedef struct {
    int m, n, nnz;
    int32_t *row_ptr, *col_idx; /* CSR, columns ascending inside a row  */
    int32_t *col_ptr, *csc_edge, *csc_row; /* CSC: CSR edge id + row of each entry, rows ascending */
    double *b2c, *c2b;          /* per CSR edge: BpEntry::bit_to_check_msg / check_to_bit_msg (bp.hpp:42-48) */
    double *llr0;               /* initial_log_prob_ratios (bp.hpp:66) */
    uint8_t *cand;              /* candidate_syndrome (bp.hpp:63) */
    /* Optional substitutes for tanh(b/2) and log((1+x)/(1-x)): tests plug the DEVICE math routines
     * (ldpc_amd/csrc/bp_math.h, compiled for the host) in here to predict on the CPU what the HIP
     * kernel computes.  NULL (the default) = the reference's libm expressions, untouched. */
    double (*tanh_half)(double);
    double (*log_ratio)(double);
} bp_oracle;

void bp_oracle_set_math(bp_oracle *o, double (*tanh_half)(double), double (*log_ratio)(double)) {
    o->tanh_half = tanh_half;
    o->log_ratio = log_ratio;
}

void bp_oracle_free(bp_oracle *o) {
    if (!o) return;
    free(o->row_ptr); free(o->col_idx); free(o->col_ptr); free(o->csc_edge); free(o->csc_row);
    free(o->b2c); free(o->c2b); free(o->llr0); free(o->cand); free(o);
}

/* CSR must have sorted, duplicate-free column indices in every row. Returns NULL on bad input. */
bp_oracle *bp_oracle_new(int m, int n, const int32_t *row_ptr, const int32_t *col_idx) {
    if (m < 0 || n < 0 || row_ptr[0] != 0) return NULL;
    int nnz = row_ptr[m];
    for (int i = 0; i < m; i++)
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) {
            if (col_idx[e] < 0 || col_idx[e] >= n) return NULL;
            if (e > row_ptr[i] && col_idx[e] <= col_idx[e - 1]) return NULL;
        }
    bp_oracle *o = (bp_oracle *)calloc(1, sizeof *o);
    o->m = m; o->n = n; o->nnz = nnz;
    o->row_ptr = (int32_t *)malloc(sizeof(int32_t) * (size_t)(m + 1));
    o->col_idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    o->col_ptr = (int32_t *)calloc((size_t)(n + 1), sizeof(int32_t));
    o->csc_edge = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    o->csc_row = (int32_t *)malloc(sizeof(int32_t) * (size_t)(nnz ? nnz : 1));
    o->b2c = (double *)calloc((size_t)(nnz ? nnz : 1), sizeof(double));
    o->c2b = (double *)calloc((size_t)(nnz ? nnz : 1), sizeof(double));
    o->llr0 = (double *)calloc((size_t)(n ? n : 1), sizeof(double));
    o->cand = (uint8_t *)calloc((size_t)(m ? m : 1), 1);
    memcpy(o->row_ptr, row_ptr, sizeof(int32_t) * (size_t)(m + 1));
    memcpy(o->col_idx, col_idx, sizeof(int32_t) * (size_t)nnz);
    for (int e = 0; e < nnz; e++) o->col_ptr[col_idx[e] + 1]++;
    for (int j = 0; j < n; j++) o->col_ptr[j + 1] += o->col_ptr[j];
    int32_t *fill = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n ? n : 1));
    memcpy(fill, o->col_ptr, sizeof(int32_t) * (size_t)n);
    for (int i = 0; i < m; i++) /* ascending i => rows ascending inside each column */
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) {
            int p = fill[col_idx[e]]++;
            o->csc_edge[p] = e;
            o->csc_row[p] = i;
        }
    free(fill);
    return o;
}

/*
 * One decode of one syndrome: bp.hpp:192-325 (PARALLEL schedule), with initialise_log_domain_bp
 * (bp.hpp:147-157) inlined.  Outputs mirror the members the Cython layer reads afterwards:
 * decoding (bp.hpp:62), log_prob_ratios (bp.hpp:65), iterations (bp.hpp:69), converge (bp.hpp:71).
 * If max_iter <= 0 the loop body never runs and the outputs are left untouched, as in the reference.
 */
void bp_oracle_decode(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                      double ms_scaling_factor, const uint8_t *syndrome, uint8_t *decoding,
                      double *log_prob_ratios, int32_t *iterations, uint8_t *converge) {
    const int m = o->m, n = o->n;
    *converge = 0;

    for (int j = 0; j < n; j++) { /* bp.hpp:149-156 */
        o->llr0[j] = log((1 - channel_probs[j]) / channel_probs[j]);
        for (int p = o->col_ptr[j]; p < o->col_ptr[j + 1]; p++) o->b2c[o->csc_edge[p]] = o->llr0[j];
    }

    for (int it = 1; it <= max_iter; it++) {
        if (bp_method == BP_PRODUCT_SUM) { /* bp.hpp:201-219 */
            for (int i = 0; i < m; i++) {
                const int lo = o->row_ptr[i], hi = o->row_ptr[i + 1];
                o->cand[i] = 0;
                double temp = 1.0;
                for (int e = lo; e < hi; e++) {
                    o->c2b[e] = temp;
                    temp *= o->tanh_half ? o->tanh_half(o->b2c[e]) : tanh(o->b2c[e] / 2);
                }
                temp = 1;
                for (int e = hi - 1; e >= lo; e--) {
                    o->c2b[e] *= temp;
                    int message_sign = syndrome[i] != 0u ? -1 : 1;
                    o->c2b[e] = message_sign * (o->log_ratio ? o->log_ratio(o->c2b[e])
                                                             : log((1 + o->c2b[e]) / (1 - o->c2b[e])));
                    temp *= o->tanh_half ? o->tanh_half(o->b2c[e]) : tanh(o->b2c[e] / 2);
                }
            }
        } else { /* bp.hpp:220-273 */
            double alpha;
            if (ms_scaling_factor == 0.0) alpha = 1.0 - pow(2.0, -1.0 * it);
            else alpha = ms_scaling_factor;
            for (int i = 0; i < m; i++) {
                const int lo = o->row_ptr[i], hi = o->row_ptr[i + 1];
                o->cand[i] = 0;
                int total_sgn = syndrome[i];
                double temp = DBL_MAX;
                for (int e = lo; e < hi; e++) {
                    if (o->b2c[e] <= 0) total_sgn += 1;
                    o->c2b[e] = temp;
                    double a = fabs(o->b2c[e]);
                    if (a < temp) temp = a;
                }
                temp = DBL_MAX;
                for (int e = hi - 1; e >= lo; e--) {
                    int sgn = total_sgn;
                    if (o->b2c[e] <= 0) sgn += 1;
                    if (temp < o->c2b[e]) o->c2b[e] = temp;
                    int message_sign = (sgn % 2 == 0) ? 1 : -1;
                    o->c2b[e] *= message_sign * alpha;
                    double a = fabs(o->b2c[e]);
                    if (a < temp) temp = a;
                }
            }
        }

        for (int j = 0; j < n; j++) { /* bp.hpp:276-298 */
            double temp = o->llr0[j];
            for (int p = o->col_ptr[j]; p < o->col_ptr[j + 1]; p++) {
                const int e = o->csc_edge[p];
                o->b2c[e] = temp;
                temp += o->c2b[e];
            }
            log_prob_ratios[j] = temp;
            if (temp <= 0) {
                decoding[j] = 1;
                for (int p = o->col_ptr[j]; p < o->col_ptr[j + 1]; p++) o->cand[o->csc_row[p]] ^= 1;
            } else {
                decoding[j] = 0;
            }
        }

        if (memcmp(o->cand, syndrome, (size_t)m) == 0) *converge = 1; /* bp.hpp:300-302 */
        *iterations = it;
        if (*converge) return;

        for (int j = 0; j < n; j++) { /* bp.hpp:311-318 */
            double temp = 0;
            for (int p = o->col_ptr[j + 1] - 1; p >= o->col_ptr[j]; p--) {
                const int e = o->csc_edge[p];
                o->b2c[e] += temp;
                temp += o->c2b[e];
            }
        }
    }
}

/*
 * Batch wrapper: rows of `syndromes` (shots x m) decoded one after another by the function above
 * (what every caller of the reference does: `for shot: decoder.decode(shot)`).
 * llr may be NULL.
 */
void bp_oracle_decode_batch(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                            double ms_scaling_factor, const uint8_t *syndromes, int64_t shots,
                            uint8_t *decodings, double *llr, int32_t *iterations,
                            uint8_t *converge) {
    double *tmp = llr ? NULL : (double *)malloc(sizeof(double) * (size_t)(o->n ? o->n : 1));
    for (int64_t b = 0; b < shots; b++) {
        iterations[b] = 0;
        bp_oracle_decode(o, channel_probs, max_iter, bp_method, ms_scaling_factor,
                         syndromes + b * o->m, decodings + b * o->n,
                         llr ? llr + b * o->n : tmp, iterations + b, converge + b);
    }
    free(tmp);
}

/* ============================================================================================== *
 * OSD-0 (order-zero ordered-statistics decoding), the post-processing BpOsdDecoder applies when BP
 * does not converge: ldpc::osd::OsdDecoder::decode with osd_order == 0 (src_cpp/osd.hpp:110-117) =
 *   soft_decision_col_sort (sort.hpp:48-62: qsort of (llr, index) ascending by llr, comparator 0 on
 *   ties; glibc's qsort is a stable merge sort for these sizes, so ties keep ascending index), then
 *   RowReduce::fast_solve (gf2sparse_linalg.hpp:298-401): greedy Gaussian elimination taking the
 *   columns in that order, stopping once the syndrome lies in the span of the pivots, and solving on
 *   the pivot columns (lu_solve, :237-288).
 * The result is the unique combination of the greedy-independent column prefix that equals the
 * syndrome; the reference's min-row-weight pivoting (:327-340) only chooses WHICH row carries a
 * pivot and cannot change that solution, so this restatement uses dense bit-packed Gauss-Jordan
 * with first-row pivoting.  Requires the syndrome to be in the image of H (true for any H e).
 * NaN log-ratios make the reference's column order implementation-defined (non-total comparator);
 * here NaN sorts after every number.
 * ============================================================================================== */
/* ============================================================================================== *
 * Syndromes OUTSIDE the image of H (rank-deficient H, e.g. toric-code checks with a measurement error).
 * No x solves H x = s; what the reference returns is the solution of the subsystem of ITS pivot rows, and which rows
 * those are is decided by the sparsity heuristic of its linked-list elimination: among the unpivoted rows with an entry in
 * the pivot column, the FIRST IN THE COLUMN'S LINKED LIST of minimal weight(U row) + weight(L row)
 * (gf2sparse_linalg.hpp:149-163 in rref, :318-333 in fast_solve).  The list order is history: swap_rows relabels rows
 * without moving their entries (sparse_matrix_base.hpp:284-299), insert_entry walks a column from the bottom to the first
 * entry with a smaller row label (:449-460), add_rows inserts / removes entry by entry (gf2sparse.hpp:277-305).  This
 * routine re-enacts exactly that -- U as bit rows per row OBJECT, labels, per-column object lists with the reference's
 * insert and remove, L only as row degrees -- over the whole column order (an out-of-image syndrome never triggers
 * fast_solve's early stop, so fast_solve and rref + lu_solve make the same choices), and returns the syndrome s' that
 * agrees with s on the chosen pivot rows and lies in the image of H: s'_i = s_i ^ (what is left of s in row i after the
 * elimination).  H x = s' has the same solutions as (pivot rows of H) x = (pivot rows of s), so every ordinary OSD routine
 * run on s' returns the reference's vector -- OSD-0 and the higher orders alike (osd.hpp:119-187 solves every candidate
 * on those rows).  Checked against the real reference: tests/test_osd_outside_image.py.
 * ============================================================================================== */
void osd_reference_pivot_rows_syndrome(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const int *order,
                                       const uint8_t *syndrome, uint8_t *corrected) {
    const int hw = (n + 63) / 64;
    uint64_t *U = (uint64_t *)calloc((size_t)(m ? m : 1) * (size_t)(hw ? hw : 1), sizeof(uint64_t));
    int *deg = (int *)calloc((size_t)(m ? m : 1), sizeof(int)), *ldeg = (int *)calloc((size_t)(m ? m : 1), sizeof(int));
    int *label = (int *)malloc(sizeof(int) * (size_t)(m ? m : 1)), *obj_at = (int *)malloc(sizeof(int) * (size_t)(m ? m : 1));
    int *len = (int *)calloc((size_t)(n ? n : 1), sizeof(int));
    int *lst = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1) * (size_t)(m ? m : 1));  /* column c: lst[c * m ..], top to bottom */
    int *targets = (int *)malloc(sizeof(int) * (size_t)(m ? m : 1));
    uint8_t *yb = (uint8_t *)malloc((size_t)(m ? m : 1));
    for (int i = 0; i < m; i++) {
        label[i] = obj_at[i] = i;
        yb[i] = syndrome[i] ? 1 : 0;
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) {
            const int c = col_idx[e];
            if ((U[(size_t)i * hw + c / 64] >> (c % 64)) & 1) continue;
            U[(size_t)i * hw + c / 64] |= 1ull << (c % 64);
            deg[i]++;
            lst[(size_t)c * m + len[c]++] = i;  /* rows ascend: the initial lists are sorted */
        }
    }
    int rank = 0;
    const int max_rank = m < n ? m : n;
    for (int t = 0; t < n && rank < max_rank; t++) {
        const int pc = order[t];
        int best = -1, bw = 0;
        for (int q = 0; q < len[pc]; q++) {
            const int o = lst[(size_t)pc * m + q];
            if (label[o] < rank) continue;
            const int w = deg[o] + ldeg[o];
            if (best < 0 || w < bw) { best = o; bw = w; }
        }
        if (best < 0) continue;
        const int sw = label[best];
        if (sw != rank) {
            const int other = obj_at[rank];
            label[best] = rank; label[other] = sw;
            obj_at[rank] = best; obj_at[sw] = other;
        }
        ldeg[best]++;  /* L.insert_entry(rank, rank) */
        int nt = 0;
        for (int q = 0; q < len[pc]; q++) {
            const int o = lst[(size_t)pc * m + q];
            if (label[o] > rank) targets[nt++] = o;
        }
        for (int k = 0; k < nt; k++) {
            const int tg = targets[k], tl = label[tg];
            for (int w = 0; w < hw; w++) {
                uint64_t bits = U[(size_t)best * hw + w];
                while (bits) {
                    const int c = w * 64 + __builtin_ctzll(bits);
                    bits &= bits - 1;
                    int *L = lst + (size_t)c * m;
                    if ((U[(size_t)tg * hw + w] >> (c % 64)) & 1) {  /* remove */
                        int q = 0;
                        while (L[q] != tg) q++;
                        for (; q + 1 < len[c]; q++) L[q] = L[q + 1];
                        len[c]--;
                        deg[tg]--;
                    } else {  /* insert below the first entry, from the bottom, whose label is smaller */
                        int pos = 0;
                        for (int q = len[c] - 1; q >= 0; q--)
                            if (label[L[q]] < tl) { pos = q + 1; break; }
                        for (int q = len[c]; q > pos; q--) L[q] = L[q - 1];
                        L[pos] = tg;
                        len[c]++;
                        deg[tg]++;
                    }
                }
            }
            for (int w = 0; w < hw; w++) U[(size_t)tg * hw + w] ^= U[(size_t)best * hw + w];
            ldeg[tg]++;  /* L.insert_entry(row, rank) */
            yb[tg] ^= yb[best];
        }
        rank++;
    }
    for (int i = 0; i < m; i++) corrected[i] = (uint8_t)((syndrome[i] ? 1 : 0) ^ (label[i] >= rank ? yb[i] : 0));
    free(U); free(deg); free(ldeg); free(label); free(obj_at); free(len); free(lst); free(targets); free(yb);
}

static int osd_less(double a, int ia, double b, int ib) {
    const int na = a != a, nb = b != b;
    if (na || nb) return na == nb ? ia < ib : nb; /* numbers before NaNs */
    if (a < b) return 1;
    if (a > b) return 0;
    return ia < ib;
}

/* 1 if H x = syndrome has a solution (plain bit-packed elimination of [H | s]) */
static int osd_syndrome_in_image(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const uint8_t *syndrome) {
    const int words = (n + 1 + 63) / 64;
    uint64_t *a = (uint64_t *)calloc((size_t)(m ? m : 1) * (size_t)words, sizeof(uint64_t));
    for (int i = 0; i < m; i++) {
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) a[(size_t)i * words + col_idx[e] / 64] |= 1ull << (col_idx[e] % 64);
        if (syndrome[i]) a[(size_t)i * words + n / 64] |= 1ull << (n % 64);
    }
    int rank = 0;
    for (int c = 0; c < n && rank < m; c++) {
        int p = -1;
        for (int i = rank; i < m; i++)
            if ((a[(size_t)i * words + c / 64] >> (c % 64)) & 1) { p = i; break; }
        if (p < 0) continue;
        for (int w = 0; w < words; w++) { const uint64_t t = a[(size_t)p * words + w]; a[(size_t)p * words + w] = a[(size_t)rank * words + w]; a[(size_t)rank * words + w] = t; }
        for (int i = rank + 1; i < m; i++)
            if ((a[(size_t)i * words + c / 64] >> (c % 64)) & 1)
                for (int w = 0; w < words; w++) a[(size_t)i * words + w] ^= a[(size_t)rank * words + w];
        rank++;
    }
    int ok = 1;
    for (int i = rank; i < m; i++)
        if ((a[(size_t)i * words + n / 64] >> (n % 64)) & 1) ok = 0;
    free(a);
    return ok;
}

/* the syndrome the ordinary routines below are run on: the caller's, or -- outside the image of H -- the one that keeps
 * the reference's pivot rows (osd_reference_pivot_rows_syndrome); returns a malloc'ed copy */
static uint8_t *osd_effective_syndrome(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *llr,
                                       const uint8_t *syndrome) {
    uint8_t *eff = (uint8_t *)malloc((size_t)(m ? m : 1));
    for (int i = 0; i < m; i++) eff[i] = syndrome[i] ? 1 : 0;
    if (osd_syndrome_in_image(m, n, row_ptr, col_idx, syndrome)) return eff;
    int *order = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int j = 0; j < n; j++) {
            const double a = llr[j], b = llr[i];
            const int na = a != a, nb = b != b;
            r += (na || nb) ? (na == nb ? j < i : nb) : (a < b ? 1 : a > b ? 0 : j < i);
        }
        order[r] = i;
    }
    osd_reference_pivot_rows_syndrome(m, n, row_ptr, col_idx, order, syndrome, eff);
    free(order);
    return eff;
}

static void osd0_oracle_in_image(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *llr,
                                 const uint8_t *syndrome, uint8_t *decoding);
void osd0_oracle(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *llr,
                 const uint8_t *syndrome, uint8_t *decoding) {
    uint8_t *eff = osd_effective_syndrome(m, n, row_ptr, col_idx, llr, syndrome);
    osd0_oracle_in_image(m, n, row_ptr, col_idx, llr, eff, decoding);
    free(eff);
}

static void osd0_oracle_in_image(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *llr,
                                 const uint8_t *syndrome, uint8_t *decoding) {
    const int words = (n + 1 + 63) / 64; /* n matrix bits + 1 augmented syndrome bit per row */
    uint64_t *a = (uint64_t *)calloc((size_t)(m ? m : 1) * (size_t)words, sizeof(uint64_t));
    int *order = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    int *pivot_col = (int *)malloc(sizeof(int) * (size_t)(m ? m : 1));
    for (int i = 0; i < m; i++) {
        pivot_col[i] = -1;
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) a[(size_t)i * words + col_idx[e] / 64] |= 1ull << (col_idx[e] % 64);
        if (syndrome[i]) a[(size_t)i * words + n / 64] |= 1ull << (n % 64);
    }
    /* stable ascending order by counting (O(n^2), fine for a checker) */
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int j = 0; j < n; j++) r += osd_less(llr[j], j, llr[i], i);
        order[r] = i;
    }
    int rank = 0;
    for (int t = 0; t < n && rank < m; t++) {
        const int c = order[t];
        int p = -1;
        for (int i = 0; i < m; i++)
            if (pivot_col[i] < 0 && ((a[(size_t)i * words + c / 64] >> (c % 64)) & 1)) { p = i; break; }
        if (p < 0) continue;
        for (int i = 0; i < m; i++)
            if (i != p && ((a[(size_t)i * words + c / 64] >> (c % 64)) & 1))
                for (int w = 0; w < words; w++) a[(size_t)i * words + w] ^= a[(size_t)p * words + w];
        pivot_col[p] = c;
        rank++;
        int in_image = 1;
        for (int i = 0; i < m; i++)
            if (pivot_col[i] < 0 && ((a[(size_t)i * words + n / 64] >> (n % 64)) & 1)) { in_image = 0; break; }
        if (in_image) break;
    }
    memset(decoding, 0, (size_t)n);
    for (int i = 0; i < m; i++)
        if (pivot_col[i] >= 0) decoding[pivot_col[i]] = (uint8_t)((a[(size_t)i * words + n / 64] >> (n % 64)) & 1);
    free(a); free(order); free(pivot_col);
}

/* ============================================================================================== *
 * Higher-order OSD: ldpc::osd::OsdDecoder::decode with osd_order > 0 (src_cpp/osd.hpp:119-187), for
 * osd_method EXHAUSTIVE (2, "OSD_E") and COMBINATION_SWEEP (3, "OSD_CS").  Restated literally:
 *   1. column order by log-ratio as for OSD-0 (sort.hpp:48-62);
 *   2. rref over that order (gf2sparse_linalg.hpp:132-226): greedy pivot columns; `cols` afterwards lists
 *      the pivot columns, then the NON-pivot columns, each group in sorted order (:210-224);
 *   3. OSD-0 solution = lu_solve(syndrome), weight = sum_i x_i log(1 / p_i) in ascending i (osd.hpp:131-136);
 *   4. candidate strings over the k = n - rank non-pivot columns (osd.hpp:75-101): OSD_E the numbers
 *      1 .. 2^order - 1, bit j -> j-th non-pivot column (util.hpp:12-38, bits >= k dropped); OSD_CS the k
 *      weight-one strings, then the pairs (i, j), i < j < order, i-major;
 *   5. per candidate: flip the syndrome by the chosen non-pivot columns, solve on the pivot columns, set the
 *      chosen bits, weigh; keep it if STRICTLY lighter (osd.hpp:156-183).
 * The elimination is dense Gauss-Jordan on [H | I_m]: the right block E records the row operations, so the
 * solve of step 5 is y = E t and x[pivot_col[r]] = y[r] (equal to the reference's forward/back substitution
 * whenever t lies in the image of H; like OSD-0 this restatement requires that).
 * OSD_CS with osd_order > k writes past the candidate string in the reference (osd.hpp:92-96, undefined
 * behaviour); here such pairs are skipped.
 * ============================================================================================== */
static void osdw_oracle_in_image(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *llr,
                                 const uint8_t *syndrome, const double *channel_probs, int osd_method, int osd_order,
                                 uint8_t *decoding, uint8_t *osd0_decoding);
void osdw_oracle(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *llr,
                 const uint8_t *syndrome, const double *channel_probs, int osd_method, int osd_order,
                 uint8_t *decoding, uint8_t *osd0_decoding) {
    uint8_t *eff = osd_effective_syndrome(m, n, row_ptr, col_idx, llr, syndrome);
    osdw_oracle_in_image(m, n, row_ptr, col_idx, llr, eff, channel_probs, osd_method, osd_order, decoding, osd0_decoding);
    free(eff);
}

static void osdw_oracle_in_image(int m, int n, const int32_t *row_ptr, const int32_t *col_idx, const double *llr,
                                 const uint8_t *syndrome, const double *channel_probs, int osd_method, int osd_order,
                                 uint8_t *decoding, uint8_t *osd0_decoding) {
    if (osd_order <= 0 || osd_method < 2) {
        osd0_oracle_in_image(m, n, row_ptr, col_idx, llr, syndrome, decoding);
        if (osd0_decoding) memcpy(osd0_decoding, decoding, (size_t)n);
        return;
    }
    const int hw = (n + 63) / 64, ew = (m + 63) / 64, words = hw + ew;
    uint64_t *a = (uint64_t *)calloc((size_t)(m ? m : 1) * words, sizeof(uint64_t));  /* [H | I] */
    int *order = (int *)malloc(sizeof(int) * (size_t)(n ? n : 1));
    int *pivot_col = (int *)malloc(sizeof(int) * (size_t)(m ? m : 1));
    uint8_t *is_pivot = (uint8_t *)calloc((size_t)(n ? n : 1), 1);
    for (int i = 0; i < m; i++) {
        pivot_col[i] = -1;
        for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++) a[(size_t)i * words + col_idx[e] / 64] |= 1ull << (col_idx[e] % 64);
        a[(size_t)i * words + hw + i / 64] |= 1ull << (i % 64);
    }
    for (int i = 0; i < n; i++) {
        int r = 0;
        for (int j = 0; j < n; j++) r += osd_less(llr[j], j, llr[i], i);
        order[r] = i;
    }
    int rank = 0;
    const int max_rank = m < n ? m : n;
    for (int t = 0; t < n && rank < max_rank; t++) {
        const int c = order[t];
        int p = -1;
        for (int i = 0; i < m; i++)
            if (pivot_col[i] < 0 && ((a[(size_t)i * words + c / 64] >> (c % 64)) & 1)) { p = i; break; }
        if (p < 0) continue;
        for (int i = 0; i < m; i++)
            if (i != p && ((a[(size_t)i * words + c / 64] >> (c % 64)) & 1))
                for (int w = 0; w < words; w++) a[(size_t)i * words + w] ^= a[(size_t)p * words + w];
        pivot_col[p] = c;
        is_pivot[c] = 1;
        rank++;
    }
    const int k = n - rank;
    int *non_pivot = (int *)malloc(sizeof(int) * (size_t)(k ? k : 1));
    for (int t = 0, q = 0; t < n; t++)
        if (!is_pivot[order[t]]) non_pivot[q++] = order[t];

    uint8_t *t_syn = (uint8_t *)malloc((size_t)(m ? m : 1));
    uint8_t *cand = (uint8_t *)malloc((size_t)(n ? n : 1));
    uint8_t *string = (uint8_t *)calloc((size_t)(k ? k : 1), 1);
#define OSDW_SOLVE(out)                                                                          \
    do {                                                                                         \
        memset((out), 0, (size_t)n);                                                             \
        for (int r_ = 0; r_ < m; r_++) {                                                         \
            if (pivot_col[r_] < 0) continue;                                                     \
            int y_ = 0;                                                                          \
            for (int i_ = 0; i_ < m; i_++)                                                       \
                if (t_syn[i_] && ((a[(size_t)r_ * words + hw + i_ / 64] >> (i_ % 64)) & 1)) y_ ^= 1; \
            (out)[pivot_col[r_]] = (uint8_t)y_;                                                  \
        }                                                                                        \
    } while (0)
    for (int i = 0; i < m; i++) t_syn[i] = syndrome[i] ? 1 : 0;
    OSDW_SOLVE(decoding);
    if (osd0_decoding) memcpy(osd0_decoding, decoding, (size_t)n);
    double min_weight = 0;
    for (int i = 0; i < n; i++)
        if (decoding[i] == 1) min_weight += log(1 / channel_probs[i]);

    const long n_exh = osd_method == 2 ? (1L << osd_order) - 1 : 0;
    const long n_single = osd_method == 3 ? k : 0;
    const long n_pair = osd_method == 3 ? (long)osd_order * (osd_order - 1) / 2 : 0;
    long pi = 0, pj = 0;  /* next pair */
    for (long c = 0; c < n_exh + n_single + n_pair; c++) {
        memset(string, 0, (size_t)k);
        if (osd_method == 2) {
            for (int j = 0; j < k && j < 31; j++) string[j] = (uint8_t)(((c + 1) >> j) & 1);
        } else if (c < n_single) {
            string[c] = 1;
        } else {
            if (c == n_single) { pi = 0; pj = 1; }
            while (pj >= osd_order) { pi++; pj = pi + 1; }
            const long i0 = pi, j0 = pj;
            pj++;
            if (j0 >= k) continue;  /* out of range in the reference (see header) */
            string[i0] = 1;
            string[j0] = 1;
        }
        for (int i = 0; i < m; i++) t_syn[i] = syndrome[i] ? 1 : 0;
        for (int q = 0; q < k; q++)
            if (string[q])
                for (int i = 0; i < m; i++)  /* column non_pivot[q] of H */
                    for (int e = row_ptr[i]; e < row_ptr[i + 1]; e++)
                        if (col_idx[e] == non_pivot[q]) t_syn[i] ^= 1;
        OSDW_SOLVE(cand);
        for (int q = 0; q < k; q++) cand[non_pivot[q]] = string[q];
        double w = 0;
        for (int i = 0; i < n; i++)
            if (cand[i] == 1) w += log(1 / channel_probs[i]);
        if (w < min_weight) {
            min_weight = w;
            memcpy(decoding, cand, (size_t)n);
        }
    }
#undef OSDW_SOLVE
    free(a); free(order); free(pivot_col); free(is_pivot); free(non_pivot); free(t_syn); free(cand); free(string);
}

/* BpOsdDecoder.decode over a batch with any osd_method / osd_order (_bposd_decoder.pyx:125-134) */
void bposdw_oracle_decode_batch(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                                double ms_scaling_factor, int osd_method, int osd_order, const uint8_t *syndromes,
                                int64_t shots, uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge) {
    double *tmp = llr ? NULL : (double *)malloc(sizeof(double) * (size_t)(o->n ? o->n : 1));
    for (int64_t b = 0; b < shots; b++) {
        double *l = llr ? llr + b * o->n : tmp;
        iterations[b] = 0;
        bp_oracle_decode(o, channel_probs, max_iter, bp_method, ms_scaling_factor, syndromes + b * o->m,
                         decodings + b * o->n, l, iterations + b, converge + b);
        if (!converge[b])
            osdw_oracle(o->m, o->n, o->row_ptr, o->col_idx, l, syndromes + b * o->m, channel_probs, osd_method, osd_order,
                        decodings + b * o->n, NULL);
    }
    free(tmp);
}

/* BpOsdDecoder.decode over a batch (_bposd_decoder.pyx:125-134): BP; rows that did not converge get OSD-0 */
void bposd0_oracle_decode_batch(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                                double ms_scaling_factor, const uint8_t *syndromes, int64_t shots,
                                uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge) {
    double *tmp = llr ? NULL : (double *)malloc(sizeof(double) * (size_t)(o->n ? o->n : 1));
    for (int64_t b = 0; b < shots; b++) {
        double *l = llr ? llr + b * o->n : tmp;
        iterations[b] = 0;
        bp_oracle_decode(o, channel_probs, max_iter, bp_method, ms_scaling_factor, syndromes + b * o->m,
                         decodings + b * o->n, l, iterations + b, converge + b);
        if (!converge[b]) osd0_oracle(o->m, o->n, o->row_ptr, o->col_idx, l, syndromes + b * o->m, decodings + b * o->n);
    }
    free(tmp);
}

/* ============================================================================================== *
 * Serial schedule: ldpc::bp::BpDecoder::bp_decode_serial (src_cpp/bp.hpp:451-545) with a FIXED bit
 * order (`serial_schedule_order`, default 0..n-1, bp.hpp:120-124).  The random (bp.hpp:468) and
 * LLR-sorted "serial_relative" (bp.hpp:470-483) orders are not restated.
 * Differences from the flooding schedule that matter for bits: the check->bit message of an edge is
 * the plain sequential product over the row's OTHER entries (bp.hpp:493-498), its sign is
 * pow(-1, syndrome byte) (bp.hpp:499), min-sum multiplies alpha * sign * magnitude (bp.hpp:519).
 * ============================================================================================== */
/* ---- std::sort as libstdc++ implements it (bits/stl_algo.h: introsort = median-of-three quicksort down to runs of 16,
 * heapsort when the recursion budget 2 floor(log2 n) runs out, one final insertion sort), restated on an int array with
 * a comparator over keys.  serial_relative sorts its bit order with it every iteration (bp.hpp:470-483), the sort is not
 * stable, and with equal keys (uniform priors in iteration 1: ALL keys equal) the outcome is whatever this exact sequence
 * of swaps leaves -- so the sequence is part of the reference's observable behaviour.  libstdc++ is a third-party
 * dependency of the reference (GCC 11, the toolchain of this image); tests/test_std_sort_port.py checks this restatement
 * against the real std::sort on the host. ------------------------------------------------------------------------- */
typedef struct { const double *key; } sort_ctx;
static inline int sort_gt(const sort_ctx *c, int a, int b) { return c->key[a] > c->key[b]; }  /* comp(a, b): key[a] > key[b] */
#define SS_SWAP(i, j) do { t = v[i]; v[i] = v[j]; v[j] = t; } while (0)

static void ss_move_median_to_first(int *v, long result, long a, long b, long c, const sort_ctx *x) {
    int t;
    if (sort_gt(x, v[a], v[b])) {
        if (sort_gt(x, v[b], v[c])) SS_SWAP(result, b);
        else if (sort_gt(x, v[a], v[c])) SS_SWAP(result, c);
        else SS_SWAP(result, a);
    } else if (sort_gt(x, v[a], v[c])) SS_SWAP(result, a);
    else if (sort_gt(x, v[b], v[c])) SS_SWAP(result, c);
    else SS_SWAP(result, b);
}
static long ss_unguarded_partition(int *v, long first, long last, long pivot, const sort_ctx *x) {
    int t;
    for (;;) {
        while (sort_gt(x, v[first], v[pivot])) ++first;
        --last;
        while (sort_gt(x, v[pivot], v[last])) --last;
        if (!(first < last)) return first;
        SS_SWAP(first, last);
        ++first;
    }
}
static void ss_push_heap(int *v, long first, long hole, long top, int value, const sort_ctx *x) {
    long parent = (hole - 1) / 2;
    while (hole > top && sort_gt(x, v[first + parent], value)) {
        v[first + hole] = v[first + parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    v[first + hole] = value;
}
static void ss_adjust_heap(int *v, long first, long hole, long len, int value, const sort_ctx *x) {
    const long top = hole;
    long child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (sort_gt(x, v[first + child], v[first + (child - 1)])) child--;
        v[first + hole] = v[first + child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        v[first + hole] = v[first + (child - 1)];
        hole = child - 1;
    }
    ss_push_heap(v, first, hole, top, value, x);
}
static void ss_heapsort(int *v, long first, long last, const sort_ctx *x) { /* __partial_sort(first, last, last) */
    const long len = last - first;
    if (len >= 2)
        for (long parent = (len - 2) / 2;; parent--) {
            ss_adjust_heap(v, first, parent, len, v[first + parent], x);
            if (parent == 0) break;
        }
    while (last - first > 1) {
        --last;
        const int value = v[last];
        v[last] = v[first];
        ss_adjust_heap(v, first, 0, last - first, value, x);
    }
}
static void ss_unguarded_linear_insert(int *v, long last, const sort_ctx *x) {
    const int val = v[last];
    long next = last - 1;
    while (sort_gt(x, val, v[next])) {
        v[last] = v[next];
        last = next;
        --next;
    }
    v[last] = val;
}
static void ss_insertion_sort(int *v, long first, long last, const sort_ctx *x) {
    if (first == last) return;
    for (long i = first + 1; i != last; ++i) {
        if (sort_gt(x, v[i], v[first])) {
            const int val = v[i];
            memmove(v + first + 1, v + first, sizeof(int) * (size_t)(i - first));
            v[first] = val;
        } else ss_unguarded_linear_insert(v, i, x);
    }
}
void oracle_std_sort_desc(int *v, long n, const double *key) { /* std::sort(v, v + n, [](a, b) { return key[a] > key[b]; }) */
    if (n <= 0) return;
    const sort_ctx ctx = {key}, *x = &ctx;
    long depth = 0;
    for (long q = n; q > 1; q >>= 1) depth++;
    depth *= 2; /* std::__lg(n) * 2 */
    /* __introsort_loop recurses into the right part and iterates on the left; the parts are disjoint ranges, so deferring
     * the right part on an explicit stack of (first, last, depth) changes nothing observable */
    long stack_first[128], stack_last[128], stack_depth[128];
    int sp = 1;
    stack_first[0] = 0; stack_last[0] = n; stack_depth[0] = depth;
    while (sp > 0) {
        --sp;
        long first = stack_first[sp], last = stack_last[sp], d = stack_depth[sp];
        while (last - first > 16) {
            if (d == 0) { ss_heapsort(v, first, last, x); break; }
            --d;
            const long mid = first + (last - first) / 2;
            ss_move_median_to_first(v, first, first + 1, mid, last - 1, x);
            const long cut = ss_unguarded_partition(v, first + 1, last, first, x);
            stack_first[sp] = cut; stack_last[sp] = last; stack_depth[sp] = d; ++sp;
            last = cut;
        }
    }
    if (n > 16) { /* __final_insertion_sort */
        ss_insertion_sort(v, 0, 16, x);
        for (long i = 16; i != n; ++i) ss_unguarded_linear_insert(v, i, x);
    } else ss_insertion_sort(v, 0, n, x);
}
#undef SS_SWAP

/* serial schedules whose order changes while decoding:
 *   order_mode 0  the fixed `order` (NULL: 0 .. n-1), as bp_oracle_decode_serial always did;
 *   order_mode 1  iteration it walks orders[min(it, n_orders) - 1][0 .. n): the random serial schedule (bp.hpp:467-468; the
 *                 shuffles themselves are std::shuffle on std::mt19937, produced by oracle/shuffle_helper.cpp);
 *   order_mode 2  serial_relative (bp.hpp:469-483): `order_state` [n] is re-sorted in place at the start of every iteration,
 *                 by descending prior in iteration 1 and by descending log_prob_ratios of the previous iteration afterwards,
 *                 and keeps its final arrangement (the reference keeps it in the decoder object from one decode to the next).
 */
static void decode_serial_dyn(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method, double ms_scaling_factor,
                              int order_mode, const int32_t *order, const int32_t *orders, int n_orders, int32_t *order_state,
                              const uint8_t *syndrome, uint8_t *decoding, double *log_prob_ratios, int32_t *iterations, uint8_t *converge);

void bp_oracle_decode_serial(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                             double ms_scaling_factor, const int32_t *order, const uint8_t *syndrome,
                             uint8_t *decoding, double *log_prob_ratios, int32_t *iterations, uint8_t *converge) {
    decode_serial_dyn(o, channel_probs, max_iter, bp_method, ms_scaling_factor, 0, order, NULL, 0, NULL, syndrome, decoding,
                      log_prob_ratios, iterations, converge);
}

/* `fresh` != 0: every row starts from the same order_state (a new decoder object per row); else rows are decoded one after
 * the other on ONE decoder object, as a loop of BpDecoder.decode calls does.  order_state [n] ends as the last row left it. */
void bp_oracle_decode_serial_relative_batch(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                                            double ms_scaling_factor, int32_t *order_state, int fresh, const uint8_t *syndromes,
                                            int64_t shots, uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge) {
    int32_t *start = (int32_t *)malloc(sizeof(int32_t) * (size_t)(o->n ? o->n : 1));
    memcpy(start, order_state, sizeof(int32_t) * (size_t)o->n);
    for (int64_t b = 0; b < shots; b++) {
        if (fresh) memcpy(order_state, start, sizeof(int32_t) * (size_t)o->n);
        iterations[b] = 0;
        decode_serial_dyn(o, channel_probs, max_iter, bp_method, ms_scaling_factor, 2, NULL, NULL, 0, order_state,
                          syndromes + b * o->m, decodings + b * o->n, llr + b * o->n, iterations + b, converge + b);
    }
    free(start);
}

/* every row walks the same per-iteration orders (a new decoder object per row, freshly seeded) */
void bp_oracle_decode_serial_orders_batch(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                                          double ms_scaling_factor, const int32_t *orders, int n_orders, const uint8_t *syndromes,
                                          int64_t shots, uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge) {
    for (int64_t b = 0; b < shots; b++) {
        iterations[b] = 0;
        decode_serial_dyn(o, channel_probs, max_iter, bp_method, ms_scaling_factor, 1, NULL, orders, n_orders, NULL,
                          syndromes + b * o->m, decodings + b * o->n, llr + b * o->n, iterations + b, converge + b);
    }
}

static void decode_serial_dyn(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method, double ms_scaling_factor,
                              int order_mode, const int32_t *order, const int32_t *orders, int n_orders, int32_t *order_state,
                              const uint8_t *syndrome, uint8_t *decoding, double *log_prob_ratios, int32_t *iterations, uint8_t *converge) {
    const int m = o->m, n = o->n;
    *converge = 0;
    for (int j = 0; j < n; j++) { /* initialise_log_domain_bp, bp.hpp:147-157 */
        o->llr0[j] = log((1 - channel_probs[j]) / channel_probs[j]);
        for (int p = o->col_ptr[j]; p < o->col_ptr[j + 1]; p++) o->b2c[o->csc_edge[p]] = o->llr0[j];
    }
    for (int it = 1; it <= max_iter; it++) {
        double alpha;
        if (ms_scaling_factor == 0.0) alpha = 1.0 - pow(2.0, -1.0 * it);
        else alpha = ms_scaling_factor;
        if (order_mode == 1) order = orders + (size_t)((it < n_orders ? it : n_orders) - 1) * (size_t)n;
        if (order_mode == 2) { /* bp.hpp:469-483 */
            oracle_std_sort_desc(order_state, n, it != 1 ? log_prob_ratios : o->llr0);
            order = order_state;
        }
        for (int t = 0; t < n; t++) {
            const int bit = order ? order[t] : t;
            log_prob_ratios[bit] = log((1 - channel_probs[bit]) / channel_probs[bit]);
            for (int p = o->col_ptr[bit]; p < o->col_ptr[bit + 1]; p++) {
                const int e = o->csc_edge[p], chk = o->csc_row[p];
                if (bp_method == BP_PRODUCT_SUM) { /* bp.hpp:491-503 */
                    o->c2b[e] = 1.0;
                    for (int g = o->row_ptr[chk]; g < o->row_ptr[chk + 1]; g++)
                        if (g != e) o->c2b[e] *= o->tanh_half ? o->tanh_half(o->b2c[g]) : tanh(o->b2c[g] / 2);
                    o->c2b[e] = pow(-1, syndrome[chk]) * (o->log_ratio ? o->log_ratio(o->c2b[e]) : log((1 + o->c2b[e]) / (1 - o->c2b[e])));
                } else { /* bp.hpp:504-523 */
                    int sgn = syndrome[chk];
                    double temp = DBL_MAX;
                    for (int g = o->row_ptr[chk]; g < o->row_ptr[chk + 1]; g++)
                        if (g != e) {
                            const double a = fabs(o->b2c[g]);
                            if (a < temp) temp = a;
                            if (o->b2c[g] <= 0) sgn += 1;
                        }
                    const double message_sign = (sgn % 2 == 0) ? 1.0 : -1.0;
                    o->c2b[e] = alpha * message_sign * temp;
                }
                o->b2c[e] = log_prob_ratios[bit];
                log_prob_ratios[bit] += o->c2b[e];
            }
            decoding[bit] = log_prob_ratios[bit] <= 0 ? 1 : 0; /* bp.hpp:525-529 */
            double temp = 0;
            for (int p = o->col_ptr[bit + 1] - 1; p >= o->col_ptr[bit]; p--) { /* bp.hpp:530-534 */
                const int e = o->csc_edge[p];
                o->b2c[e] += temp;
                temp += o->c2b[e];
            }
        }
        /* candidate syndrome of the current hard decision, bp.hpp:537-543 */
        int equal = 1;
        for (int i = 0; i < m && equal; i++) {
            uint8_t s = 0;
            for (int g = o->row_ptr[i]; g < o->row_ptr[i + 1]; g++) s ^= decoding[o->col_idx[g]];
            if (s != syndrome[i]) equal = 0;
        }
        *iterations = it;
        if (equal) { *converge = 1; return; }
    }
}

/* ============================================================================================== *
 * Soft-syndrome serial min-sum: BpDecoder::soft_info_decode_serial (bp.hpp:547-660), what
 * SoftInfoBpDecoder.decode runs (_bp_decoder.pyx:761-785).  The analog syndrome s_i is scaled to
 * 2 s_i / sigma^2 (:553), its sign gives the hard syndrome (:554-558); during the serial sweep a check
 * whose scaled magnitude is below `cutoff` and below the smallest incoming magnitude acts as a virtual
 * variable node: it caps the outgoing magnitude and is itself updated or flipped (:597-621).
 * ms_scaling_factor is used as it is (no adaptive 1 - 2^-it here); the convergence test compares H x with
 * the CURRENT (possibly flipped) hard syndrome (:645-653).  `order` may be NULL (0 .. n-1).
 * ============================================================================================== */
static void soft_info_decode_orders(bp_oracle *o, const double *channel_probs, int max_iter, double ms_scaling_factor,
                                    const int32_t *order, const int32_t *orders, int n_orders, const double *soft_info_syndrome,
                                    double cutoff, double sigma, uint8_t *decoding, double *log_prob_ratios, int32_t *iterations,
                                    uint8_t *converge, double *soft_syndrome);

void bp_oracle_soft_info_decode(bp_oracle *o, const double *channel_probs, int max_iter, double ms_scaling_factor,
                                const int32_t *order, const double *soft_info_syndrome, double cutoff, double sigma,
                                uint8_t *decoding, double *log_prob_ratios, int32_t *iterations, uint8_t *converge,
                                double *soft_syndrome) {
    soft_info_decode_orders(o, channel_probs, max_iter, ms_scaling_factor, order, NULL, 0, soft_info_syndrome, cutoff, sigma,
                            decoding, log_prob_ratios, iterations, converge, soft_syndrome);
}

/* `orders` [n_orders][n], if given: the arrangement walked in iteration it is orders[min(it, n_orders) - 1] -- the random serial
 * schedule of this routine (bp.hpp:573-577: the order is reshuffled at the top of every iteration that still runs) */
static void soft_info_decode_orders(bp_oracle *o, const double *channel_probs, int max_iter, double ms_scaling_factor,
                                    const int32_t *order, const int32_t *orders, int n_orders, const double *soft_info_syndrome,
                                    double cutoff, double sigma, uint8_t *decoding, double *log_prob_ratios, int32_t *iterations,
                                    uint8_t *converge, double *soft_syndrome) {
    const int m = o->m, n = o->n;
    uint8_t *syndrome = (uint8_t *)malloc((size_t)(m ? m : 1));
    for (int i = 0; i < m; i++) {
        soft_syndrome[i] = 2 * soft_info_syndrome[i] / (sigma * sigma);
        syndrome[i] = soft_syndrome[i] <= 0 ? 1 : 0;
    }
    *converge = 0;
    for (int j = 0; j < n; j++) { /* initialise_log_domain_bp, bp.hpp:147-157 */
        o->llr0[j] = log((1 - channel_probs[j]) / channel_probs[j]);
        for (int p = o->col_ptr[j]; p < o->col_ptr[j + 1]; p++) o->b2c[o->csc_edge[p]] = o->llr0[j];
    }
    int converged = 0;
    for (int it = 1; it <= max_iter; it++) {
        if (converged) continue;
        if (orders) order = orders + (size_t)((it < n_orders ? it : n_orders) - 1) * (size_t)n;
        for (int t = 0; t < n; t++) {
            const int bit = order ? order[t] : t;
            log_prob_ratios[bit] = log((1 - channel_probs[bit]) / channel_probs[bit]);
            for (int p = o->col_ptr[bit]; p < o->col_ptr[bit + 1]; p++) {
                const int e = o->csc_edge[p], chk = o->csc_row[p];
                int sgn = 0;
                double temp = DBL_MAX;
                for (int g = o->row_ptr[chk]; g < o->row_ptr[chk + 1]; g++)
                    if (g != e) {
                        if (fabs(o->b2c[g]) < temp) temp = fabs(o->b2c[g]);
                        if (o->b2c[g] <= 0) sgn ^= 1;
                    }
                const double min_b2c = temp;
                double propagated = min_b2c;
                const double magnitude = fabs(soft_syndrome[chk]);
                if (magnitude < cutoff) {
                    if (magnitude < fabs(min_b2c)) {
                        propagated = magnitude;
                        int check_node_sgn = sgn;
                        if (o->b2c[e] <= 0) check_node_sgn ^= 1;
                        if (check_node_sgn == syndrome[chk]) {
                            if (fabs(o->b2c[e]) < min_b2c) soft_syndrome[chk] = pow(-1, syndrome[chk]) * fabs(o->b2c[e]);
                            else soft_syndrome[chk] = pow(-1, syndrome[chk]) * min_b2c;
                        } else {
                            syndrome[chk] ^= 1;
                            soft_syndrome[chk] *= -1;
                        }
                    }
                }
                sgn ^= syndrome[chk];
                o->c2b[e] = ms_scaling_factor * pow(-1, sgn) * propagated;
                o->b2c[e] = log_prob_ratios[bit];
                log_prob_ratios[bit] += o->c2b[e];
            }
            decoding[bit] = log_prob_ratios[bit] <= 0 ? 1 : 0;
            double temp = 0;
            for (int p = o->col_ptr[bit + 1] - 1; p >= o->col_ptr[bit]; p--) {
                const int e = o->csc_edge[p];
                o->b2c[e] += temp;
                temp += o->c2b[e];
            }
        }
        converged = 1;
        for (int i = 0; i < m && converged; i++) {
            uint8_t c = 0;
            for (int g = o->row_ptr[i]; g < o->row_ptr[i + 1]; g++) c ^= decoding[o->col_idx[g]];
            if (c != syndrome[i]) converged = 0;
        }
        *iterations = it;
    }
    *converge = (uint8_t)converged;
    free(syndrome);
}

void bp_oracle_soft_info_decode_batch(bp_oracle *o, const double *channel_probs, int max_iter, double ms_scaling_factor,
                                      const int32_t *order, const double *soft_syndromes, int64_t shots, double cutoff,
                                      double sigma, uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge,
                                      double *soft_syndromes_out) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(o->n + o->m + 1));
    for (int64_t b = 0; b < shots; b++) {
        iterations[b] = 0;
        bp_oracle_soft_info_decode(o, channel_probs, max_iter, ms_scaling_factor, order, soft_syndromes + b * o->m, cutoff, sigma,
                                   decodings + b * o->n, llr ? llr + b * o->n : tmp, iterations + b, converge + b,
                                   soft_syndromes_out ? soft_syndromes_out + b * o->m : tmp + o->n);
    }
    free(tmp);
}

/* every row walks the same per-iteration orders (a new decoder object per row) */
void bp_oracle_soft_info_decode_orders_batch(bp_oracle *o, const double *channel_probs, int max_iter, double ms_scaling_factor,
                                             const int32_t *orders, int n_orders, const double *soft_syndromes, int64_t shots,
                                             double cutoff, double sigma, uint8_t *decodings, double *llr, int32_t *iterations,
                                             uint8_t *converge, double *soft_syndromes_out) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)(o->n + o->m + 1));
    for (int64_t b = 0; b < shots; b++) {
        iterations[b] = 0;
        soft_info_decode_orders(o, channel_probs, max_iter, ms_scaling_factor, NULL, orders, n_orders, soft_syndromes + b * o->m,
                                cutoff, sigma, decodings + b * o->n, llr ? llr + b * o->n : tmp, iterations + b, converge + b,
                                soft_syndromes_out ? soft_syndromes_out + b * o->m : tmp + o->n);
    }
    free(tmp);
}

void bp_oracle_decode_serial_batch(bp_oracle *o, const double *channel_probs, int max_iter, int bp_method,
                                   double ms_scaling_factor, const int32_t *order, const uint8_t *syndromes,
                                   int64_t shots, uint8_t *decodings, double *llr, int32_t *iterations, uint8_t *converge) {
    double *tmp = llr ? NULL : (double *)malloc(sizeof(double) * (size_t)(o->n ? o->n : 1));
    for (int64_t b = 0; b < shots; b++) {
        iterations[b] = 0;
        bp_oracle_decode_serial(o, channel_probs, max_iter, bp_method, ms_scaling_factor, order, syndromes + b * o->m,
                                decodings + b * o->n, llr ? llr + b * o->n : tmp, iterations + b, converge + b);
    }
    free(tmp);
}
