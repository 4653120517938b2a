#!/usr/bin/env python3
"""Prototype (numpy, CPU) of the blocked Gauss-Jordan elimination planned for osd_big_kernel ("Next").

Today's kernel makes one pivot per step: two barriers around a read-modify-write of every hit row in L2 / MALL, ~768 times
for a 768 x 1600 matrix.  With the columns stored in SORTED order, the 64 columns of a block are one 64-bit plane:
  1. the block's pivots are found on that plane alone (in LDS), recording for every row r the set c[r] of block pivots
     that were XORed into it, in the order they were made;
  2. A_i := "pivot row i as it was when it became a pivot" = orig[p_i] ^ XOR_{j in c[p_i], j < i} A_j   (64 short rows, LDS);
  3. every row takes ONE combined update on all the other planes:  row_r ^= XOR_{i in c[r]} A_i  (pivot rows included, for
     the pivots made after them) -- one load and one store per row per block instead of one per pivot.
This file checks that 1-3 reproduce the plain one-pivot-at-a-time elimination (same reduced matrix, same pivots) on
random matrices, rank-deficient ones included.  It is not used by the library.
"""
import numpy as np


def plain(mat, m, n):
    """Greedy column-ordered Gauss-Jordan, first unpivoted row with the bit becomes the pivot (osd_kernels.h)."""
    mat = mat.copy()
    pivcol = -np.ones(m, dtype=np.int64)
    for c in range(n):
        rows = np.flatnonzero(mat[:, c] & (pivcol < 0))
        if rows.size == 0:
            continue
        p = rows[0]
        hit = np.flatnonzero(mat[:, c])
        hit = hit[hit != p]
        mat[hit] ^= mat[p]
        pivcol[p] = c
    return mat, pivcol


def blocked(mat, m, n, block=64):
    mat = mat.copy()
    pivcol = -np.ones(m, dtype=np.int64)
    for c0 in range(0, n, block):
        c1 = min(n, c0 + block)
        plane = mat[:, c0:c1].copy()           # step 1 works on this copy only
        orig = mat.copy()                       # the other planes as they are before the block
        combo = [[] for _ in range(m)]          # c[r]: block pivots XORed into row r, in order
        pivots = []                             # (row, local column) in the order they are made
        for lc in range(c1 - c0):
            rows = np.flatnonzero(plane[:, lc] & (pivcol < 0))
            if rows.size == 0:
                continue
            p = rows[0]
            i = len(pivots)
            pivots.append((p, lc))
            pivcol[p] = c0 + lc
            for r in np.flatnonzero(plane[:, lc]):
                if r != p:
                    plane[r] ^= plane[p]
                    combo[r].append(i)
        # step 2: A_i, sequentially (pivot row i had the pivots in combo[p_i] that precede i applied before it was used)
        A = []
        for i, (p, _) in enumerate(pivots):
            a = orig[p].copy()
            for j in combo[p]:
                if j < i:
                    a ^= A[j]
            A.append(a)
        # step 3: one combined update per row
        for r in range(m):
            for i in combo[r]:
                mat[r] ^= A[i]
        assert np.array_equal(mat[:, c0:c1], plane), "the block's own plane comes out as step 1 left it"
    return mat, pivcol


def main():
    rng = np.random.default_rng(0)
    for trial in range(60):
        m = int(rng.integers(1, 90))
        n = int(rng.integers(1, 200))
        dens = rng.uniform(0.02, 0.5)
        mat = (rng.random((m, n)) < dens).astype(np.uint8)
        if trial % 5 == 0 and m > 2:
            mat[-1] = mat[0] ^ mat[1]          # rank deficiency
        want, wp = plain(mat, m, n)
        for block in (64, 8, 1):
            got, gp = blocked(mat, m, n, block)
            assert np.array_equal(got, want) and np.array_equal(gp, wp), (trial, m, n, block)
    print("blocked elimination == one pivot at a time on 60 random matrices (blocks of 64, 8, 1)")


if __name__ == "__main__":
    main()


def blocked_orig(mat, m, n, block=64):
    """The form osd_big_kernel uses: every row carries a mask M_r over the block's pivots meaning
    row_r (now) = row_r (at block start) ^ XOR_{j in M_r} pivot_row_j (at block start); taking pivot p's row means
    M_r ^= M_p ^ {p} -- no sequential step 2, every row's update reads block-start rows only."""
    mat = mat.copy()
    pivcol = -np.ones(m, dtype=np.int64)
    for c0 in range(0, n, block):
        c1 = min(n, c0 + block)
        plane = mat[:, c0:c1].copy()
        orig = mat.copy()
        M = np.zeros((m, block), dtype=np.uint8)
        piv_rows = []
        for lc in range(c1 - c0):
            rows = np.flatnonzero(plane[:, lc] & (pivcol < 0))
            if rows.size == 0:
                continue
            p = rows[0]
            j = len(piv_rows)
            piv_rows.append(p)
            pivcol[p] = c0 + lc
            unit = np.zeros(block, dtype=np.uint8)
            unit[j] = 1
            for r in np.flatnonzero(plane[:, lc]):
                if r != p:
                    plane[r] ^= plane[p]
                    M[r] ^= M[p] ^ unit
        for r in range(m):
            for j in np.flatnonzero(M[r]):
                mat[r] ^= orig[piv_rows[j]]
        assert np.array_equal(mat[:, c0:c1], plane)
    return mat, pivcol


def check_orig_form():
    rng = np.random.default_rng(1)
    for trial in range(60):
        m = int(rng.integers(1, 90))
        n = int(rng.integers(1, 200))
        mat = (rng.random((m, n + 1)) < rng.uniform(0.02, 0.5)).astype(np.uint8)  # last column: the syndrome rides along
        if trial % 5 == 0 and m > 2:
            mat[-1] = mat[0] ^ mat[1]
        want, wp = plain(mat[:, :n].copy(), m, n)
        # the syndrome column under the plain elimination
        aug, _ = plain_aug(mat, m, n)
        for block in (64, 8):
            got, gp = blocked_orig_aug(mat, m, n, block)
            assert np.array_equal(got[:, :n], want) and np.array_equal(gp, wp) and np.array_equal(got[:, n], aug[:, n]), (trial, block)
    print("mask-over-block-start-rows form == one pivot at a time, syndrome column included")


def plain_aug(mat, m, n):
    mat = mat.copy()
    pivcol = -np.ones(m, dtype=np.int64)
    for c in range(n):
        rows = np.flatnonzero(mat[:, c] & (pivcol < 0))
        if rows.size == 0:
            continue
        p = rows[0]
        hit = np.flatnonzero(mat[:, c])
        hit = hit[hit != p]
        mat[hit] ^= mat[p]
        pivcol[p] = c
    return mat, pivcol


def blocked_orig_aug(mat, m, n, block):
    """blocked_orig on [H | s]: the syndrome bit of row r takes parity(M_r & S) with S_j = syndrome bit of pivot row j at block start."""
    h, pc = blocked_orig(mat[:, :n].copy(), m, n, block)
    # redo with the syndrome riding along (same masks: recompute them)
    full = mat.copy()
    pivcol = -np.ones(m, dtype=np.int64)
    for c0 in range(0, n, block):
        c1 = min(n, c0 + block)
        plane = full[:, c0:c1].copy()
        orig = full.copy()
        M = np.zeros((m, block), dtype=np.uint8)
        piv_rows = []
        for lc in range(c1 - c0):
            rows = np.flatnonzero(plane[:, lc] & (pivcol < 0))
            if rows.size == 0:
                continue
            p = rows[0]
            j = len(piv_rows)
            piv_rows.append(p)
            pivcol[p] = c0 + lc
            unit = np.zeros(block, dtype=np.uint8)
            unit[j] = 1
            for r in np.flatnonzero(plane[:, lc]):
                if r != p:
                    plane[r] ^= plane[p]
                    M[r] ^= M[p] ^ unit
        S = np.array([orig[p, n] for p in piv_rows] + [0] * (block - len(piv_rows)), dtype=np.uint8)
        for r in range(m):
            for j in np.flatnonzero(M[r]):
                full[r, :n] ^= orig[piv_rows[j], :n]
            full[r, n] ^= int((M[r] & S).sum() & 1)
    assert np.array_equal(full[:, :n], h)
    return full, pivcol


if __name__ == "__main__":
    check_orig_form()
