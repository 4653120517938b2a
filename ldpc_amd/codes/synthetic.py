"""Generators for the codes BASELINE.json's GPU configs name.

None of these exist in the reference.  They are deterministic functions of their arguments and of
``ldpc_amd.prng`` only (never of NumPy's global RNG), so the bench, the tests and the golden-vector
generator all see the same matrices.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

from ldpc_amd.prng import sm64_int


def regular_ldpc_code(n: int = 10_000, dv: int = 3, dc: int = 6, seed: int = 1) -> sp.csr_matrix:
    """(dv, dc)-regular LDPC parity-check matrix from the configuration model.

    Check ``i`` owns sockets ``dc*i .. dc*i + dc - 1``; the bit-side socket list
    ``repeat(arange(n), dv)`` is Fisher-Yates shuffled with the SplitMix64 stream for ``seed``;
    a check that received the same bit twice has the offending socket swapped with a further
    pseudo-random socket until every check has ``dc`` distinct bits (so every row has weight
    exactly ``dc`` and every column weight exactly ``dv``).
    """
    if (n * dv) % dc:
        raise ValueError("n * dv must be divisible by dc")
    edges = n * dv
    m = edges // dc
    sock = [b for b in range(n) for _ in range(dv)]
    ctr = 0
    for i in range(edges - 1, 0, -1):
        j = sm64_int(seed, ctr) % (i + 1)
        ctr += 1
        sock[i], sock[j] = sock[j], sock[i]

    def row_ok(c: int) -> bool:
        r = sock[dc * c: dc * c + dc]
        return len(set(r)) == dc

    for c in range(m):
        guard = 0
        while not row_ok(c):
            row = sock[dc * c: dc * c + dc]
            pos = next(dc * c + k for k in range(dc) if row[k] in row[:k])
            q = sm64_int(seed, ctr) % edges
            ctr += 1
            guard += 1
            if guard > 100_000:
                raise RuntimeError("multi-edge repair did not terminate")
            c2 = q // dc
            if c2 == c:
                continue
            sock[pos], sock[q] = sock[q], sock[pos]
            if not row_ok(c2):
                sock[pos], sock[q] = sock[q], sock[pos]  # undo: never break another check
    rows = np.repeat(np.arange(m), dc)
    cols = np.asarray(sock, dtype=np.int64)
    h = sp.csr_matrix((np.ones(edges, dtype=np.uint8), (rows, cols)), shape=(m, n), dtype=np.uint8)
    h.sum_duplicates()
    h.sort_indices()
    assert h.nnz == edges and int(h.data.max()) == 1
    return h


def irregular_ldpc_code(n: int = 10_000, m: int = 5_000, seed: int = 1,
                        row_weights=(3, 4, 5, 6, 7, 8, 9, 10, 12, 16),
                        col_weights=((2, 0.20), (3, 0.50), (6, 0.15), (8, 0.15))) -> sp.csr_matrix:
    """An irregular LDPC parity-check matrix from the configuration model: check ``i`` has weight ``row_weights[i % len]`` (3 ... 16 by
    default, mean 8), the bits get the weights of ``col_weights`` (weight, share) in the shares given -- adjusted on the heaviest class
    so that both sides count the same edges -- and the bit-side sockets are shuffled with the SplitMix64 stream of ``seed``; multi-edges
    are repaired by socket swaps as in ``regular_ldpc_code``.  The workload of ``tools/bench_configs.py irregular``: every register
    bound of the streamed kernels (rows of up to 16, columns of up to 8) with one matrix.
    """
    rw = [int(row_weights[i % len(row_weights)]) for i in range(m)]
    edges = sum(rw)
    cw = []
    for w, share in col_weights:
        cw += [int(w)] * int(round(share * n))
    cw = (cw + [int(col_weights[0][0])] * n)[:n]
    diff = edges - sum(cw)  # settle the difference one edge at a time, heaviest bits first (down) / lightest first (up)
    order = sorted(range(n), key=lambda j: -cw[j]) if diff < 0 else sorted(range(n), key=lambda j: cw[j])
    k = 0
    while diff != 0:
        j = order[k % n]
        if diff < 0 and cw[j] > 1:
            cw[j] -= 1
            diff += 1
        elif diff > 0:
            cw[j] += 1
            diff -= 1
        k += 1
    # interleave the weight classes over the bit indices (a deterministic shuffle of which bit is heavy)
    perm = list(range(n))
    ctr = 0
    for i in range(n - 1, 0, -1):
        j = sm64_int(seed ^ 0x5151, ctr) % (i + 1)
        ctr += 1
        perm[i], perm[j] = perm[j], perm[i]
    cw = [cw[perm[j]] for j in range(n)]
    sock = [b for b in range(n) for _ in range(cw[b])]
    ctr = 0
    for i in range(edges - 1, 0, -1):
        j = sm64_int(seed, ctr) % (i + 1)
        ctr += 1
        sock[i], sock[j] = sock[j], sock[i]
    start = [0] * (m + 1)
    for i in range(m):
        start[i + 1] = start[i] + rw[i]
    owner = [0] * edges
    for i in range(m):
        for q in range(start[i], start[i + 1]):
            owner[q] = i

    def row_ok(c: int) -> bool:
        r = sock[start[c]: start[c + 1]]
        return len(set(r)) == len(r)

    for c in range(m):
        guard = 0
        while not row_ok(c):
            row = sock[start[c]: start[c + 1]]
            pos = next(start[c] + k for k in range(len(row)) if row[k] in row[:k])
            q = sm64_int(seed, ctr) % edges
            ctr += 1
            guard += 1
            if guard > 100_000:
                raise RuntimeError("multi-edge repair did not terminate")
            c2 = owner[q]
            if c2 == c:
                continue
            sock[pos], sock[q] = sock[q], sock[pos]
            if not row_ok(c2):
                sock[pos], sock[q] = sock[q], sock[pos]
    rows = np.repeat(np.arange(m), rw)
    h = sp.csr_matrix((np.ones(edges, dtype=np.uint8), (rows, np.asarray(sock, dtype=np.int64))), shape=(m, n), dtype=np.uint8)
    h.sum_duplicates()
    h.sort_indices()
    assert h.nnz == edges and int(h.data.max()) == 1
    return h


def rotated_surface_code_x(d: int = 21) -> sp.csr_matrix:
    """X-check matrix of the distance-``d`` rotated surface code (``(d*d-1)/2 x d*d``).

    Data qubit ``(r, c)`` has index ``r*d + c``.  Plaquette ``(i, j)``, ``i, j in [-1, d-1]``, covers
    the qubits ``(i, j), (i, j+1), (i+1, j), (i+1, j+1)`` that exist; X-type plaquettes are those with
    ``(i + j)`` even: all bulk ones (weight 4) plus weight-2 ones on the top (``i = -1``) and bottom
    (``i = d-1``) boundaries.  ``d = 21`` gives the 220 x 441, nnz = 840 matrix of BASELINE config 3.
    """
    if d < 3 or d % 2 == 0:
        raise ValueError("d must be odd and >= 3")
    rows, cols = [], []
    r_idx = 0
    for i in range(-1, d):
        for j in range(0, d - 1):
            if (i + j) % 2:
                continue
            qs = [(i + a, j + b) for a in (0, 1) for b in (0, 1)]
            qs = [(r, c) for (r, c) in qs if 0 <= r < d and 0 <= c < d]
            for (r, c) in qs:
                rows.append(r_idx)
                cols.append(r * d + c)
            r_idx += 1
    h = sp.csr_matrix((np.ones(len(rows), dtype=np.uint8), (rows, cols)), shape=(r_idx, d * d), dtype=np.uint8)
    h.sort_indices()
    assert r_idx == (d * d - 1) // 2
    return h


def _shift(k: int) -> np.ndarray:
    return np.roll(np.eye(k, dtype=np.int64), 1, axis=1)


def bivariate_bicycle_hx(
    l: int = 12,
    m: int = 6,
    a_terms=(("x", 3), ("y", 1), ("y", 2)),
    b_terms=(("y", 3), ("x", 1), ("x", 2)),
) -> sp.csr_matrix:
    """``hx = [A | B]`` of a bivariate-bicycle code; defaults give [[144,12,12]] (72 x 144, nnz 432).

    ``x = S_l (x) I_m``, ``y = I_l (x) S_m`` with ``S_k`` the cyclic shift; ``A = x^3 + y + y^2``,
    ``B = y^3 + x + x^2`` over GF(2).
    """
    x = np.kron(_shift(l), np.eye(m, dtype=np.int64))
    y = np.kron(np.eye(l, dtype=np.int64), _shift(m))
    var = {"x": x, "y": y}

    def poly(terms):
        acc = np.zeros((l * m, l * m), dtype=np.int64)
        for v, e in terms:
            acc = (acc + np.linalg.matrix_power(var[v], e)) % 2
        return acc

    h = np.concatenate([poly(a_terms), poly(b_terms)], axis=1).astype(np.uint8)
    out = sp.csr_matrix(h, dtype=np.uint8)
    out.sort_indices()
    return out


def hypergraph_product_hx(h1, h2=None) -> sp.csr_matrix:
    """X-check matrix ``[H1 (x) I_n2 | I_m1 (x) H2^T]`` of the hypergraph product of two classical codes (H2 = H1 if omitted).

    ``hypergraph_product_hx(regular_ldpc_code(n=32, dv=3, dc=4, seed=5))`` is the 768 x 1600 matrix of a [[1600, 64]] code
    used by ``tools/bench_configs.py hgp1600`` and the ``*hgp1600*`` fixtures: the smallest of the usual BP+OSD benchmark
    sizes whose ``[H | s]`` no longer fits LDS.
    """
    h1 = sp.csr_matrix(h1, dtype=np.uint8)
    h2 = h1 if h2 is None else sp.csr_matrix(h2, dtype=np.uint8)
    (m1, _), (_, n2) = h1.shape, h2.shape
    hx = sp.hstack([sp.kron(h1, sp.identity(n2, dtype=np.uint8)), sp.kron(sp.identity(m1, dtype=np.uint8), h2.T)]).tocsr().astype(np.uint8)
    hx.sort_indices()
    return hx

