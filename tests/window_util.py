"""Phenomenological detector error models for the overlapping-window tests, written as stim DEM text (no stim needed).

A classical code H (m x n) is measured for ``rounds`` rounds; detector (t, i) = check i of round t XOR the same check of
round t - 1.  Round t has one error per bit j (flips the detectors (t, i) of the checks on j, and observable L0 if j is
in ``logical``) and, except in the last round, one measurement error per check (flips (t, i) and (t + 1, i)).
The text uses ``repeat`` and ``shift_detectors`` so the reader's flattening is exercised too.
"""
import numpy as np
import scipy.sparse as sp


def phenomenological_dem(h, rounds, p_data, p_meas, logical=(0,)):
    h = sp.csc_matrix(h)
    m, n = h.shape
    data_lines = []
    for j in range(n):
        checks = h.indices[h.indptr[j]:h.indptr[j + 1]]
        tgt = " ".join(f"D{i}" for i in checks) + (" L0" if j in logical else "")
        data_lines.append(f"    error({p_data[j] if np.ndim(p_data) else p_data}) {tgt}".rstrip())
    meas_lines = [f"    error({p_meas}) D{i} D{i + m}" for i in range(m)]
    body = "\n".join(data_lines + meas_lines)
    last = "\n".join(line[4:] for line in data_lines)
    return (f"# phenomenological noise, {rounds} rounds of a {m} x {n} code\n"
            f"repeat {rounds - 1} {{\n{body}\n    shift_detectors(0, 0, 1) {m}\n}}\n{last}\n"
            f"detector(0, 0, 0) D{m - 1}\nlogical_observable L0\n")


def phenomenological_matrices(h, rounds, p_data, p_meas, logical=(0,)):
    """The same model built directly: (check_matrix, observables_matrix, priors) in the DEM's column order."""
    h = sp.csc_matrix(h)
    m, n = h.shape
    cols, obs, pri = [], [], []
    for t in range(rounds):
        for j in range(n):
            cols.append([t * m + i for i in h.indices[h.indptr[j]:h.indptr[j + 1]]])
            obs.append(j in logical)
            pri.append(p_data[j] if np.ndim(p_data) else p_data)
        if t < rounds - 1:
            for i in range(m):
                cols.append([t * m + i, (t + 1) * m + i])
                obs.append(False)
                pri.append(p_meas)
    rows = [r for c in cols for r in c]
    cidx = [k for k, c in enumerate(cols) for _ in c]
    check = sp.csc_matrix((np.ones(len(rows), np.uint8), (rows, cidx)), shape=(rounds * m, len(cols)))
    observables = sp.csc_matrix(np.array(obs, np.uint8)[None, :])
    return check, observables, np.array(pri, np.float64)


def ring_code(n):
    h = np.zeros((n, n), np.uint8)
    for i in range(n):
        h[i, i] = h[i, (i + 1) % n] = 1
    return sp.csr_matrix(h)


def sample_shots(check, priors, shots, seed):
    rng = np.random.default_rng(seed)
    e = (rng.random((shots, check.shape[1])) < priors[None, :]).astype(np.uint8)
    return np.ascontiguousarray((sp.csr_matrix(check) @ e.T % 2).T.astype(np.uint8)), e
