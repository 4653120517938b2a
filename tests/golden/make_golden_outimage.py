#!/usr/bin/env python3
"""Golden fixtures for OSD on syndromes OUTSIDE the image of H, through the REAL reference (oracle/_ref/libref_bp.so).
Build container only:   make -C oracle ref && python tests/golden/make_golden_outimage.py

H = X checks of a toric code (L x L vertices, 2 L^2 qubits): rank L^2 - 1, the checks sum to zero.  Half of the rows are
syndromes of random errors (inside the image), the other half the same with ONE check flipped (a measurement error: the
parity of the checks is odd, no x solves H x = s).  What the reference returns for the latter is the solution of the
subsystem of its own pivot rows (gf2sparse_linalg.hpp:237-288 after :298-401 ran out of columns); the device flags such
rows (ldpc_hip_bposd_get_status == 2) instead of reproducing that vector -- include/ldpc_hip.h.  The fixture keeps the
reference's outputs for every row: the in-image rows are compared bit for bit, the others document the difference.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import RefBpOsd, csr_arrays  # noqa: E402
from make_golden import bsc_syndromes, h_crc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def toric_hx(L):
    """Vertex checks of an L x L torus: qubits on the 2 L^2 edges; vertex (r, c) touches its four edges."""
    rows, cols = [], []
    for r in range(L):
        for c in range(L):
            v = r * L + c
            edges = [r * L + c, r * L + (c - 1) % L, L * L + r * L + c, L * L + ((r - 1) % L) * L + c]  # right, left, down, up
            for e in edges:
                rows.append(v)
                cols.append(e)
    return sp.csr_matrix((np.ones(len(rows), np.uint8), (rows, cols)), shape=(L * L, 2 * L * L))


def run(name, L, *, osd_method, osd_order, max_iter, p, bp_method, alpha, rows=192):
    h = toric_hx(L)
    m, n, rp, ci = csr_arrays(h)
    s = bsc_syndromes(h, 23, p, 0, rows)
    rng = np.random.default_rng(5)
    inside = np.ones(rows, bool)
    for b in range(1, rows, 2):
        s[b, rng.integers(m)] ^= 1
        inside[b] = False
    assert np.all((s.sum(axis=1) % 2 == 0) == inside)  # the X checks of a torus sum to zero
    ref = RefBpOsd(h, osd_method=osd_method, osd_order=osd_order, error_rate=p, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=alpha)
    dec, llr, it, conv = ref.decode_batch(s)
    solved = ~(((h.astype(np.int64) @ dec.T.astype(np.int64)).T % 2) != s).any(axis=1)
    assert np.all(solved[inside]) and not np.any(solved[~inside & ~conv])
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, name=name, m=m, n=n, h_crc=np.uint32(h_crc(h)), row_ptr=rp, col_idx=ci, p=np.float64(p), max_iter=np.int32(max_iter),
                        bp_method=np.int32(0 if bp_method == "product_sum" else 1), ms_scaling_factor=np.float64(alpha),
                        osd_method=np.int32(osd_method), osd_order=np.int32(osd_order), syndromes=np.packbits(s, axis=1),
                        inside_image=inside, decoding=np.packbits(dec, axis=1), converge=conv, iterations=it, llr=llr)
    print(f"{name:32s} {m} x {n} rows={rows} BP converged={int(conv.sum())} OSD inside={int((~conv & inside).sum())} outside={int((~conv & ~inside).sum())} "
          f"{os.path.getsize(path) / 1024:.1f} KiB")


def main():
    run("outimage_toric6_osd0_ms", 6, osd_method=1, osd_order=0, max_iter=4, p=0.08, bp_method="minimum_sum", alpha=0.625)
    run("outimage_toric6_cs6_ps", 6, osd_method=3, osd_order=6, max_iter=3, p=0.08, bp_method="product_sum", alpha=1.0)
    run("outimage_toric8_e4_ms", 8, osd_method=2, osd_order=4, max_iter=3, p=0.06, bp_method="minimum_sum", alpha=0.8)
    run("outimage_toric16_osd0_ms", 16, osd_method=1, osd_order=0, max_iter=3, p=0.05, bp_method="minimum_sum", alpha=0.625, rows=96)


if __name__ == "__main__":
    main()
