#!/usr/bin/env python3
"""Golden fixtures for OSD on matrices that do not fit the one-wavefront kernels: a [[1600,64]] hypergraph product
(hx 768 x 1600, [H | s] beyond LDS: workgroup kernel with H in an HBM scratch slot) and a 400 x 900 matrix (workgroup
kernel with H in LDS), through the REAL reference (oracle/_ref/libref_bp.so).  Build container only:

    make -C oracle ref && python tests/golden/make_golden_osd_big.py

Same schema as make_golden_osdw.py (the fixtures are named osdw_* and picked up by the same tests): ``decoding`` = the
swept solution, ``osd0_decoding`` = the same decoder at order 0.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ldpc_amd import codes  # noqa: E402
from make_golden import bsc_syndromes  # noqa: E402
from make_golden_osdw import run  # noqa: E402


def hgp1600():
    return codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5))


def main():
    hx = hgp1600()
    s = bsc_syndromes(hx, 21, 0.05, 0, 48)
    run("osdw_cs10_hgp1600_ms12", hx, s, osd_method="osd_cs", osd_order=10, max_iter=12, error_rate=0.05, bp_method="minimum_sum",
        ms_scaling_factor=0.625, note="768 x 1600: [H | s] is 156 KiB bit-packed, beyond LDS")
    run("osdw_cs100_hgp1600_ms12", hx, s[:16], osd_method="osd_cs", osd_order=100, max_iter=12, error_rate=0.05, bp_method="minimum_sum",
        ms_scaling_factor=0.625, note="osd_order 100: pairs across T planes 0 and 1 of the workgroup kernel (round 4)")
    run("osdw_e6_hgp1600_ms12", hx, s[:24], osd_method="osd_e", osd_order=6, max_iter=12, error_rate=0.05, bp_method="minimum_sum",
        ms_scaling_factor=0.625)
    rng = np.random.default_rng(7)
    m, n = 400, 900
    h = sp.csr_matrix((np.ones(m * 6, np.uint8), (np.repeat(np.arange(m), 6), rng.integers(0, n, size=m * 6))), shape=(m, n))
    h.sum_duplicates()
    h.data[:] = 1
    chan = rng.uniform(0.01, 0.08, size=n)
    run("osdw_cs8_random400x900_ps6", h, bsc_syndromes(h, 5, 0.04, 0, 64), osd_method="osd_cs", osd_order=8, max_iter=6,
        error_channel=chan, note="three one-wavefront kernels per CU at most: the workgroup kernel with H in LDS takes it")




def bp_fixtures():
    """BP alone on the [[1600,64]] matrix, log-ratios included: two wavefronts per CU, so bp_wave_kernel runs with the
    log-ratios stored straight to HBM (WaveArgs.llr_direct) -- pinned here to the real reference, both methods."""
    from make_golden import run_case
    hx = hgp1600()
    s = bsc_syndromes(hx, 31, 0.03, 0, 64)
    run_case("hgp1600_ms20_p030", hx, s, error_rate=0.03, max_iter=20, bp_method="minimum_sum", ms_scaling_factor=0.625, full_llr=16)
    run_case("hgp1600_ps12_p030", hx, s[:32], error_rate=0.03, max_iter=12, bp_method="product_sum", full_llr=8)


if __name__ == "__main__":
    main()
    bp_fixtures()
