#!/usr/bin/env python3
"""Golden fixtures of the flooding schedule (bp.hpp:192-325) on IRREGULAR LDPC matrices -- rows of 3 .. 16 entries and columns of 2 .. 8
(the shape of tools/bench_configs.py irregular), and rows of 3 .. 8 -- through the REAL reference (oracle/_ref/libref_bp.so).  These are
the matrices that have no fixed-degree ring variant: the streamed decode takes the per-pass kernels from the first iteration (product-sum),
the register variants of the persistent kernel, or its variable-degree LDS ring.  Run in the build container only:

    make -C oracle ref && python tests/golden/make_golden_irregular.py

Data only: CSR arrays of the matrix (+ checksum), the syndromes, the reference's outputs.
"""
from __future__ import annotations

import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import have_ref  # noqa: E402
from ldpc_amd import codes  # noqa: E402
from make_golden import bsc_syndromes, run_case  # noqa: E402


def main():
    if not have_ref():
        sys.exit("oracle/_ref/libref_bp.so missing: make -C oracle ref")
    # rows 3 .. 16, columns 2 .. 8: around the threshold of this ensemble, so that rows stop at different iterations and some never do
    h = codes.irregular_ldpc_code(600, 300, seed=3)
    s = bsc_syndromes(h, 13, 0.03, 0, 200)
    s[::41, 5] = 2  # syndrome bytes > 1: never converge (bp.hpp:300), sign by non-zero byte in product-sum (:213)
    run_case("irregular_ldpc_n600_ps16_p030", h, s, error_rate=0.03, max_iter=16, bp_method="product_sum", full_llr=24,
             note="irregular_ldpc_code(600,300,seed=3): rows 3..16, columns 2..8; error seed 13; some syndrome bytes = 2")
    run_case("irregular_ldpc_n600_ms16_p030_adaptive", h, bsc_syndromes(h, 13, 0.03, 200, 160), error_rate=0.03, max_iter=16,
             bp_method="minimum_sum", ms_scaling_factor=0.0, full_llr=24, note="adaptive min-sum scaling 1 - 2^-it; shots 200..359")
    h2 = codes.irregular_ldpc_code(2400, 1200, seed=7)
    run_case("irregular_ldpc_n2400_ps30_p075", h2, bsc_syndromes(h2, 5, 0.075, 0, 130), error_rate=0.075, max_iter=30, bp_method="product_sum",
             full_llr=6, note="irregular_ldpc_code(2400,1200,seed=7); two tiles and two rows; rows stop anywhere between iteration 9 and 30, a fifth never")
    run_case("irregular_ldpc_n2400_ms30_p060_a0625", h2, bsc_syndromes(h2, 5, 0.06, 130, 96), error_rate=0.06, max_iter=30,
             bp_method="minimum_sum", ms_scaling_factor=0.625, full_llr=6)
    # rows 3 .. 8 (the 8-entry register variants / per-pass kernels), columns 2 .. 8
    h3 = codes.irregular_ldpc_code(1200, 600, seed=2, row_weights=(3, 4, 5, 6, 7, 8), col_weights=((2, 0.35), (3, 0.5), (8, 0.15)))
    run_case("irregular8_ldpc_n1200_ps20_p040", h3, bsc_syndromes(h3, 9, 0.04, 0, 140), error_rate=0.04, max_iter=20, bp_method="product_sum",
             full_llr=8, note="irregular_ldpc_code(1200,600,seed=2,row_weights=3..8,col_weights=2/3/8)")


if __name__ == "__main__":
    main()
