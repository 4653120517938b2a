#!/usr/bin/env python3
"""Golden fixtures for the overlapping-window decoders (ckt_noise/base_overlapping_window_decoder.py:96-214,
bposd_overlapping_window.py).  Build container only:

    make -C oracle ref && python tests/golden/make_golden_window.py

What is pinned and what is not: every window decode inside a fixture was done by the REAL reference BP + OSD
(oracle/_ref/libref_bp.so).  The window loop around them is the restatement in oracle/window_oracle.py -- the
reference's own module cannot be imported here (it starts with ``import stim``, which this image lacks), and the
reference has no test of it.  Models are phenomenological detector error models written as DEM text
(tests/window_util.py); shots are sampled from the model's own priors with numpy's PCG64.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
from oracle.window_oracle import WindowOracle  # noqa: E402
from ldpc_amd import codes  # noqa: E402
from window_util import phenomenological_dem, phenomenological_matrices, ring_code, sample_shots  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(name, h, rounds, p_data, p_meas, logical, *, decodings, window, commit, shots, seed, scale=1.0, **cfg):
    assert oracle.have_ref(), "make -C oracle ref first"
    assert (window - commit) + decodings * commit == rounds
    text = phenomenological_dem(h, rounds, p_data, p_meas, logical)
    check, obs, pri = phenomenological_matrices(h, rounds, p_data, p_meas, logical)
    synd, _ = sample_shots(check, np.minimum(pri * scale, 0.5), shots, seed)
    synd[0] = 0  # the all-zero shot takes BpOsdDecoder.decode's shortcut in every window
    w = WindowOracle(check, obs, pri, decodings=decodings, window=window, commit=commit, num_checks=h.shape[0], inner="ref", **cfg)
    preds, corrs, after = w.decode_batch(synd)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), name=name, dem_text=text, num_checks=h.shape[0], decodings=decodings,
                        window=window, commit=commit, config_keys=np.array(sorted(cfg)), config_vals=np.array([str(cfg[k]) for k in sorted(cfg)]),
                        shots=np.packbits(synd, axis=1, bitorder="little"), num_detectors=synd.shape[1],
                        predictions=preds, corrections=corrs, shots_after=np.packbits(after, axis=1, bitorder="little"),
                        priors_after=w.weights)
    print(f"{name}: {shots} shots, {check.shape[0]} detectors x {check.shape[1]} errors, corrections with weight "
          f"{corrs.sum(axis=1).mean():.2f} on average, {int(preds.sum())} flipped observables")


def main():
    ring = ring_code(8)
    run("window_ring8_d2_w4_c2", ring, 6, 0.04, 0.03, (0,), decodings=2, window=4, commit=2, shots=96, seed=1, max_iter=30)
    run("window_ring8_d4_w3_c1", ring, 6, 0.04, 0.03, (0, 3), decodings=4, window=3, commit=1, shots=96, seed=2, max_iter=20,
        bp_method="product_sum")
    run("window_ring8_single_window", ring, 6, 0.04, 0.03, (0,), decodings=1, window=6, commit=6, shots=64, seed=3, max_iter=30)
    run("window_ring8_d2_w4_c2_osdcs", ring, 6, np.linspace(0.02, 0.08, 8), 0.05, (0,), decodings=2, window=4, commit=2, shots=96,
        seed=4, scale=2.0, max_iter=4, osd_method="osd_cs", osd_order=4, ms_scaling_factor=0.625)
    ham = np.array([[1, 0, 1, 0, 1, 0, 1], [0, 1, 1, 0, 0, 1, 1], [0, 0, 0, 1, 1, 1, 1]], np.uint8)
    run("window_hamming_d3_w3_c2", ham, 7, 0.03, 0.02, (0, 1, 2), decodings=3, window=3, commit=2, shots=96, seed=5, scale=2.0, max_iter=10)
    bb = codes.bivariate_bicycle_hx()
    run("window_bb144_d2_w3_c2", bb, 5, 0.004, 0.004, tuple(range(12)), decodings=2, window=3, commit=2, shots=48, seed=6, scale=2.0,
        max_iter=20, ms_scaling_factor=0.625)
    run("window_bb144_d2_w6_c3", bb, 9, 0.004, 0.004, tuple(range(12)), decodings=2, window=6, commit=3, shots=40, seed=7, scale=2.5,
        max_iter=12, ms_scaling_factor=0.625)  # 432-row windows: OSD-0 with a workgroup per syndrome on the device


if __name__ == "__main__":
    main()
