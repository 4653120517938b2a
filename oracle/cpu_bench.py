"""Timed CPU baseline for bench.py -- TEST INFRASTRUCTURE, not product code.

Decodes a bounded sample of the benchmark's own syndromes with the CPU checker, one decoder instance
per host thread over disjoint slices (the reference object is single-threaded and non-reentrant,
bp.hpp:136-140, so this is how a user would use all cores).  ``kind`` is ``"reference"`` when the
real reference build (oracle/_ref/libref_bp.so, compiled from the reference's headers in the build
container) is present, else ``"port"`` (oracle/libbp_oracle.so, the bit-exact C restatement).
ctypes releases the GIL around the foreign call, so plain threads run in parallel.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

import oracle


def run(h, error_rate, max_iter, bp_method, ms_scaling_factor, syndromes, cores=None, prefer_reference=True):
    """Decode ``syndromes`` (S, m) split over ``cores`` threads; returns (result dict, outputs)."""
    cores = int(cores or os.cpu_count() or 1)
    s = np.ascontiguousarray(syndromes, np.uint8)
    cores = max(1, min(cores, len(s)))
    use_ref = prefer_reference and oracle.have_ref()
    cls = oracle.RefBp if use_ref else oracle.BpOracle
    decoders = [cls(h, error_rate=error_rate, max_iter=max_iter, bp_method=bp_method,
                    ms_scaling_factor=ms_scaling_factor) for _ in range(cores)]
    bounds = np.linspace(0, len(s), cores + 1).astype(int)

    def work(k):
        return decoders[k].decode_batch(s[bounds[k]:bounds[k + 1]])

    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        parts = list(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    dec = np.concatenate([p[0] for p in parts])
    llr = np.concatenate([p[1] for p in parts])
    it = np.concatenate([p[2] for p in parts])
    cv = np.concatenate([p[3] for p in parts])
    res = {
        "value": len(s) / dt, "unit": "syndromes/s", "cores": cores,
        "kind": "reference" if use_ref else "port",
        "sample": f"{len(s)} syndromes of the timed batch, {cores} threads x one decoder each, {dt:.2f} s wall",
        "per_core": len(s) / dt / cores,
    }
    return res, (dec, llr, it, cv)
