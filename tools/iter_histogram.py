#!/usr/bin/env python3
"""Iteration histogram of the headline code at an operating point (default p = 0.05), and what tile-wise execution costs:
mean iterations per syndrome vs mean of the per-tile maximum (64 consecutive syndromes).  Run on an MI355X."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd.codes import regular_ldpc_code  # noqa: E402
from ldpc_amd.engine import HipBpEngine  # noqa: E402

p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.05
B = 65536
h = regular_ldpc_code(10000, 3, 6, seed=1)
eng = HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, p), 50, 0, 1.0)
s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device="cuda:0")
out = eng.decode_batch(s, want_llr=False)
it = out[2].cpu().numpy()
cv = out[3].cpu().numpy().astype(bool)
hist = np.bincount(it, minlength=51)
tile_max = it.reshape(-1, 64).max(axis=1)
print(json.dumps({"p": p, "batch": B, "mean_iterations": float(it.mean()), "converged": float(cv.mean()), "mean_tile_max": float(tile_max.mean()),
                  "histogram": {str(k): int(v) for k, v in enumerate(hist) if v}}))
