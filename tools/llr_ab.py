#!/usr/bin/env python3
"""Does the log-ratio output cost the lane = edge kernels time?  Kernel ms with and without it at 262 144 syndromes (it does not).
Run on an MI355X:   python tools/llr_ab.py"""
import os, sys, json
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.getcwd())
from ldpc_amd import codes
from ldpc_amd.engine import HipBpEngine
for name, h, p, it in (("C3 p=0.05", codes.rotated_surface_code_x(21), 0.05, 30), ("C3 p=0.01", codes.rotated_surface_code_x(21), 0.01, 30), ("BB144 ms", codes.bivariate_bicycle_hx(), 0.05, 50)):
    h = sp.csr_matrix(h); n = h.shape[1]
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), it, 1, 0.625)
    eng.set_small_code_kernel(6)
    s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=262144, device="cuda:0")
    row = {}
    for want in (False, True):
        out = eng.decode_batch(s, want_llr=want)
        ks = []
        for _ in range(5):
            out = eng.decode_batch(s, out=out, want_llr=want, asynchronous=True)
            torch.cuda.synchronize()
            ks.append(eng.last_kernel_ms())
        row["llr" if want else "no llr"] = round(float(np.median(ks)), 4)
    print(json.dumps({"config": name, "kernel_ms": row}), flush=True)
