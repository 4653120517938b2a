#!/usr/bin/env python3
"""Work distribution of the lane = edge kernels: share of the batch handed out statically (EDGE_STATIC_PCT) x chunk of the
work-counter pulls (EDGE_CHUNK), kernel time by HIP events.  Run on an MI355X:   python tools/sweep_edge_split.py"""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd import codes  # noqa: E402
from ldpc_amd.engine import HipBpEngine  # noqa: E402


def main():
    cases = [("C3 d=21 p=0.05", codes.rotated_surface_code_x(21), 0.05, 30), ("C3 d=21 p=0.01", codes.rotated_surface_code_x(21), 0.01, 30),
             ("surface d=9 p=0.05", codes.rotated_surface_code_x(9), 0.05, 30), ("BB144 p=0.05", codes.bivariate_bicycle_hx(), 0.05, 50)]
    for name, h, p, it in cases:
        h = sp.csr_matrix(h)
        n = h.shape[1]
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), it, 1, 0.625)
        eng.set_small_code_kernel(6)
        for batch in (4096, 16384, 65536, 262144):
            s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=batch, device="cuda:0")
            out = eng.decode_batch(s)
            row = {}
            for pct, chunk in ((-1, 0), (0, 1), (0, 2), (0, 4), (25, 1), (50, 1), (75, 1), (100, 1)):
                eng.set_debug_switch("EDGE_STATIC_PCT", pct)
                eng.set_debug_switch("EDGE_CHUNK", chunk if chunk else -1)
                ks = []
                for _ in range(5):
                    out = eng.decode_batch(s, out=out, asynchronous=True)
                    torch.cuda.synchronize()
                    ks.append(eng.last_kernel_ms())
                row[f"{'default' if pct < 0 else pct}/{chunk or 'auto'}"] = round(float(np.median(ks)), 4)
            print(json.dumps({"config": name, "batch": batch, "kernel_ms by static_pct/chunk": row}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
