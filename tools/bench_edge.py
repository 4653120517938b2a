#!/usr/bin/env python3
"""A/B of bp_edge_kernel (lane = edge, messages in registers; ldpc_hip_bp_set_small_code_kernel 6) against the lane = node on-chip
kernel in its two forms (4: one wavefront per syndrome, 5: a workgroup per syndrome) on the surface-code family, with a
bit-for-bit comparison of every output.  Run on an MI355X:   python tools/bench_edge.py [--quick]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd import codes  # noqa: E402
from ldpc_amd.engine import HipBpEngine  # noqa: E402


def run(name, h, p, it, alpha, batch, reps=5, modes=(6, 4, 5)):
    h = sp.csr_matrix(h)
    m, n = h.shape
    res, outs = {}, {}
    for mode in modes:
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), it, 1, alpha)
        eng.set_small_code_kernel(mode)
        s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=batch, device="cuda:0")
        out = eng.decode_batch(s)
        torch.cuda.synchronize()
        ts, ks = [], []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = eng.decode_batch(s, out=out, asynchronous=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
            ks.append(eng.last_kernel_ms())
        res[mode] = (float(np.median(ts)) * 1e3, float(np.median(ks)))
        outs[mode] = [o.clone() for o in out]
        eng.close()
    same = all(bool(torch.equal(a.view(torch.int64) if a.dtype == torch.float64 else a, b.view(torch.int64) if b.dtype == torch.float64 else b))
               for mode in modes[1:] for a, b in zip(outs[modes[0]], outs[mode]))
    it_mean = float(outs[modes[0]][2].float().mean())
    print(json.dumps({"config": name, "m": m, "n": n, "p": p, "batch": batch, "mean_iterations": round(it_mean, 2),
                      "ms": {str(k): round(v[0], 4) for k, v in res.items()}, "kernel_ms": {str(k): round(v[1], 4) for k, v in res.items()},
                      "Msyndromes_per_s": {str(k): round(batch / v[0] / 1e3, 2) for k, v in res.items()}, "identical": same}), flush=True)
    return same


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    ok = True
    ok &= run("C3 surface d=21 min-sum 30", codes.rotated_surface_code_x(21), 0.05, 30, 0.625, 262144)
    ok &= run("C3 surface d=21 min-sum 30", codes.rotated_surface_code_x(21), 0.01, 30, 0.625, 262144)
    if not args.quick:
        for d, batch in ((5, 65536), (9, 65536), (13, 65536), (17, 65536), (25, 65536), (31, 65536)):
            ok &= run(f"surface d={d}", codes.rotated_surface_code_x(d), 0.05, 30, 0.625, batch)
        for batch in (64, 512, 4096):
            ok &= run("surface d=21 small batch", codes.rotated_surface_code_x(21), 0.05, 30, 0.625, batch, reps=9)
        ok &= run("surface d=21 adaptive alpha", codes.rotated_surface_code_x(21), 0.08, 30, 0.0, 65536)
        ok &= run("ring 200", codes.ring_code(200), 0.1, 40, 0.9, 65536)
        for batch in (512, 8192, 65536, 262144):  # bp_edge8_kernel: rows in 8-lane groups
            ok &= run("BB144 min-sum 50", codes.bivariate_bicycle_hx(), 0.05, 50, 0.625, batch)
    print("ALL IDENTICAL" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
