#!/usr/bin/env python3
"""A/B of the two forms of the on-chip BP kernels (ldpc_hip_bp_set_small_code_kernel 4 / 5 / -1): one wavefront per syndrome, a
workgroup ("team") per syndrome, and what the library picks by itself.  Run on an MI355X:

    python tools/bench_team.py            # mid-size and small codes at large batches
    python tools/bench_team.py --small    # small codes at small batches (the latency regime)

Results are identical between the forms (tests/test_gpu_parity.py, tests/test_gpu_fuzz.py); this prints times only."""
import argparse
import os
import sys
import time

import numpy as np
import scipy.sparse as sp
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd import codes  # noqa: E402
from ldpc_amd.engine import HipBpEngine  # noqa: E402


def bench(name, h, p, it, method, alpha, batch, reps=5):
    h = sp.csr_matrix(h)
    m, n = h.shape
    res = []
    for mode in (4, 5, -1):
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), it, method, alpha)
        eng.set_small_code_kernel(mode)
        s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=batch, device="cuda:0")
        try:
            out = eng.decode_batch(s)
        except Exception:  # the form does not fit this code
            res.append(float("nan"))
            continue
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            out = eng.decode_batch(s, out=out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        res.append(float(np.median(ts)) * 1e3)
    print(f"{name:26s} {m:5d} x {n:<5d} p={p} B={batch:<7d} one wavefront {res[0]:8.3f} ms   team {res[1]:8.3f} ms   automatic {res[2]:8.3f} ms")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args()
    bb = codes.bivariate_bicycle_hx()
    if args.small:
        for d in (9, 13, 17):
            for batch in (512, 4096, 16384, 65536):
                bench(f"surface d={d} min-sum", codes.rotated_surface_code_x(d), 0.05, 30, 1, 0.625, batch, reps=9)
        for batch in (512, 4096, 16384, 65536):
            bench("BB144 min-sum 50", bb, 0.05, 50, 1, 0.625, batch, reps=9)
        return
    for cn, name in ((32, "HGP [[1600,64]]"), (24, "HGP [[900,36]]"), (16, "HGP [[400,16]]")):
        hg = codes.hypergraph_product_hx(codes.regular_ldpc_code(n=cn, dv=3, dc=4, seed=5))
        bench(name + " min-sum", hg, 0.02, 30, 1, 0.625, 65536)
        if cn == 32:
            bench(name + " product-sum", hg, 0.02, 30, 0, 1.0, 65536)
    for n in (600, 1200, 2400):
        bench(f"(3,6) n={n} min-sum", codes.regular_ldpc_code(n, 3, 6, seed=1), 0.05, 30, 1, 0.75, 32768)
        bench(f"(3,6) n={n} product-sum", codes.regular_ldpc_code(n, 3, 6, seed=1), 0.05, 30, 0, 1.0, 32768)
    bench("surface d=21 min-sum (C3)", codes.rotated_surface_code_x(21), 0.05, 30, 1, 0.625, 262144)
    bench("surface d=31 min-sum", codes.rotated_surface_code_x(31), 0.05, 30, 1, 0.625, 131072)
    bench("surface d=41 min-sum", codes.rotated_surface_code_x(41), 0.03, 30, 1, 0.625, 65536)


if __name__ == "__main__":
    main()
