#!/usr/bin/env python3
"""serial_relative (bp.hpp:469-483 + 477-540) on a code whose state is beyond LDS: the [[1600,64]] hypergraph-product code
(hx 768 x 1600, rows of 7, columns of 3 / 4).  One JSON line per kernel form:

    python tools/bench_relative.py [--batch 16384] [--p 0.02] [--method 1] [--forms all]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16384)
    ap.add_argument("--p", type=float, default=0.02)
    ap.add_argument("--method", type=int, default=1)
    ap.add_argument("--alpha", type=float, default=0.625)
    ap.add_argument("--max-iter", type=int, default=30)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--forms", default="all")
    ap.add_argument("--code", default="hgp1600", choices=["hgp1600", "surface21", "bb144"])
    args = ap.parse_args()
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    if args.code == "hgp1600":
        h = codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5))
    elif args.code == "surface21":
        h = codes.rotated_surface_code_x(21)
    else:
        h = codes.bivariate_bicycle_hx()
    m, n = h.shape
    forms = [("per-lane kernel (REL_LDS 0)", (("REL_LDS", 0),)), ("default", ())]
    if args.forms == "one":
        forms = forms[1:2]
    if args.forms == "all":
        forms += [("REL_GLOBAL 0 (all state in LDS)", (("REL_GLOBAL", 0),)), ("REL_GLOBAL 1", (("REL_GLOBAL", 1),)), ("REL_GLOBAL 2", (("REL_GLOBAL", 2),))]
    ref = None
    for label, switches in forms:
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, args.p), args.max_iter, args.method, args.alpha if args.method else 1.0)
        eng.set_schedule("serial_relative")
        try:
            for k, v in switches:
                eng.set_debug_switch(k, v)
        except Exception as exc:
            print(json.dumps({"form": label, "skipped": str(exc)[:100]}), flush=True)
            eng.close()
            continue
        s = eng.gen_bsc_syndromes(7, args.p, shot0=0, shots=args.batch, device="cuda:0")
        out = eng.decode_batch(s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = eng.decode_batch(s)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        it = out[2].cpu().numpy()
        cv = out[3].cpu().numpy().astype(bool)
        dec = out[0].cpu().numpy()
        if ref is None:
            ref = (dec, it, cv)
        same = bool(np.array_equal(dec, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(cv, ref[2]))
        print(json.dumps({"form": label, "code": args.code, "batch": args.batch, "p": args.p, "method": args.method, "syndromes_per_s": round(args.batch / ms * 1e3),
                          "ms_per_decode": round(ms, 2), "kernel_ms": round(eng.last_kernel_ms(), 2), "mean_iterations": round(float(it.mean()), 3),
                          "converged": round(float(cv.mean()), 5), "same_as_first_form": same}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
