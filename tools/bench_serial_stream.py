#!/usr/bin/env python3
"""Serial schedule on the BASELINE configs[1] code ((3,6)-regular n = 10 000, product-sum, 50 iterations, p = 0.05, B = 65 536): the
level-parallel kernel against the forms of the streamed one (bp_serial_stream_kernel.h).  One JSON line per form:

    python tools/bench_serial_stream.py [--batch 65536] [--p 0.05] [--forms all|default]
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
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--p", type=float, default=0.05)
    ap.add_argument("--method", type=int, default=0)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--forms", default="all")
    ap.add_argument("--code", default="ldpc36", choices=["ldpc36", "ldpc48", "irregular"],
                    help="ldpc36: (3,6)-regular (configs[1]); ldpc48: (4,8)-regular (rows of 8, columns of 4); irregular: rows of 3 .. 16, columns of 2 .. 8 -- all n = 10 000")
    args = ap.parse_args()
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = {"ldpc36": lambda: codes.regular_ldpc_code(10000, 3, 6, seed=1), "ldpc48": lambda: codes.regular_ldpc_code(10000, 4, 8, seed=1),
         "irregular": lambda: codes.irregular_ldpc_code(10000, 5000, seed=1)}[args.code]()
    m, n = h.shape
    row_deg = np.diff(h.indptr).astype(np.float64)
    forms = [  # (label, serial_kernel, repack, switches, want_llr)
        ("level kernel (round 1-4)", 1, -1, (), True),
        ("streamed, one pass, 16 waves ring 1", 2, 0, (), True),
        ("streamed, passes (default)", 2, -1, (), True),
        ("streamed, passes, no log-ratios", 2, -1, (), False),
        ("streamed, passes, later passes 8 waves", 2, -1, (("SER_WAVES2", 8),), True),
        ("streamed, passes, lanes <= 512 rows", 2, -1, (("SER_LANE_MAX", 512),), True),
        ("streamed, passes, lanes <= 8192 rows", 2, -1, (("SER_LANE_MAX", 8192),), True),
        ("streamed, first pass 5", 2, 5, (), True),
        ("streamed, first pass 3", 2, 3, (), True),
    ]
    if args.code != "ldpc36":  # the item form (bp_serial_var_kernel.h) against the level kernel it replaces
        forms = [("level kernel (before round 6)", 2, -1, (("SER_VAR", 0),), True),
                 ("items, one pass", 2, 0, (), True),
                 ("items, passes (default)", 2, -1, (), True),
                 ("items, passes, no log-ratios", 2, -1, (), False),
                 ("items, passes, 12 KiB queues", 2, -1, (("SER_VAR_UNITS", 12), ("SER_WAVES", 12)), True),
                 ("items, passes, 16 KiB queues", 2, -1, (("SER_VAR_UNITS", 16), ("SER_WAVES", 9)), True),
                 ("items, passes, 8 wavefronts", 2, -1, (("SER_WAVES", 8),), True),
                 ("items, first pass 3", 2, 3, (), True)]
    elif args.forms == "all":
        forms.append(("items (SER_VAR 1), passes", 2, -1, (("SER_VAR", 1),), True))
    if args.forms == "default":
        forms = forms[:1] + forms[2:4]
    if args.forms == "one":
        forms = forms[2:3]  # (the default form of either list: passes)
    ref = None
    for label, mode, repack, switches, want_llr in forms:
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, args.p), 50, args.method, args.alpha)
        eng.set_schedule("serial")
        eng.set_serial_kernel(mode)
        eng.set_repack(repack)
        for k, v in switches:
            eng.set_debug_switch(k, v)
        s = eng.gen_bsc_syndromes(7, args.p, shot0=0, shots=args.batch, device="cuda:0")
        out = eng.decode_batch(s, want_llr=want_llr)  # warm-up (and the histogram that steers the next call)
        out = eng.decode_batch(s, want_llr=want_llr, out=out)
        torch.cuda.synchronize()
        c0 = eng.clock_probe()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.decode_batch(s, want_llr=want_llr, out=out)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / args.steps * 1e3
        c1 = eng.clock_probe()
        it = out[2].cpu().numpy()
        cv = out[3].cpu().numpy().astype(bool)
        dec = out[0].cpu().numpy()
        if ref is None:
            ref = (dec, it, cv)
        same = bool(np.array_equal(dec, ref[0]) and np.array_equal(it, ref[1]) and np.array_equal(cv, ref[2]))
        alg = float(np.sum(it.astype(np.float64) * 4.0 * h.nnz * 8.0 + (m + n + 8.0 * n + 5.0)))
        # what the schedule itself moves per lane and iteration: every entry read once by each other bit of its row, written once = sum d^2 segments
        moved = float(np.sum(it.astype(np.float64) * float(np.sum(row_deg * row_deg)) * 8.0))
        print(json.dumps({"form": label, "code": args.code, "batch": args.batch, "p": args.p, "syndromes_per_s": round(args.batch / ms * 1e3), "ms_per_decode": round(ms, 2),
                          "kernel_ms": round(eng.last_kernel_ms(), 2), "mean_iterations": round(float(it.mean()), 3), "converged": round(float(cv.mean()), 5),
                          "hbm_frac_4E_bytes": round(alg / (ms * 1e-3) / 8e12, 4), "hbm_frac_serial_bytes_per_lane_iteration": round(moved / (ms * 1e-3) / 8e12, 4),
                          "clock_ghz": round(HipBpEngine.clock_ghz(c0, c1) or 0.0, 3), "same_as_first_form": same}), flush=True)
        eng.close()


if __name__ == "__main__":
    main()
