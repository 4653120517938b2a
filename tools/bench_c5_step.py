#!/usr/bin/env python3
"""BASELINE config 5's step (BB [[144,12,12]] hx, product_sum 50 iterations + OSD-0, B = 8192, p = 0.05) timed closely: K decode calls
queued back to back between two device synchronisations, several rounds, best / median; with --check the outputs (decisions, log-ratios,
iterations, convergence, OSD status) are compared bit for bit with those of the library named by LDPC_HIP_LIB_REF in a child process.

    python tools/bench_c5_step.py [--steps 200] [--rounds 5] [--batch 8192] [--dump out.npz]
Same-box A/B of library builds: run it once per library (LDPC_HIP_LIB=ldpc_amd/lib/variants/<name>.so), interleaved, as tools/ab_bench.sh does."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--batch", type=int, default=8192)
    ap.add_argument("--dump", default=None, help="write the outputs of one decode to this .npz (for a bit-for-bit comparison between libraries)")
    ap.add_argument("--bp-only", action="store_true")
    args = ap.parse_args()
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    h = codes.bivariate_bicycle_hx()
    m, n = h.shape
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.05), 50, 0, 1.0)
    s = eng.gen_bsc_syndromes(7, 0.05, shot0=0, shots=args.batch, device="cuda:0")
    osd0 = not args.bp_only
    out = eng.decode_batch(s, osd0=osd0)
    torch.cuda.synchronize()
    times = []
    for _ in range(args.rounds):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.decode_batch(s, out=out, osd0=osd0, asynchronous=True)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) / args.steps * 1e3)
    rec = {"config": "c5 BB144 product_sum 50 it" + (" + OSD-0" if osd0 else " (BP only)"), "batch": args.batch, "steps": args.steps,
           "ms_best": round(min(times), 5), "ms_median": round(float(np.median(times)), 5), "ms_rounds": [round(t, 5) for t in times],
           "bp_kernel_ms": round(eng.last_kernel_ms(), 5), "bp_converged": float(out[3].float().mean()),
           "lib": os.environ.get("LDPC_HIP_LIB", "default")}
    if args.dump:
        st = eng.osd_status(args.batch) if osd0 else np.zeros(0, np.uint8)
        np.savez(args.dump, dec=out[0].cpu().numpy(), llr=out[1].cpu().numpy().view(np.uint64), it=out[2].cpu().numpy(), cv=out[3].cpu().numpy(), status=st)
    print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
