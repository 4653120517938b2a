#!/usr/bin/env python3
"""Throughput of overlapping-window decoding (ldpc_amd.ckt_noise.BpOsdOverlappingWindowDecoder.decode_batch).

Model: BB [[144,12,12]] hx measured for 12 rounds with phenomenological noise (tests/window_util.py), 3 windows of 6
rounds committing 3 (864 detectors x 2520 errors; a window is 432 x ~1370 after dropping untouched columns), min-sum
30 iterations + OSD-0 (the reference's defaults, ckt_noise/config.py).  Host arrays in, predictions out, so the figure
includes the PCIe copies.  (The same model is checked against the shot-by-shot checker, and that loop timed, in
tests/test_ckt_noise_gpu.py::test_bench_model_against_the_checker.)
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shots", type=int, default=32768)
    ap.add_argument("--p", type=float, default=0.003)
    ap.add_argument("--packed", action="store_true", help="bit-packed shots in, bit-packed predictions out (sinter's decode_shots_bit_packed)")
    ap.add_argument("--windowing", default="3,6,3", help="decodings,window,commit (12 rounds in all)")
    ap.add_argument("--bp-method", default="minimum_sum")
    ap.add_argument("--small-mode", type=int, default=None, help="ldpc_hip_bp_set_small_code_kernel for every window engine")
    args = ap.parse_args()
    from ldpc_amd import codes
    from ldpc_amd.ckt_noise import BpOsdOverlappingWindowDecoder
    from window_util import phenomenological_dem, phenomenological_matrices, sample_shots
    h = codes.bivariate_bicycle_hx()
    decodings, window, commit = (int(x) for x in args.windowing.split(","))
    rounds = (window - commit) + decodings * commit
    text = phenomenological_dem(h, rounds, args.p, args.p, logical=tuple(range(12)))
    check, obs, pri = phenomenological_matrices(h, rounds, args.p, args.p, logical=tuple(range(12)))
    shots, _ = sample_shots(check, pri, args.shots, seed=11)
    cfg = dict(max_iter=30, bp_method=args.bp_method, ms_scaling_factor=0.625)
    dec = BpOsdOverlappingWindowDecoder(text, decodings=decodings, window=window, commit=commit, num_checks=h.shape[0], decoder_config=cfg)
    dec.decode_batch(shots[:256].copy())  # builds the window decoders
    if args.small_mode is not None:
        for d in dec._decoders.values():
            d.inner._get_engine().set_small_code_kernel(args.small_mode)
        dec.decode_batch(shots[:256].copy())
    if args.packed:
        packed = np.packbits(shots, axis=1, bitorder="little")
        t0 = time.perf_counter()
        preds = dec.decode_batch(packed, bit_packed_shots=True, bit_packed_predictions=True)
        dt = time.perf_counter() - t0
        preds = np.unpackbits(preds, axis=1, bitorder="little", count=obs.shape[0]).astype(bool)
    else:
        work = shots.copy()
        t0 = time.perf_counter()
        preds = dec.decode_batch(work)
        dt = time.perf_counter() - t0
    out = {"config": f"BB144 x {rounds} rounds, {decodings} windows of {window} committing {commit}, min_sum 30 it + OSD-0, p={args.p}",
           "shots": args.shots, "packed_io": bool(args.packed), "detectors": check.shape[0], "errors": check.shape[1],
           "window_columns": [int(len(d.cols)) for d in dec._decoders.values()],
           "small_mode": args.small_mode, "bp_method": args.bp_method, "mean_iterations": [float(d.inner.iter_batch.float().mean()) for d in dec._decoders.values()], "bp_kernel_ms": [d.inner._get_engine().last_kernel_ms() for d in dec._decoders.values()], "shots_per_s": args.shots / dt, "seconds": dt, "flipped_observables": int(preds.sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
