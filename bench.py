#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one JSON line on rank 0.

Workload (configs[1]): (3,6)-regular LDPC, n = 10 000 (m = 5 000, E = 30 000), code seed 1;
product_sum flooding BP, max_iter = 50; batch = 65 536 syndromes PER GPU (weak scaling: N GPUs decode
N x 65 536); iid BSC errors p = 0.09 from the counter-based stream (error seed 7), syndromes = H e,
generated ON the device before the timed region, so inputs are resident in HBM when timing starts.

A "step" = one decode of the rank's whole batch through the C ABI (pack -> BP kernel -> unpack ->
LLR transpose) plus, for N > 1, the single gather of decoded rows onto rank 0 (RCCL over xGMI).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch-per-gpu B] [--p P]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes(iters: np.ndarray, m: int, n: int, nnz: int) -> float:
    """per syndrome  iters_run * 4*E*8  +  (m + n + 8n + 5)  bytes."""
    return float(np.sum(iters.astype(np.float64) * (4.0 * nnz * 8.0) + (m + n + 8.0 * n + 5.0)))


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=65536)
    ap.add_argument("--p", type=float, default=0.09, help="BSC error rate (0.09: primary point, 0.05: early-exit point)")
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--waves", type=int, default=0, help="waves per workgroup (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="syndromes for the CPU baseline and parity gate (-1 = max(1024, 4 per host thread), 0 = skip)")
    ap.add_argument("--math", default="libm_exact", choices=["libm_exact", "fast"], help="device tanh/log (include/ldpc_hip.h)")
    ap.add_argument("--bp-method", default="product_sum", choices=["product_sum", "minimum_sum"],
                    help="product_sum is the BASELINE workload; minimum_sum (alpha 0.625) is a diagnostic memory-only run")
    ap.add_argument("--handoff", type=int, default=-1, help="straggler hand-off threshold in tiles (-1 auto, 0 off; diagnostic)")
    ap.add_argument("--ring", type=int, default=1, help="LDS-DMA ring: 0 = register-prefetch variant, 1 = default depth, 2/3 = depth (diagnostic)")
    ap.add_argument("--no-llr", action="store_true", help="skip the LLR output (not the BASELINE workload)")
    ap.add_argument("--repack", type=int, default=-1, help="first-pass iterations of the repacked schedule (-1 = steered by the previous decode, 0 = off; diagnostic)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.sharding import gather_rows

    n = args.n
    h = regular_ldpc_code(n, 3, 6, seed=1)
    m, nnz = h.shape[0], h.nnz
    B = args.batch_per_gpu
    total = B * world
    method_id = 0 if args.bp_method == "product_sum" else 1
    alpha = 1.0 if method_id == 0 else 0.625
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, args.p), args.max_iter, method_id, alpha, device=local_rank)
    if args.waves:
        eng.set_tuning(waves_per_workgroup=args.waves)
    eng.set_math(args.math)
    eng.set_ring(args.ring)
    eng.set_handoff(args.handoff)
    eng.set_repack(args.repack)

    # inputs resident in HBM before the timed region; this rank's shard of the global shot stream
    synd = eng.gen_bsc_syndromes(7, args.p, shot0=rank * B, shots=B, device=dev)
    dec = torch.empty((B, n), dtype=torch.uint8, device=dev)
    llr = None if args.no_llr else torch.empty((B, n), dtype=torch.float64, device=dev)
    it = torch.empty((B,), dtype=torch.int32, device=dev)
    cv = torch.empty((B,), dtype=torch.uint8, device=dev)
    out = (dec, llr, it, cv)

    kernel_ms = []
    phase_ms = []

    def step(record: bool):
        eng.decode_batch(synd, want_llr=llr is not None, out=out, asynchronous=True)
        if record:
            kernel_ms.append(eng.last_kernel_ms())  # HIP events around the BP kernels on the launch stream
            phase_ms.append(eng.last_phase_ms())
        if world > 1:  # the only collective: gather decoded rows (bit-packed on the device first) + flags onto rank 0
            gather_rows(eng.pack_b8(dec), total, 0)
            gather_rows(cv, total, 0)
            gather_rows(it, total, 0)

    for _ in range(args.warmup):
        step(False)
    if world > 1 and args.warmup == 0:
        # RCCL sets up its peer-to-peer connections at the first gather: keep that out of the timed region
        gather_rows(torch.zeros((8, 8), dtype=torch.uint8, device=dev), 8 * world, 0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        iters = it.cpu().numpy()
        conv = cv.cpu().numpy().astype(bool)
        alg = algorithmic_bytes(iters, m, n, nnz)
        k_ms = float(np.mean(kernel_ms))
        pers_ms = float(np.mean([p[0] for p in phase_ms]))
        spread_ms = float(np.mean([p[1] for p in phase_ms]))
        achieved = alg / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch from the committed PMC run of this same workload (profiles/hbm_traffic.json,
        # produced by tools/profile_bench.sh + tools/prof_parse.py); null when absent or for another workload
        traffic = None
        tj = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tj) and world == 1 and method_id == 0:
            try:
                t = json.load(open(tj))
                if t.get("batch_per_gpu") == B and t.get("p") == args.p and t.get("max_iter") == args.max_iter and t.get("n") == n:
                    traffic = t["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        res = {
            "metric": "syndromes_per_sec_batched_bp50_product_sum_ldpc36_n10k" if method_id == 0 else "syndromes_per_sec_DIAGNOSTIC_min_sum",
            "value": total * args.steps / elapsed,
            "unit": "syndromes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": f"configs[1]: (3,6)-regular LDPC n={n} (m={m}, E={nnz}), {args.bp_method} flooding BP, "
                            f"max_iter={args.max_iter}, batch={B} syndromes per GPU, BSC p={args.p}, code seed 1, error seed 7",
                "batch_per_gpu": B, "global_batch": total, "p": args.p, "max_iter": args.max_iter,
                "outputs": "decoding u8, log_prob_ratios f64, iterations i32, converge u8" if llr is not None else "no LLR",
                "parallelism": f"batch-sharded x{world}, one gather of decoded rows" if world > 1 else "single GPU",
                "device_math": args.math, "mean_iterations": float(iters.mean()), "converged_fraction": float(conv.mean()),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": "bp_decode_kernel (persistent, one workgroup per 64-syndrome tile) + bp_spread_* per-pass launches for the last tiles",
                "kernel_ms": k_ms, "kernel_ms_persistent": pers_ms, "kernel_ms_per_pass": spread_ms,
                "algorithmic_bytes_per_launch": alg,
                "note": "one decode = the whole batch; algorithmic bytes = sum over syndromes of iters_run*4*E*8 + (m+n+8n+5); "
                        "kernel_ms = HIP events on the launch stream around all BP kernels of the decode (the persistent kernel "
                        "hands its last <= 256 tiles to per-pass launches: compare kernel_ms_persistent with rocprofv3's "
                        "bp_decode_kernel average and kernel_ms_per_pass with the sum of the bp_spread_* kernels); traffic = "
                        "PMC HBM bytes of the same kernels (profiles/hbm_traffic.json)",
            },
        }
        # ---- CPU baseline + parity gate on a bounded sample of THIS batch (rank 0, N = 1 only) ----
        if world == 1 and args.cpu_sample != 0:
            from oracle import cpu_bench  # checker / baseline only
            from ldpc_amd.noise_models import generate_bsc_batch
            cores = os.cpu_count() or 1
            sample = args.cpu_sample if args.cpu_sample > 0 else min(B, max(1024, 4 * cores))
            # the device shot generator against its host twin on the first rows ...
            head = min(B, 64)
            err = generate_bsc_batch(n, args.p, 7, 0, head)
            s_head = np.asarray((h.astype(np.int32) @ err.T.astype(np.int32)).T % 2, dtype=np.uint8)
            assert np.array_equal(s_head, synd[:head].cpu().numpy()), "device shot generator differs from its host twin"
            # ... and the decoder against the CPU reference on a random subset of the timed batch
            rows = np.sort(np.random.default_rng(12345).choice(B, size=sample, replace=False))
            rows_t = torch.from_numpy(rows).to(dev)
            s_host = synd[rows_t].cpu().numpy()
            cpu, (cd, cl, ci, cc) = cpu_bench.run(h, args.p, args.max_iter, args.bp_method, alpha, s_host, cores=cores)
            one, _ = cpu_bench.run(h, args.p, args.max_iter, args.bp_method, alpha, s_host[:6], cores=1)  # (i) one thread alone
            cpu["single_thread"] = {"value": one["value"], "unit": "syndromes/s", "sample": one["sample"]}
            gd = dec[rows_t].cpu().numpy()
            ok = bool(np.array_equal(gd, cd) and np.array_equal(iters[rows], ci) and np.array_equal(conv[rows], cc))
            if llr is not None:
                gl = llr[rows_t].cpu().numpy()
                from oracle import llr_close
                ok = ok and llr_close(gl, cl, rtol=1e-5)
            cpu["parity_vs_gpu"] = {"syndromes": int(sample), "subset": "random rows of the timed batch (seed 12345)",
                                    "hard_decisions_iters_converge_exact_llr_1e-5": ok}
            res["cpu_baseline"] = cpu
            if not ok:
                res["parity_failed"] = True
        else:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
