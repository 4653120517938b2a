#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config, one JSON line on rank 0.

Workload (configs[1] / configs[3]): (3,6)-regular LDPC, n = 10 000 (m = 5 000, E = 30 000), code seed 1;
product_sum flooding BP, max_iter = 50; iid BSC errors p = 0.09 from the counter-based stream (error seed 7),
syndromes = H e, generated ON the device before the timed region (inputs resident in HBM when timing starts).

    N = 1   batch = 65 536 syndromes                      = configs[1]
    N > 1   batch = 131 072 syndromes PER GPU (weak scaling; N = 8: 1 048 576 syndromes = configs[3]); rank r decodes
            shots [r B, (r + 1) B) of the same stream.  Throughput per GPU does not depend on B at these sizes
            (profiles/: 8 192 ... 65 536 within the box-to-box spread), so N = 1 is comparable with the rest.

A "step" = one decode of the rank's whole batch through the C ABI (pack -> BP kernels -> unpack -> LLR transpose)
plus, when a process group exists, the single gather of decoded rows (bit-packed on the device) + flags onto rank 0
(RCCL over xGMI).

    python bench.py [--gpus N] [--steps K] [--warmup W]        # N > 1: spawns its N ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

After the headline, at N = 1, the two on-chip BASELINE configs (and schedule = serial_relative on the same two codes: `schedule_configs`) are timed as well (`secondary`): configs[2] (rotated
surface code d = 21, min-sum 30, B = 262 144) and configs[4] (BB [[144,12,12]] product-sum 50 + OSD-0, B = 8 192).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
LDS_PEAK_GBPS = 150000.0  # ds_read_b64 / b128 aggregate at 2.4 GHz, same guide (LDS table)
SIMDS = 1024             # 256 CUs x 4


def algorithmic_bytes(iters: np.ndarray, m: int, n: int, nnz: int) -> float:
    """per syndrome  iters_run * 4*E*8  +  (m + n + 8n + 5)  bytes."""
    return float(np.sum(iters.astype(np.float64) * (4.0 * nnz * 8.0) + (m + n + 8.0 * n + 5.0)))


def kernel_sources_sha16() -> str:
    """Fingerprint of the kernel sources (ldpc_amd/csrc): the committed PMC files under profiles/ carry the fingerprint of the build
    they were measured on, so a line can say whether its copied counters belong to THIS build."""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(ROOT, "ldpc_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "ldpc_amd", "csrc", "*.hip"))):
        hsh.update(os.path.basename(path).encode())
        hsh.update(open(path, "rb").read())
    return hsh.hexdigest()[:16]


def physical_cores() -> int:
    """Distinct (package, core) pairs of /proc/cpuinfo; falls back to the logical count."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return os.cpu_count() or 1


class SclkSampler:
    """Corroboration for the device-side clock probe: the driver's own reading of the shader clock (hwmon freq1_input, Hz), sampled by a
    host thread every 20 ms while a timed region runs.  Absent file -> no samples -> None."""

    def __init__(self, device_index: int = 0):
        import glob
        self.paths = sorted(glob.glob(f"/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
        self.path = self.paths[min(device_index, len(self.paths) - 1)] if self.paths else None
        self.samples, self._stop, self._thread = [], False, None

    def _run(self):
        while not self._stop:
            try:
                self.samples.append(float(open(self.path).read().strip()))
            except Exception:
                pass
            time.sleep(0.02)

    def __enter__(self):
        if self.path:
            import threading
            self._stop = False
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()
        return self

    def __exit__(self, *exc):
        self._stop = True
        if self._thread:
            self._thread.join(timeout=1.0)

    def ghz(self):
        return float(np.mean(self.samples)) / 1e9 if self.samples else None


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n_ranks: int, argv: list[str], port: int | None = None) -> list[str]:
    """The command the driver uses for N > 1, built here when bench.py is started bare (`python bench.py --gpus N`)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()), os.path.abspath(__file__), *argv]


RANK_ENV_DEFAULTS = {
    "HSA_ENABLE_IPC_MODE_LEGACY": "0",  # dmabuf IPC: without it RCCL fails with `hipIpcGetMemHandle: invalid argument` on this pool
    "OMP_NUM_THREADS": "1",
}


def apply_rank_env_defaults(env=None) -> dict:
    """Environment every rank needs BEFORE `import torch` / the first HIP call -- set in main() itself, so it holds under the
    driver's own `python -m torch.distributed.run ... bench.py --gpus N` as well as under self_launch()."""
    env = os.environ if env is None else env
    for k, v in RANK_ENV_DEFAULTS.items():
        env.setdefault(k, v)
    return env


def self_launch(n_ranks: int) -> int:
    argv = [a for a in sys.argv[1:] if a != "--force-launch"]
    return subprocess.call(launch_command(n_ranks, argv), env=apply_rank_env_defaults(dict(os.environ)))


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch-per-gpu", type=int, default=-1, help="-1: 65536 at N = 1 (configs[1]), 131072 at N > 1 (configs[3] at N = 8)")
    ap.add_argument("--p", type=float, default=0.09, help="BSC error rate (0.09: primary point, 0.05: early-exit point)")
    ap.add_argument("--max-iter", type=int, default=50)
    ap.add_argument("--n", type=int, default=10000)
    ap.add_argument("--waves", type=int, default=0, help="waves per workgroup (0 = library default)")
    ap.add_argument("--cpu-sample", type=int, default=-1, help="syndromes for the CPU baseline and parity gate at N = 1 (-1 = 1024, 0 = skip)")
    ap.add_argument("--rank-parity", type=int, default=128, help="rows of EVERY rank's shard checked against the CPU checker when N > 1 (0 = skip)")
    ap.add_argument("--math", default="libm_exact", choices=["libm_exact", "fast"], help="device tanh/log (include/ldpc_hip.h)")
    ap.add_argument("--bp-method", default="product_sum", choices=["product_sum", "minimum_sum"],
                    help="product_sum is the BASELINE workload; minimum_sum (alpha 0.625) is a diagnostic memory-only run")
    ap.add_argument("--handoff", type=int, default=-1, help="straggler hand-off threshold in tiles (-1 auto, 0 off; diagnostic)")
    ap.add_argument("--ring", type=int, default=1, help="LDS-DMA ring: 0 = register-prefetch variant, 1 = default depth, 2/3 = depth (diagnostic)")
    ap.add_argument("--no-llr", action="store_true", help="skip the LLR output (not the BASELINE workload)")
    ap.add_argument("--repack", type=int, default=-1, help="first-pass iterations of the repacked schedule (-1 = steered by the previous decode, 0 = off; diagnostic)")
    ap.add_argument("--secondary", type=int, default=1, help="1: also time configs[2] and configs[4] (N = 1 only); 0: skip")
    ap.add_argument("--host-io", type=int, default=1, help="1: also time the same batch through BpDecoder.decode_batch with pageable NumPy in / out (N = 1 only); 0: skip")
    ap.add_argument("--dry-ranks", action="store_true", help="launcher self-test: every rank joins a gloo group, rank 0 prints the ranks it saw; no GPU work")
    ap.add_argument("--force-launch", action="store_true", help="go through torch.distributed.run even for --gpus 1 (exercises the N > 1 code path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="diagnostic for a one-GPU box: the N ranks all use cuda:0 and form a gloo group (RCCL refuses two ranks on "
                    "one device), collectives carry host tensors -- exercises the N > 1 bookkeeping on hardware; the rate means nothing")
    return ap.parse_args()


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1): the real reference (oracle/_ref) or the C restatement, on a bounded sample of the batch
# ------------------------------------------------------------------------------------------------------------------
def cpu_baseline(h, p, max_iter, method, alpha, s_host):
    """Thread-count sweep on small slices, then the whole sample at the best count (whose outputs are the parity reference)."""
    from oracle import cpu_bench  # checker / baseline only
    logical, phys = os.cpu_count() or 1, physical_cores()
    counts = sorted({c for c in (1, 8, 16, 32, 64, phys // 2, phys, logical) if 1 <= c <= logical})
    sweep = []
    for c in counts:
        k = min(len(s_host), max(6, 2 * c))
        r, _ = cpu_bench.run(h, p, max_iter, method, alpha, s_host[:k], cores=c)
        sweep.append({"threads": r["cores"], "value": r["value"], "per_thread": r["per_core"], "syndromes": k})
    best = max(sweep, key=lambda e: e["value"])
    res, outs = cpu_bench.run(h, p, max_iter, method, alpha, s_host, cores=best["threads"])
    res["host"] = {"logical_cpus": logical, "physical_cores": phys}
    res["thread_sweep"] = sweep
    res["single_thread"] = {"value": sweep[0]["value"], "unit": "syndromes/s"}
    res["note"] = ("one decoder object per thread over disjoint slices (the reference object is single-threaded); `value` is the "
                   "best thread count of the sweep -- the linked-list matrix of every decoder (1.7 MB) competes for cache, so "
                   "per-thread speed falls as threads are added")
    return res, outs


# ------------------------------------------------------------------------------------------------------------------
# secondary configs: on-chip kernels, so the bound is LDS / FP64 VALU, not HBM
# ------------------------------------------------------------------------------------------------------------------
def secondary_configs(dev, steps):
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    import oracle  # checker only

    out = []
    valu = {}
    try:
        valu = json.load(open(os.path.join(ROOT, "profiles", "secondary_valu.json")))
    except Exception:
        valu = {}
    specs = [
        dict(key="c3", name="configs[2]: rotated surface code d=21 X checks (220 x 441, E=840), minimum_sum alpha=0.625, max_iter=30, "
                            "batch=262144, BSC p=0.05", h=codes.rotated_surface_code_x(21), p=0.05, max_iter=30, method=1,
             alpha=0.625, batch=262144, osd0=False, dr=4, dc=2),
        dict(key="c5", name="configs[4]: BB [[144,12,12]] hx (72 x 144, E=432), product_sum max_iter=50 + OSD-0, batch=8192, BSC p=0.05",
             h=codes.bivariate_bicycle_hx(), p=0.05, max_iter=50, method=0, alpha=1.0, batch=8192, osd0=True, dr=6, dc=3),
    ]
    for sp in specs:
        h, p, B = sp["h"], sp["p"], sp["batch"]
        m, n, nnz = h.shape[0], h.shape[1], h.nnz
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), sp["max_iter"], sp["method"], sp["alpha"], device=dev.index or 0)
        s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
        res = eng.decode_batch(s, osd0=sp["osd0"])  # warm-up; also the outputs that are checked
        torch.cuda.synchronize()
        kms, step_ms = [], []
        probe0 = eng.clock_probe()
        with SclkSampler(dev.index or 0) as sclk:
            for _ in range(steps):  # each decode timed on its own, the MEDIAN reported: these are sub-millisecond .. 10 ms calls, and one
                t0 = time.perf_counter()  # hiccup of the box (seen once: a 0.75 ms kernel taking 7.5 ms) would otherwise be the result
                eng.decode_batch(s, out=res, osd0=sp["osd0"], asynchronous=True)
                kms.append(eng.last_kernel_ms())
                torch.cuda.synchronize()
                step_ms.append((time.perf_counter() - t0) * 1e3)
        clock_run = eng.clock_ghz(probe0, eng.clock_probe())  # shader clock of the BP kernel's workgroups over these steps (device counters)
        ms = float(np.median(step_ms))
        # the same decodes queued back to back with ONE synchronisation behind them: what a caller that keeps the stream busy pays per decode
        # (`ms` above carries a launch and a host synchronisation per decode, ~40 us, which matters for the sub-millisecond entries only)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max(steps, 10)):
            eng.decode_batch(s, out=res, osd0=sp["osd0"], asynchronous=True)
        torch.cuda.synchronize()
        ms_queued = (time.perf_counter() - t0) * 1e3 / max(steps, 10)
        it = res[2].cpu().numpy()
        cv = res[3].cpu().numpy().astype(bool)
        # parity: a sample of the timed batch against the CPU checker (bit-exact decisions / iterations / flags, LLR 1e-5)
        rows = np.sort(np.random.default_rng(4321).choice(B, size=256, replace=False))
        rt = torch.from_numpy(rows).to(dev)
        s_host = s[rt].cpu().numpy()
        orc = oracle.BpOracle(h, error_rate=p, max_iter=sp["max_iter"], bp_method=sp["method"], ms_scaling_factor=sp["alpha"])
        od, ol, oi, oc = orc.bposd0_decode_batch(s_host) if sp["osd0"] else orc.decode_batch(s_host)
        ok = bool(np.array_equal(res[0][rt].cpu().numpy(), od) and np.array_equal(it[rows], oi) and np.array_equal(cv[rows], oc)
                  and oracle.llr_close(res[1][rt].cpu().numpy(), ol, rtol=1e-5))
        iters_total = float(it.astype(np.float64).sum())
        io_bytes = B * (m + n + 8.0 * n + 5.0)
        entry = {"config": sp["name"], "key": sp["key"], "value": B / ms * 1e3, "unit": "syndromes/s", "ms": ms, "ms_steps": [round(v, 4) for v in step_ms], "ms_queued_back_to_back": ms_queued,
                 "bp_kernel_ms": float(np.median(kms)),
                 "mean_iterations": float(it.mean()), "bp_converged_fraction": float(cv.mean()),
                 "io_hbm_GBps": io_bytes / (ms * 1e-3) / 1e9, "parity_vs_oracle": ok,
                 "algorithmic_message_bytes_per_s_GBps": iters_total * 4.0 * nnz * 8.0 / (ms * 1e-3) / 1e9}
        if sp["method"] == 1:
            # min-sum on chip (bp_edge_kernel: lane = edge, messages in registers).  The kernel is bound by instruction ISSUE: a SIMD
            # issues one vector and one scalar instruction per 4-cycle turn, and the lane-mask parities make the scalar stream as
            # long as the vector one.  Instructions per syndrome-iteration come from the committed PMC profile of this kernel
            # (profiles/secondary_c3.json: SQ_INSTS_VALU, SQ_INSTS_SALU, SQ_LDS_IDX_ACTIVE); the
            # fractions below use THIS run's kernel time and THIS run's clock.  The larger one is the bound.
            try:
                c3 = json.load(open(os.path.join(ROOT, "profiles", "secondary_c3.json")))["c3"]
            except Exception:
                c3 = {}
            k_s = float(np.median(kms)) * 1e-3
            if c3.get("valu_insts_per_syndrome_iteration") and c3.get("batch") == B and clock_run:
                slots = SIMDS * clock_run * 1e9 * k_s / 4.0  # issue turns on offer while the kernel ran, at the clock its workgroups measured in THIS run
                fv = iters_total * c3["valu_insts_per_syndrome_iteration"] / slots
                fs = iters_total * c3["salu_insts_per_syndrome_iteration"] / slots
                bounds = {"valu_issue": fv, "salu_issue": fs, "lds_array_busy": c3.get("lds_array_busy_frac_measured")}
                which = "valu_issue" if fv >= fs else "salu_issue"
                entry.update({"bound": which, "frac": bounds[which], "frac_measured_in_profile": c3.get(which + "_frac_measured"), "bounds": bounds, "kernel": c3.get("kernel"),
                              "clock_ghz_this_run": clock_run, "clock_ghz_in_profile": c3.get("clock_ghz"), "sclk_sysfs_ghz": sclk.ghz(),
                              "valu_insts_per_syndrome_iteration": c3["valu_insts_per_syndrome_iteration"],
                              "salu_insts_per_syndrome_iteration": c3["salu_insts_per_syndrome_iteration"],
                              "lds_bank_conflict_share": c3.get("lds_bank_conflict_share"),
                              "bound_note": "issue turns used / turns on offer = instructions per syndrome-iteration (profiles/secondary_c3.json, valid for this build when "
                                            "counters_match_this_build) x this run's iterations / (1024 SIMDs x clock_ghz_this_run x this run's kernel time / 4 cycles); "
                                            "clock_ghz_this_run = shader cycles / constant-rate ticks summed over the kernel's workgroups in this run "
                                            "(ldpc_hip_bp_clock_probe); not capped"})
            else:
                entry.update({"bound": "valu_issue", "frac": None, "clock_ghz_this_run": clock_run,
                              "bound_note": "profiles/secondary_c3.json absent or for another batch, or no clock reading"})
        else:
            # product-sum on chip: FP64 VALU issue.  Instructions per entry-iteration come from the committed PMC profile of
            # this kernel (profiles/secondary_valu.json: SQ_INSTS_VALU / (entries x iterations)); a wave-instruction takes
            # 4 cycles of its SIMD, and the chip has 1024 SIMDs at the clock this run's workgroups measured.
            per = valu.get(sp["key"], {}).get("valu_insts_per_entry_iteration")
            if per and clock_run:
                wave_insts = iters_total * nnz * per / 64.0
                frac = wave_insts * 4.0 / (SIMDS * clock_run * 1e9 * float(np.median(kms)) * 1e-3)
                entry.update({"bound": "fp64_valu", "frac": frac, "valu_insts_per_entry_iteration": per, "clock_ghz_this_run": clock_run,
                              "clock_ghz_in_profile": valu[sp["key"]].get("clock_ghz"), "sclk_sysfs_ghz": sclk.ghz(),
                              "bound_note": "VALU issue turns used by the BP kernel (instructions per entry-iteration from profiles/secondary_valu.json x this "
                                            "run's iterations) / turns of 1024 SIMDs at clock_ghz_this_run (device counters of this run) over this run's "
                                            "BP kernel time; at 8 192 syndromes the rest is latency: the 50 iterations of the slowest syndromes, four barriers each"})
            else:
                entry.update({"bound": "fp64_valu", "frac": None, "clock_ghz_this_run": clock_run, "bound_note": "profiles/secondary_valu.json absent, or no clock reading"})
        src = (c3 if sp["method"] == 1 else valu.get(sp["key"], {})) or {}
        if src.get("kernel_sources_sha16"):
            entry["counters_match_this_build"] = src["kernel_sources_sha16"] == kernel_sources_sha16()
        out.append(entry)
        eng.close()
    return out


def schedule_configs(dev, steps):
    """schedule = serial_relative (every syndrome its own bit order, re-sorted with std::sort at the top of every
    iteration) on the two codes it is used on, B = 65 536, through `bp_relative_lds_kernel` (a syndrome's whole decode in LDS; DESIGN.md
    section 6).  The bound is instruction issue at LDS-bound occupancy; what is reported is throughput and a BIT-EXACT sample against
    the checker (decisions, iterations, flags, log-ratio bits, the order left behind)."""
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    import oracle  # checker only

    out = []
    specs = [
        dict(key="f1_rel_surface", name="serial_relative: rotated surface code d=21 X checks (220 x 441), minimum_sum alpha=0.625, max_iter=30, batch=65536, BSC p=0.05",
             h=codes.rotated_surface_code_x(21), p=0.05, max_iter=30, method=1, alpha=0.625),
        dict(key="f1_rel_bb144", name="serial_relative: BB [[144,12,12]] hx (72 x 144), product_sum max_iter=50, batch=65536, BSC p=0.05",
             h=codes.bivariate_bicycle_hx(), p=0.05, max_iter=50, method=0, alpha=1.0),
        # state beyond LDS (61 KB a syndrome + 93 KB of tables): messages, posteriors, priors and per-entry records in global memory, six wavefronts
        # per compute unit (bp_relative_lds_kernel<..., EXT = 1>, round 6; the per-lane kernel that ran before: 7.0 k syndromes/s, profiles/r6_serial_relative_hgp1600.txt)
        dict(key="f1_rel_hgp1600", name="serial_relative: hypergraph-product [[1600,64]] hx (768 x 1600), minimum_sum alpha=0.625, max_iter=30, batch=65536, BSC p=0.02",
             h=codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5)), p=0.02, max_iter=30, method=1, alpha=0.625),
    ]
    for sp in specs:
        B = sp.get("batch", 65536)
        h, p = sp["h"], sp["p"]
        n = h.shape[1]
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), sp["max_iter"], sp["method"], sp["alpha"], device=dev.index or 0)
        eng.set_schedule("serial_relative")
        s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
        res = eng.decode_batch(s)  # warm-up; every row starts from the handle's order, which a call leaves as its last row left it:
        step_ms, kms = [], []      # the timed calls start from that (a permutation either way)
        clk0 = eng.clock_probe()
        for _ in range(max(2, min(steps, 3))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = eng.decode_batch(s)
            torch.cuda.synchronize()
            step_ms.append((time.perf_counter() - t0) * 1e3)
            kms.append(eng.last_kernel_ms())
        clock_run = HipBpEngine.clock_ghz(clk0, eng.clock_probe())
        eng.close()
        ms = float(np.median(step_ms))
        # parity: the first rows and the last one on a fresh handle against the checker (a batch's rows all start from the same order)
        rows = np.r_[0:24, B - 1]
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), sp["max_iter"], sp["method"], sp["alpha"], device=dev.index or 0)
        eng.set_schedule("serial_relative")
        s_host = s[torch.from_numpy(rows).to(dev)].cpu().numpy()
        got = eng.decode_batch(s_host)
        order_after = eng.schedule_order()
        eng.close()
        orc = oracle.BpOracle(h, error_rate=p, max_iter=sp["max_iter"], bp_method=sp["method"], ms_scaling_factor=sp["alpha"])
        want = orc.decode_serial_relative_batch(s_host, fresh=True)
        ok = bool(np.array_equal(got[0], want[0]) and oracle.bits_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
                  and np.array_equal(np.asarray(got[3], bool), want[3]) and np.array_equal(order_after, want[4]))
        it = res[2].cpu().numpy() if hasattr(res[2], "cpu") else np.asarray(res[2])
        cv = res[3].cpu().numpy() if hasattr(res[3], "cpu") else np.asarray(res[3])
        out.append({"config": sp["name"], "key": sp["key"], "value": B / ms * 1e3, "unit": "syndromes/s", "ms": ms, "ms_steps": [round(v, 3) for v in step_ms],
                    "bp_kernel_ms": float(np.median(kms)), "mean_iterations": float(np.asarray(it, np.float64).mean()),
                    "bp_converged_fraction": float(np.asarray(cv, np.float64).mean()), "parity_vs_oracle": ok, "parity": "bit-exact, 25 rows, order left behind included",
                    "bound": "valu_issue", "frac": None, "clock_ghz_this_run": clock_run})
        # the bound, measured: vector instructions per syndrome-iteration of this kernel (committed PMC profile) x this run's syndrome-iterations x 4
        # cycles, against the issue turns of 1024 SIMDs at THIS run's clock over THIS run's kernel time -- as for configs[2] (secondary_configs)
        try:
            with open(os.path.join(ROOT, "profiles", "f1_rel_valu.json")) as f:
                prof = json.load(f)
            per = prof[sp["key"]]
            k_ms = float(np.median(kms))
            if clock_run and k_ms > 0:
                synd_iters = float(np.asarray(it, np.float64).sum())
                turns = SIMDS * clock_run * 1e9 * k_ms * 1e-3
                out[-1].update({"frac": per["valu_wave_insts_per_syndrome_iteration"] * synd_iters * 4.0 / turns,
                                "salu_issue_frac": per["salu_wave_insts_per_syndrome_iteration"] * synd_iters * 4.0 / turns,  # (one scalar instruction per SIMD and 4-cycle turn, as for configs[2])
                                "valu_busy_frac_in_profile": per["valu_busy_frac_in_profile"], "insts_per_syndrome_iteration": {k: round(v, 1) for k, v in per.items() if k.endswith("_iteration")},
                                "counters_match_this_build": prof.get("kernel_sources_sha16") == kernel_sources_sha16(),
                                "bound_note": "frac = VALU wave-instructions (profiles/f1_rel_valu.json) x 4 cycles / (1024 SIMDs x in-run clock x kernel time); "
                                              "the kernel's wavefronts sit at LDS-bound occupancy (11 per compute unit on the surface code) and two thirds of its vector work is the re-enacted std::sort"})
        except Exception as exc:
            out[-1]["bound_note"] = f"profiles/f1_rel_valu.json: {exc!r}"[:200]
    return out


def serial_headline_config(dev, steps, code="ldpc36"):
    """SURVEY.md section 8 (f1), codes beyond LDS: schedule = serial (bp.hpp:451-545) at the early-exit point p = 0.05, product_sum, 50
    iterations, B = 65 536, n = 10 000 -- on configs[1]'s (3,6)-regular code (`f1_serial_c2`: bp_serial_stream_kernel, the form built around
    the (6,3) record) and, since round 6, on a (4,8)-regular code (rows of 8, columns of 4) and the irregular code of `f3_irregular` (rows of
    3 .. 16, columns of 2 .. 8), both on the item form (bp_serial_var_kernel.h); all decoded in passes with what they leave finishing a
    workgroup per syndrome.  `frac` is priced in SURVEY.md 8(d)'s bytes (4 E x 8 per iteration) like every other entry; `frac_moved` in the
    bytes the serial schedule itself moves per lane-iteration -- every entry is read once by each OTHER bit of its row and written once: 8 x the
    sum of the squared row weights = 6 E x 8 on the (6,3) code, 8 E x 8 on the (4,8) code, 9.75 E x 8 on the irregular one.  Parity: a sample of
    rows bit-exact against the checker (LLR bits included)."""
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    import oracle  # checker only

    h, what, key = {"ldpc36": (lambda: codes.regular_ldpc_code(10000, 3, 6, seed=1), "(3,6)-regular LDPC n=10000", "f1_serial_c2"),
                    "ldpc48": (lambda: codes.regular_ldpc_code(10000, 4, 8, seed=1), "(4,8)-regular LDPC n=10000 (rows of 8, columns of 4)", "f1_serial_ldpc48"),
                    "irregular": (lambda: codes.irregular_ldpc_code(10000, 5000, seed=1), "irregular LDPC n=10000 m=5000 E=40000 (rows 3..16, columns 2..8)", "f1_serial_irregular")}[code]
    h = h()
    m, n = h.shape
    p, B, max_iter = 0.05, 65536, 50
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 0, 1.0, device=dev.index or 0)
    eng.set_schedule("serial")
    s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
    out = eng.decode_batch(s)
    step_ms, kms = [], []
    c0 = eng.clock_probe()
    for _ in range(max(2, min(steps, 3))):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_batch(s, out=out)
        torch.cuda.synchronize()
        step_ms.append((time.perf_counter() - t0) * 1e3)
        kms.append(eng.last_kernel_ms())
    c1 = eng.clock_probe()
    try:  # the box's copy rate for tiles of this size (ldpc_hip_bp_copy_probe), right after the timed decodes
        copy_gbps = float(np.median([eng.copy_probe(1024, h.nnz, passes=2)[1] for _ in range(2)]))
    except Exception:
        copy_gbps = None
    ms = float(np.median(step_ms))
    it = out[2].cpu().numpy()
    cv = out[3].cpu().numpy()
    rows = np.r_[0:160, B - 32:B] if code == "ldpc36" else np.r_[0:40, B - 8:B]
    idx = torch.from_numpy(rows).to(dev)
    s_host = s[idx].cpu().numpy()
    want = oracle.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=0).decode_serial_batch(s_host, None)
    ok = bool(np.array_equal(out[0][idx].cpu().numpy(), want[0]) and oracle.bits_equal(out[1][idx].cpu().numpy(), want[1])
              and np.array_equal(it[rows], want[2]) and np.array_equal(cv[rows].astype(bool), want[3]))
    eng.close()
    alg = algorithmic_bytes(it, m, n, h.nnz)
    row_deg = np.diff(h.indptr).astype(np.float64)
    sq = float(np.sum(row_deg * row_deg))
    moved = float(np.sum(it.astype(np.float64)) * sq * 8.0 + B * (m + 9.0 * n + 5.0))
    return {"config": f"serial schedule (fixed order): {what}, product_sum max_iter=50, batch=65536, BSC p=0.05", "key": key,
            "value": B / ms * 1e3, "unit": "syndromes/s", "ms": ms, "ms_steps": [round(v, 3) for v in step_ms], "bp_kernel_ms": float(np.median(kms)),
            "mean_iterations": float(it.mean()), "bp_converged_fraction": float(cv.astype(np.float64).mean()), "parity_vs_oracle": ok,
            "parity": f"bit-exact, {len(rows)} rows (decisions, iterations, flags, log-ratio bits)", "bound": "hbm",
            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "frac_moved": moved / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "clock_ghz_this_run": HipBpEngine.clock_ghz(c0, c1),
            "moved_segments_per_edge_iteration": sq / h.nnz, "copy_GBps_this_run": copy_gbps,
            "frac_ceiling_at_copy_rate": (copy_gbps / HBM_PEAK_GBPS * 4.0 * h.nnz / sq) if copy_gbps else None,
            "bound_note": "frac: SURVEY 8(d) bytes (4 E x 8 per iteration) over the step; frac_moved: the bytes the serial schedule itself moves per lane-iteration "
                          f"(sum of squared row weights x 8 = {sq / h.nnz:.2f} E x 8 here: with exact in-order products an entry must be read by every other bit of its row, "
                          "DESIGN.md section 4) -- so in the 4 E accounting the schedule cannot exceed frac_ceiling_at_copy_rate = this run's bare copy rate / 8 TB/s x 4 E / sum d^2; counters of the (6,3) form's first pass in profiles/r5_serial_stream_summary.txt (traffic 5.1 TB/s = the chip's copy rate)"}


def irregular_config(dev, steps):
    """matrices that are not regular.  The irregular LDPC code of tools/bench_configs.py (n = 10 000, m = 5 000,
    E = 40 000; rows of 3 .. 16 entries, columns of 2 .. 8), product_sum, 50 iterations, B = 32 768 at p = 0.12 (nothing converges: every
    syndrome runs all 50 iterations) -- since round 5 on the per-pass kernels from the first iteration (csrc/host_stream.h: per_pass_first).
    `frac` = SURVEY.md 8(d)'s bytes over the step over 8 TB/s; the path's second bound is FP64 issue (VALU busy 0.80 / 0.96 in the
    check / bit kernel: profiles/r5_irregular_per_pass_pmc_summary.txt).  Parity: rows bit-exact against the checker, LLR bits included."""
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    import oracle  # checker only

    h = codes.irregular_ldpc_code(10000, 5000, seed=1)
    m, n = h.shape
    p, B, max_iter = 0.12, 32768, 50
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 0, 1.0, device=dev.index or 0)
    s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
    out = eng.decode_batch(s)
    step_ms, kms = [], []
    for _ in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.decode_batch(s, out=out)
        torch.cuda.synchronize()
        step_ms.append((time.perf_counter() - t0) * 1e3)
        kms.append(eng.last_kernel_ms())
    ms = float(np.median(step_ms))
    it = out[2].cpu().numpy()
    cv = out[3].cpu().numpy()
    rows = np.r_[0:24, B - 8:B]
    idx = torch.from_numpy(rows).to(dev)
    want = oracle.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=0).decode_batch(s[idx].cpu().numpy())
    ok = bool(np.array_equal(out[0][idx].cpu().numpy(), want[0]) and oracle.bits_equal(out[1][idx].cpu().numpy(), want[1])
              and np.array_equal(it[rows], want[2]) and np.array_equal(cv[rows].astype(bool), want[3].astype(bool)))
    eng.close()
    alg = algorithmic_bytes(it, m, n, h.nnz)
    return {"config": "irregular LDPC n=10000 m=5000 E=40000 (rows 3..16, columns 2..8), product_sum max_iter=50, batch=32768, BSC p=0.12", "key": "f3_irregular",
            "value": B / ms * 1e3, "unit": "syndromes/s", "ms": ms, "ms_steps": [round(v, 3) for v in step_ms], "bp_kernel_ms": float(np.median(kms)),
            "mean_iterations": float(it.mean()), "bp_converged_fraction": float(cv.astype(np.float64).mean()), "parity_vs_oracle": ok,
            "parity": "bit-exact, 32 rows (decisions, iterations, flags, log-ratio bits)", "bound": "hbm",
            "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "bound_note": "SURVEY 8(d) bytes over the step; per-pass kernels from the first iteration (traffic 1.00 x algorithmic, VALU busy 0.80 check / 0.96 bit: "
                          "profiles/r5_irregular_per_pass_pmc_summary.txt; the three paths side by side: profiles/r5_irregular_paths.txt)"}


def host_io_leg(h, args, alpha, synd_dev, dec_dev, it_dev, device_value):
    """The drop-in's own I/O path (_bp_decoder.pyx:642-695 is NumPy in, NumPy out): the SAME batch as pageable host arrays through
    `BpDecoder.decode_batch` -- validation, the all-zero-row shortcut, H2D, kernels, D2H, all inside the timed call -- without and
    with the log-ratio output (655 MB resp. 5.9 GB of results at B = 65 536), next to the device-resident rate."""
    from ldpc_amd.bp_decoder import BpDecoder
    s_host = synd_dev.cpu().numpy()
    dec = BpDecoder(h, error_rate=args.p, max_iter=args.max_iter, bp_method=args.bp_method, ms_scaling_factor=alpha, input_vector_type="syndrome")
    B = s_host.shape[0]
    out = {"api": "ldpc_amd.bp_decoder.BpDecoder.decode_batch((B, m) uint8 ndarray, pageable) -> (B, n) ndarray", "batch": B, "unit": "syndromes/s"}
    dec.recycle_log_prob_ratios = True  # (this loop does not keep a call's log-ratio array: the next call may write the same page-locked memory)
    dec.decode_batch(s_host[: min(B, 4096)], want_log_prob_ratios=False)  # module load, handle, first allocations
    for key, want in (("no_llr", False), ("with_llr", True)):
        dec.decode_batch(s_host, want_log_prob_ratios=want)  # warm-up of this shape: pinned staging buffers
        ms = []
        for _ in range(2):
            t0 = time.perf_counter()
            got = dec.decode_batch(s_host, want_log_prob_ratios=want)
            ms.append((time.perf_counter() - t0) * 1e3)
        best = min(ms)
        out[key] = {"value": B / best * 1e3, "ms": best, "ms_calls": [round(v, 2) for v in ms], "vs_device_resident": B / best * 1e3 / device_value}
        if key == "no_llr":  # what came back is what the device-resident decode produced
            ok = bool(np.array_equal(dec.iter_batch, it_dev.cpu().numpy()))
            rows = np.sort(np.random.default_rng(99).choice(B, size=min(B, 2048), replace=False))
            ok = ok and bool(np.array_equal(got[rows], dec_dev.cpu().numpy()[rows]))
            out["equal_to_device_resident_outputs"] = ok
    out["note"] = ("whole call timed on the host clock, best of two after a warm-up; pageable arrays move through pinned double-buffered chunks "
                   "that overlap the kernels (include/ldpc_hip.h: ldpc_hip_bp_decode_batch with host pointers); chunks of <= 16 384 rows take the "
                   "per-pass kernels, and the two-pass compaction does not apply to them")
    return out


def error_line(args, exc, stage: str) -> dict:
    """What rank 0 prints instead of the result line when the run fails: still ONE parseable JSON line, never a bare traceback."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    info = {"ranks": world, "backend": "nccl" if "WORLD_SIZE" in os.environ else None, "launcher": "torch.distributed.run" if "WORLD_SIZE" in os.environ else None,
            "env": {k: os.environ.get(k) for k in (*RANK_ENV_DEFAULTS, "MASTER_ADDR", "MASTER_PORT", "RANK", "LOCAL_RANK", "WORLD_SIZE")}}
    return {"metric": "syndromes_per_sec_batched_bp50_product_sum_ldpc36_n10k", "value": None, "unit": "syndromes/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "error": f"{type(exc).__name__}: {exc}"[:600], "stage": stage,
            "rank": int(os.environ.get("RANK", "0")), "rccl": info}


def early_exit_point(eng, h, dev, B, args, out, alpha, p=0.05):
    """configs[1]'s code and batch at p = 0.05, where BP converges in ~7 iterations: what a decoder below threshold actually runs.
    `frac` = algorithmic bytes with the per-syndrome iterations actually executed / kernel time / 8 TB/s."""
    import torch
    import oracle  # checker only
    m, n, nnz = h.shape[0], h.shape[1], h.nnz
    eng.set_channel(np.full(n, p))
    s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=B, device=dev)
    eng.decode_batch(s, out=out, asynchronous=True)  # warm-up (also steers the repacking heuristic, as a second call would see it)
    torch.cuda.synchronize()
    kms, step_ms = [], []
    for _ in range(max(3, args.steps)):
        t0 = time.perf_counter()
        eng.decode_batch(s, out=out, asynchronous=True)
        kms.append(eng.last_kernel_ms())
        torch.cuda.synchronize()
        step_ms.append((time.perf_counter() - t0) * 1e3)
    ms, k_ms = float(np.median(step_ms)), float(np.median(kms))
    it = out[2].cpu().numpy()
    cv = out[3].cpu().numpy().astype(bool)
    rows = np.sort(np.random.default_rng(2468).choice(B, size=256, replace=False))
    rt = torch.from_numpy(rows).to(dev)
    orc = oracle.BpOracle(h, error_rate=p, max_iter=args.max_iter, bp_method=args.bp_method, ms_scaling_factor=alpha)
    od, ol, oi, oc = orc.decode_batch(s[rt].cpu().numpy())
    ok = bool(np.array_equal(out[0][rt].cpu().numpy(), od) and np.array_equal(it[rows], oi) and np.array_equal(cv[rows], oc)
              and (out[1] is None or oracle.llr_close(out[1][rt].cpu().numpy(), ol, rtol=1e-5)))
    alg = algorithmic_bytes(it, m, n, nnz)
    eng.set_channel(np.full(n, args.p))
    return {"config": f"configs[1] at its early-exit point: (3,6)-regular LDPC n={n}, {args.bp_method} flooding BP, max_iter={args.max_iter}, "
                      f"batch={B}, BSC p={p}", "key": "c2_p050", "value": B / ms * 1e3, "unit": "syndromes/s", "ms": ms,
            "ms_steps": [round(v, 3) for v in step_ms], "bp_kernel_ms": k_ms, "mean_iterations": float(it.mean()),
            "max_iterations": int(it.max()), "bp_converged_fraction": float(cv.mean()), "parity_vs_oracle": ok,
            "bound": "hbm", "achieved": alg / (k_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "bound_unit": "GB/s",
            "frac": alg / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "algorithmic_bytes_per_launch": alg,
            "bound_note": "algorithmic bytes = sum over syndromes of iters_run * 4 * E * 8 + (m + 9 n + 5); a 64-syndrome tile runs until its "
                          "slowest syndrome has converged and moves all 64 lanes' messages until then -- so, steered by the previous decode's "
                          "iteration histogram, the live lanes' message state is compacted into dense tiles once mid-decode (DESIGN.md section 4)"}


def main() -> None:
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.force_launch):
        sys.exit(self_launch(args.gpus))
    apply_rank_env_defaults()  # before `import torch`: the driver's own torch.distributed.run command does not pass through self_launch

    # ONE JSON line on stdout: libraries that greet on stdout (RCCL prints a version banner from its first communicator) are
    # sent to stderr by pointing file descriptor 1 there for the duration of the run; the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    stage = ["start"]
    rank = int(os.environ.get("RANK", "0"))

    def terminated(signum, _frame):  # the launcher stops the surviving ranks when one of them dies: rank 0 still leaves a line
        if rank == 0 and not stage[0] == "done":
            print(json.dumps(error_line(args, RuntimeError(f"signal {signum} from the launcher (another rank failed?)"), stage[0])),
                  file=real_stdout, flush=True)
        os._exit(1)

    if "WORLD_SIZE" in os.environ:
        import signal
        signal.signal(signal.SIGTERM, terminated)
    try:
        run(args, real_stdout, stage)
        stage[0] = "done"
    except SystemExit:
        raise
    except BaseException as exc:  # noqa: BLE001 -- RCCL / HIP failures surface as RuntimeError, DistBackendError, ...
        import traceback
        traceback.print_exc(file=sys.stderr)
        line = json.dumps(error_line(args, exc, stage[0]))
        # rank 0 owns stdout's one line; another rank's failure goes to stderr (rank 0 then reports the launcher's signal)
        print(line, file=real_stdout if rank == 0 else sys.stderr, flush=True)
        sys.exit(1)


def run(args, real_stdout, stage) -> None:
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    grouped = "WORLD_SIZE" in os.environ  # launched by torch.distributed.run (the driver's command, or self_launch)
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if args.dry_ranks:  # does `python bench.py --gpus N` really become N ranks that find each other?  (tests/test_bench_launch.py)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        stage[0] = "init_process_group(gloo)"
        dist.init_process_group("gloo")
        if os.environ.get("BENCH_DRY_FAIL_RANK") == str(rank):  # test hook: what a rank's failure leaves on stdout
            raise RuntimeError("injected failure (BENCH_DRY_FAIL_RANK)")
        seen = [None] * world
        dist.all_gather_object(seen, {"rank": rank, "local_rank": local_rank, "pid": os.getpid(),
                                      "env": {k: os.environ.get(k) for k in RANK_ENV_DEFAULTS}})
        if rank == 0:
            print(json.dumps({"dry_ranks": seen, "world": dist.get_world_size(), "backend": dist.get_backend()}), file=real_stdout, flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    coll = torch.device("cpu") if args.share_gpu else dev  # where the tensors of a collective live (gloo: host)
    if grouped:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            stage[0] = "init_process_group(gloo)"
            dist.init_process_group("gloo")
        else:
            stage[0] = "init_process_group(nccl)"
            dist.init_process_group("nccl", device_id=dev)
    stage[0] = "engine"

    from ldpc_amd.codes import regular_ldpc_code
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd.sharding import gather_rows

    n = args.n
    h = regular_ldpc_code(n, 3, 6, seed=1)
    m, nnz = h.shape[0], h.nnz
    B = args.batch_per_gpu if args.batch_per_gpu > 0 else (65536 if world == 1 else 131072)
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

    kernel_ms, phase_ms, gather_ms = [], [], []
    gathered = {}

    def step(record: bool):
        eng.decode_batch(synd, want_llr=llr is not None, out=out, asynchronous=True)
        if record:
            kernel_ms.append(eng.last_kernel_ms())  # HIP events around the BP kernels on the launch stream
            phase_ms.append(eng.last_phase_ms())
        if grouped:  # the only collective: gather decoded rows (bit-packed on the device first) + flags onto rank 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            gathered["dec_b8"] = gather_rows(eng.pack_b8(dec).to(coll), total, 0)
            gathered["conv"] = gather_rows(cv.to(coll), total, 0)
            gathered["iters"] = gather_rows(it.to(coll), total, 0)
            e1.record()
            if record:
                gather_ms.append((e0, e1))

    stage[0] = "warmup (first decode + first gather)"
    for _ in range(args.warmup):
        step(False)
    if grouped and args.warmup == 0:
        # RCCL sets up its peer-to-peer connections at the first gather: keep that out of the timed region
        gather_rows(torch.zeros((8, 8), dtype=torch.uint8, device=coll), 8 * world, 0)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    stage[0] = "timed steps"
    probe0 = eng.clock_probe()  # (waits for the stream: outside the timed region)
    sclk = SclkSampler(local_rank)
    sclk.__enter__()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(True)
    torch.cuda.synchronize()
    if grouped:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    sclk.__exit__()
    clock_run = eng.clock_ghz(probe0, eng.clock_probe())  # shader clock of the persistent kernel's workgroups over the timed steps
    # box normaliser (outside the timed steps): a bare copy of the same message segments between the handle's two message arrays --
    # what THIS box gives, right after the timed steps (same thermal state), to any kernel that moves these bytes once
    copy_gbps = None
    try:
        copy_runs = [eng.copy_probe(min((B + 63) // 64, 1024), eng.nnz, passes=2)[1] for _ in range(3)]
        copy_gbps = float(np.median(copy_runs))
    except Exception as exc:  # an older library build (LDPC_HIP_LIB A/B runs) has no probe
        print(f"[bench] copy probe unavailable: {exc}", file=sys.stderr)
    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    stage[0] = "parity + report"
    iters = it.cpu().numpy()
    conv = cv.cpu().numpy().astype(bool)
    k_ms = float(np.mean(kernel_ms))
    g_ms = float(np.mean([a.elapsed_time(b) for a, b in gather_ms])) if gather_ms else 0.0

    # ---- every rank: parity of a sample of ITS shard against the CPU checker; per-rank timings to rank 0 -------------
    rank_ok = 1
    rank_rows = 0
    if grouped and args.rank_parity > 0:
        from oracle import cpu_bench, llr_close  # checker only
        rank_rows = min(B, args.rank_parity)
        rows = np.sort(np.random.default_rng(777 + rank).choice(B, size=rank_rows, replace=False))
        rows_t = torch.from_numpy(rows).to(dev)
        _, (cd, cl, ci, cc) = cpu_bench.run(h, args.p, args.max_iter, args.bp_method, alpha, synd[rows_t].cpu().numpy(),
                                            cores=max(1, min(32, physical_cores() // world)))
        ok = np.array_equal(dec[rows_t].cpu().numpy(), cd) and np.array_equal(iters[rows], ci) and np.array_equal(conv[rows], cc)
        if llr is not None:
            ok = ok and llr_close(llr[rows_t].cpu().numpy(), cl, rtol=1e-5)
        if rank == 0 and ok:  # what arrived through the gather is what this rank produced
            own = eng.unpack_b8(gathered["dec_b8"][:B].contiguous().to(dev), n)
            ok = bool(torch.equal(own, dec)) and bool(torch.equal(gathered["conv"][:B].to(dev), cv)) and bool(torch.equal(gathered["iters"][:B].to(dev), it))
        rank_ok = int(bool(ok))
    per_rank = None
    if grouped:
        mine = torch.tensor([k_ms, g_ms, float(rank_ok), float(iters.mean())], dtype=torch.float64, device=coll)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(v) for v in t_.cpu()] for t_ in allr]

    if rank == 0:
        alg = algorithmic_bytes(iters, m, n, nnz)
        pers_ms = float(np.mean([p_[0] for p_ in phase_ms]))
        spread_ms = float(np.mean([p_[1] for p_ in phase_ms]))
        achieved = alg / (k_ms * 1e-3) / 1e9
        # HBM bytes per launch and the VALU / clock picture from the committed PMC run of this same workload
        # (profiles/hbm_traffic.json, profiles/valu_clock.json: tools/profile_bench.sh + tools/prof_parse.py); null when
        # absent or measured on another workload
        traffic, second = None, None
        same = world == 1 and method_id == 0

        def committed(name):
            try:
                t_ = json.load(open(os.path.join(ROOT, "profiles", name)))
                if t_.get("batch_per_gpu") == B and t_.get("p") == args.p and t_.get("max_iter") == args.max_iter and t_.get("n") == n \
                        and t_.get("math", "libm_exact") == args.math:
                    return t_
            except Exception:
                pass
            return None
        build_tag = kernel_sources_sha16()
        profile_tag = None
        if same:
            tj = committed("hbm_traffic.json")
            traffic = tj["hbm_bytes_per_launch"] if tj else None
            profile_tag = (tj or {}).get("kernel_sources_sha16")
            vj = committed("valu_clock.json")
            if vj and vj.get("valu_insts_per_edge_iteration") and clock_run:
                # VALU issue: instructions per edge-iteration (PMC profile of this workload; belongs to this build when the fingerprints
                # match) x this run's edge-iterations, against the issue turns of 1024 SIMDs at THIS run's clock over THIS run's kernel time
                per = float(vj["valu_insts_per_edge_iteration"])
                wave_insts = float(iters.astype(np.float64).sum()) * nnz * per / 64.0
                turns = SIMDS * clock_run * 1e9 * k_ms * 1e-3
                f64_per = float(vj.get("fp64_arith_insts_per_edge_iteration") or 0.0)
                second = {"kind": "fp64_valu", "issue_frac": wave_insts * 4.0 / turns,
                          # the same count with FP64 arithmetic at 4 cycles and everything else (selects, moves, compares, integer work) at 2: a LOWER
                          # bound of the occupancy; the hardware's own figure of the profiled run is valu_busy_frac_in_profile
                          "issue_frac_fp64_at_4_others_at_2": (wave_insts * (f64_per / per) * 4.0 + wave_insts * (1.0 - f64_per / per) * 2.0) / turns if f64_per else None,
                          "valu_busy_frac_in_profile": vj.get("valu_busy_frac"),
                          "clock_ghz_this_run": clock_run, "sclk_sysfs_ghz": sclk.ghz(), "insts_per_edge_iter": per, "fp64_arith_insts_per_edge_iter": f64_per or None,
                          "issue_frac_in_profile": vj.get("valu_issue_frac"), "clock_ghz_in_profile": vj.get("clock_ghz"),
                          "source": "instructions per edge-iteration: profiles/valu_clock.json (SQ_INSTS_VALU of this workload); clock: shader cycles / "
                                    "constant-rate ticks summed over the persistent kernel's workgroups during the timed steps (ldpc_hip_bp_clock_probe); "
                                    "kernel time: HIP events of this run"}
            elif clock_run:
                second = {"kind": "fp64_valu", "issue_frac": None, "clock_ghz_this_run": clock_run, "sclk_sysfs_ghz": sclk.ghz()}
        res = {
            "metric": "syndromes_per_sec_batched_bp50_product_sum_ldpc36_n10k" if method_id == 0 else "syndromes_per_sec_DIAGNOSTIC_min_sum",
            "value": total * args.steps / elapsed,
            "unit": "syndromes/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": (f"configs[1]: " if world == 1 else f"configs[3] geometry ({B} syndromes per GPU; N = 8 is configs[3] itself): ")
                            + f"(3,6)-regular LDPC n={n} (m={m}, E={nnz}), {args.bp_method} flooding BP, "
                              f"max_iter={args.max_iter}, batch={B} syndromes per GPU, BSC p={args.p}, code seed 1, error seed 7",
                "batch_per_gpu": B, "global_batch": total, "p": args.p, "max_iter": args.max_iter,
                "outputs": "decoding u8, log_prob_ratios f64, iterations i32, converge u8" if llr is not None else "no LLR",
                "parallelism": f"batch-sharded x{world}, one gather of bit-packed decoded rows + flags onto rank 0" if grouped else "single GPU, no process group",
                "device_math": args.math, "mean_iterations": float(iters.mean()), "converged_fraction": float(conv.mean()),
            },
            "roofline": {
                "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                "kernel": "bp_decode_kernel (persistent, one workgroup per 64-syndrome tile) + bp_spread_* per-pass launches for the last tiles",
                "kernel_ms": k_ms, "kernel_ms_persistent": pers_ms, "kernel_ms_per_pass": spread_ms,
                "clock_ghz_this_run": clock_run,
                "copy_GBps_this_run": copy_gbps, "frac_of_copy": (achieved / copy_gbps) if copy_gbps else None,
                "copy_note": "copy_GBps_this_run = bytes read + written per second by ldpc_hip_bp_copy_probe right after the timed steps: one workgroup per "
                             "tile copying the tile's 30 000 message segments of 512 bytes to the other message array, non-temporal, no arithmetic "
                             "(median of 3 x 2 passes, events on the launch stream); boxes of the pool differ by ~10 % here, builds do not",
                "algorithmic_bytes_per_launch": alg,
                "second_bound": second,
                "counters_from_build": profile_tag, "this_build": build_tag,
                "counters_match_this_build": (profile_tag == build_tag) if profile_tag else None,
                "note": "one decode = the whole batch of this rank; algorithmic bytes = sum over syndromes of iters_run*4*E*8 + (m+n+8n+5); "
                        "kernel_ms = HIP events on the launch stream around all BP kernels of the decode (the persistent kernel "
                        "hands its last <= 256 tiles to per-pass launches: compare kernel_ms_persistent with rocprofv3's "
                        "bp_decode_kernel average and kernel_ms_per_pass with the sum of the bp_spread_* kernels); traffic = PMC "
                        "counters of the same kernels from the committed profile run (profiles/); second_bound = that profile's instruction count per "
                        "edge-iteration with THIS run's clock (clock_ghz_this_run: device cycle / tick counters), iterations and kernel time",
            },
        }
        if grouped:
            res["rccl"] = {"ranks": dist.get_world_size(), "backend": dist.get_backend(),
                           "version": ".".join(str(v) for v in torch.cuda.nccl.version()),
                           "devices": [torch.cuda.get_device_name(local_rank)], "launcher": "torch.distributed.run"}
            if args.share_gpu:
                res["rccl"]["note"] = "--share-gpu diagnostic: every rank on cuda:0, gloo group, collectives on host tensors; the rate means nothing"

            res["per_rank"] = {"kernel_ms": [r[0] for r in per_rank], "gather_ms": [r[1] for r in per_rank],
                               "mean_iterations": [r[3] for r in per_rank],
                               "parity_ok": [bool(r[2]) for r in per_rank], "parity_rows_per_rank": rank_rows,
                               "parity_all_ranks": bool(all(r[2] for r in per_rank)),
                               "parity_note": "every rank: decisions, iterations, converge flags bit-exact and LLR <= 1e-5 vs the CPU checker on "
                                              "random rows of its own shard; rank 0 also checks that its rows came back unchanged through the gather"}
            res["gather"] = {"ms": g_ms, "bytes_to_rank0": (world - 1) * B * ((n + 7) // 8 + 5),
                             "rows_on_rank0": int(gathered["dec_b8"].shape[0]) if gathered.get("dec_b8") is not None else 0}
            if not res["per_rank"]["parity_all_ranks"]:
                res["parity_failed"] = True
        # ---- CPU baseline + parity gate on a bounded sample of THIS batch (rank 0, N = 1 only) ----
        if world == 1 and args.cpu_sample != 0:
            from oracle import llr_close  # checker only
            from ldpc_amd.noise_models import generate_bsc_batch
            sample = args.cpu_sample if args.cpu_sample > 0 else min(B, 1024)
            # the device shot generator against its host twin on the first rows ...
            head = min(B, 64)
            err = generate_bsc_batch(n, args.p, 7, 0, head)
            s_head = np.asarray((h.astype(np.int32) @ err.T.astype(np.int32)).T % 2, dtype=np.uint8)
            assert np.array_equal(s_head, synd[:head].cpu().numpy()), "device shot generator differs from its host twin"
            # ... and the decoder against the CPU reference on a random subset of the timed batch
            rows = np.sort(np.random.default_rng(12345).choice(B, size=sample, replace=False))
            rows_t = torch.from_numpy(rows).to(dev)
            cpu, (cd, cl, ci, cc) = cpu_baseline(h, args.p, args.max_iter, args.bp_method, alpha, synd[rows_t].cpu().numpy())
            gd = dec[rows_t].cpu().numpy()
            ok = bool(np.array_equal(gd, cd) and np.array_equal(iters[rows], ci) and np.array_equal(conv[rows], cc))
            if llr is not None:
                ok = ok and llr_close(llr[rows_t].cpu().numpy(), cl, rtol=1e-5)
            cpu["parity_vs_gpu"] = {"syndromes": int(sample), "subset": "random rows of the timed batch (seed 12345)",
                                    "hard_decisions_iters_converge_exact_llr_1e-5": ok}
            res["cpu_baseline"] = cpu
            if not ok:
                res["parity_failed"] = True
        else:
            res["cpu_baseline"] = "N = 1 only" if world > 1 else None  # (None: --cpu-sample 0 asked for no CPU leg)
            if world > 1:
                res["roofline"]["traffic_note"] = "PMC traffic is collected at N = 1 only (profiles/hbm_traffic.json)"
        if world == 1 and not grouped and args.host_io and method_id == 0:
            stage[0] = "host_io"
            try:
                res["host_io"] = host_io_leg(h, args, alpha, synd, dec, it, res["value"])
                if not res["host_io"].get("equal_to_device_resident_outputs", False):
                    res["parity_failed"] = True
            except Exception as exc:  # the headline line must not be lost to this leg
                res["host_io"] = {"error": repr(exc)[:300]}
        if world == 1 and args.secondary and method_id == 0:
            early = None
            try:  # the headline code at its early-exit operating point (p = 0.05), same engine, same buffers
                early = early_exit_point(eng, h, dev, B, args, (dec, llr, it, cv), alpha)
            except Exception as exc:
                early = {"key": "c2_p050", "error": repr(exc)[:300]}
            eng.close()
            del synd, dec, llr, out
            torch.cuda.empty_cache()
            try:
                res["secondary"] = [early] + secondary_configs(dev, max(5, args.steps))  # (millisecond calls: a median of at least five)
                res["secondary"] += schedule_configs(dev, args.steps)
                res["secondary"].append(serial_headline_config(dev, args.steps))
                res["secondary"].append(serial_headline_config(dev, 2, "ldpc48"))
                res["secondary"].append(serial_headline_config(dev, 2, "irregular"))
                res["secondary"].append(irregular_config(dev, args.steps))
                if not all(e.get("parity_vs_oracle", False) for e in res["secondary"]):
                    res["parity_failed"] = True
            except Exception as exc:  # the headline line must not be lost to a secondary config
                res["secondary"] = {"error": repr(exc)[:300]}
        print(json.dumps(res), file=real_stdout, flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
