#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs (they are parity-test cases, not bench.py lines; this records them).

  c3: rotated surface code d=21 X checks (220x441, nnz 840), minimum_sum alpha=0.625, 30 iterations, B = 262144
  c5: BB [[144,12,12]] hx (72x144, nnz 432), product_sum 50 iterations + OSD-0, B = 8192 (and a large batch)
Inputs are generated on the device; timing = host clock around K decode calls bracketed by device synchronisation.
"""
import argparse
import json
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(name, h, p, max_iter, method, alpha, batch, osd0, steps=3, math="libm_exact", schedule="parallel", osd=None, random_serial=None, switches=()):
    import torch
    from ldpc_amd.engine import HipBpEngine
    m, n = h.shape
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, method, alpha)
    eng.set_math(math)
    eng.set_schedule(schedule)
    if random_serial is not None:
        eng.set_random_serial(True, random_serial)
    for key, val in switches:
        eng.set_debug_switch(key, val)
    if os.environ.get("LDPC_BENCH_HANDOFF"):  # (A/B of the hand-off threshold: a value >= the number of tiles sends the whole batch to the per-pass kernels)
        eng.set_handoff(int(os.environ["LDPC_BENCH_HANDOFF"]))
    s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=batch, device="cuda:0")
    if osd is not None:  # (osd_method, osd_order): 2 = OSD_E, 3 = OSD_CS
        eng.set_osd(*osd)
    out = eng.decode_batch(s, osd0=osd0, osd=osd is not None)  # warm-up + result for statistics
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.decode_batch(s, out=out, osd0=osd0, osd=osd is not None, asynchronous=True)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    it = out[2].cpu().numpy()
    cv = out[3].cpu().numpy().astype(bool)
    alg = float(np.sum(it.astype(np.float64) * 4.0 * h.nnz * 8.0 + (m + n + 8.0 * n + 5.0)))
    print(json.dumps({"config": name, "batch": batch, "syndromes_per_s": batch / ms * 1e3, "ms_per_decode": ms,
                      "bp_kernel_ms": eng.last_kernel_ms(), "mean_iterations": float(it.mean()),
                      "bp_converged": float(cv.mean()), "algorithmic_GBps": alg / ms / 1e6, "math": math, "schedule": schedule,
                      "hbm_frac_of_8TBps_kernel_time": alg / (eng.last_kernel_ms() * 1e-3) / 8e12 if eng.last_kernel_ms() > 0 else None}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["c3", "c5"])
    args = ap.parse_args()
    from ldpc_amd import codes
    if "c3" in args.which:
        h = codes.rotated_surface_code_x(21)
        run("c3 surface d=21 min_sum 30 it p=0.05", h, 0.05, 30, 1, 0.625, 262144, False)
        run("c3 surface d=21 min_sum 30 it p=0.01", h, 0.01, 30, 1, 0.625, 262144, False)
    if "c3p05" in args.which:  # config 3 at its primary operating point alone (tools/profile_c3.sh counts its instructions)
        run("c3 surface d=21 min_sum 30 it p=0.05", codes.rotated_surface_code_x(21), 0.05, 30, 1, 0.625, 262144, False)
    if "irregular" in args.which:
        # an irregular LDPC code (rows of 3 ... 16 entries, columns of 2 / 3 / 6 / 8): the streamed kernels' generic-degree variants
        # (bp_decode_kernel<., ., 16, 8, 0>: one row in registers, no register double buffer, 8-wavefront workgroups)
        h = codes.irregular_ldpc_code(10000, 5000, seed=1)
        for p_, meth, alpha in ((0.03, 0, 1.0), (0.06, 0, 1.0), (0.06, 1, 0.75), (0.12, 0, 1.0), (0.12, 1, 0.75)):  # (0.12: nothing converges -- every tile runs all 50 iterations)
            run(f"irregular LDPC n=10000 m=5000 E=40000 (rows 3..16, columns 2..8), {'product_sum' if meth == 0 else 'minimum_sum'} 50 it p={p_}",
                h, p_, 50, meth, alpha, 32768, False)
    if "irregular_fast" in args.which:  # the same code with the fast arithmetic (~1 ulp), product-sum
        h = codes.irregular_ldpc_code(10000, 5000, seed=1)
        for p_ in (0.06, 0.12):
            run(f"irregular LDPC n=10000 m=5000 E=40000 (rows 3..16, columns 2..8), product_sum fast math 50 it p={p_}", h, p_, 50, 0, 1.0, 32768, False, math="fast")
    if "irregular8" in args.which:  # irregular codes whose rows fit the 8-entry register variants: <8,4,0> (columns 2..4) and <8,8,0> (columns 2..8)
        for tag, cw in (("columns 2..4", ((2, 0.3), (3, 0.5), (4, 0.2))), ("columns 2..8", ((2, 0.35), (3, 0.5), (8, 0.15)))):
            h = codes.irregular_ldpc_code(10000, 5000, seed=2, row_weights=(3, 4, 5, 6, 7, 8), col_weights=cw)
            for p_, meth, alpha in ((0.05, 0, 1.0), (0.12, 0, 1.0), (0.12, 1, 0.75)):
                run(f"irregular LDPC n=10000 m=5000 E={h.nnz} (rows 3..8, {tag}), {'product_sum' if meth == 0 else 'minimum_sum'} 50 it p={p_}", h, p_, 50, meth, alpha, 32768, False)
    if "irregular_ps12" in args.which:  # the all-50-iterations point alone (counter passes)
        run("irregular LDPC n=10000 m=5000 E=40000 (rows 3..16, columns 2..8), product_sum 50 it p=0.12", codes.irregular_ldpc_code(10000, 5000, seed=1), 0.12, 50, 0, 1.0, 32768, False)
    if "serial" in args.which:
        serial()
    if "serial_big" in args.which:
        serial_big()
    if "stateful" in args.which:
        stateful()
    if "soft" in args.which:
        soft()
    if "hgp" in args.which:
        hgp()
    if "hgp1600" in args.which:
        hgp1600()
    if "osdw" in args.which:
        h = codes.bivariate_bicycle_hx()
        run("c5 BB144 product_sum 50 it + OSD_CS order 10 p=0.05", h, 0.05, 50, 0, 1.0, 8192, False, osd=(3, 10))
        run("c5 BB144 product_sum 50 it + OSD_CS order 10 p=0.05, B=262144", h, 0.05, 50, 0, 1.0, 262144, False, osd=(3, 10))
        run("c5 BB144 product_sum 50 it + OSD_CS order 60 p=0.05, B=262144", h, 0.05, 50, 0, 1.0, 262144, False, osd=(3, 60))
        run("c5 BB144 product_sum 50 it + OSD_E order 10 p=0.05, B=262144", h, 0.05, 50, 0, 1.0, 262144, False, osd=(2, 10))
        run("c5 BB144 product_sum 50 it + OSD-0 p=0.05, B=262144", h, 0.05, 50, 0, 1.0, 262144, True)
    if "c5bp" in args.which:  # the BP stage of config 5 alone, one workload (tools/profile_secondary.sh counts its VALU instructions)
        run("c5 BB144 product_sum 50 it (BP only) p=0.05", codes.bivariate_bicycle_hx(), 0.05, 50, 0, 1.0, 8192, False)
    if "c5" in args.which:
        h = codes.bivariate_bicycle_hx()
        run("c5 BB144 product_sum 50 it + OSD-0 p=0.05", h, 0.05, 50, 0, 1.0, 8192, True)
        run("c5 BB144 product_sum 50 it (BP only) p=0.05", h, 0.05, 50, 0, 1.0, 8192, False)
        run("c5 BB144 product_sum 50 it + OSD-0 p=0.05, B=262144", h, 0.05, 50, 0, 1.0, 262144, True)
        run("c5 BB144 product_sum 50 it + OSD-0 p=0.05, B=262144 fast math", h, 0.05, 50, 0, 1.0, 262144, True, math="fast")


def hgp():
    """The [[400,16,6]] hypergraph-product code of the reference's test_qcodes.py (matrix taken from the committed fixture):
    min-sum 0.625, 30 iterations, OSD_0 / OSD_CS 10 at p = 0.02."""
    import scipy.sparse as sp
    z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "qcodes_400_16_6_ms_par_osd0.npz"))
    h = sp.csr_matrix((np.ones(len(z["col_idx"]), np.uint8), z["col_idx"], z["row_ptr"]), shape=(int(z["m"]), int(z["n"])))
    run("hgp [[400,16,6]] min_sum 30 it (BP only) p=0.02", h, 0.02, 30, 1, 0.625, 65536, False)
    run("hgp [[400,16,6]] min_sum 30 it + OSD-0 p=0.02", h, 0.02, 30, 1, 0.625, 65536, True)
    run("hgp [[400,16,6]] min_sum 30 it + OSD_CS order 10 p=0.02", h, 0.02, 30, 1, 0.625, 65536, False, osd=(3, 10))


def hgp1600():
    """A [[1600, 64]] hypergraph product of a random (3,4)-regular 24 x 32 code with itself: hx is 768 x 1600, [H|s] is 156 KiB
    bit-packed -- beyond LDS, so OSD runs through osd_big_kernel (H in an HBM scratch slot)."""
    import scipy.sparse as sp
    from ldpc_amd import codes
    hx = codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5))
    for p in (0.02, 0.04):
        run(f"hgp [[1600,64]] min_sum 30 it (BP only) p={p}", hx, p, 30, 1, 0.625, 65536, False)
        run(f"hgp [[1600,64]] min_sum 30 it + OSD-0 p={p}", hx, p, 30, 1, 0.625, 65536, True)
        run(f"hgp [[1600,64]] min_sum 30 it + OSD_CS order 10 p={p}", hx, p, 30, 1, 0.625, 65536, False, osd=(3, 10))


def soft():
    """SoftInfoBpDecoder path: analog syndromes (+-2 by the true bit, triangular noise), serial minimum-sum."""
    import time
    import torch
    from ldpc_amd import codes
    from ldpc_amd.engine import HipBpEngine
    for name, h, p, max_iter, alpha in (("BB144", codes.bivariate_bicycle_hx(), 0.05, 50, 0.9),
                                        ("surface d=21", codes.rotated_surface_code_x(21), 0.05, 30, 0.625)):
        m, n = h.shape
        for batch in (65536, 262144):
            eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 1, alpha)
            s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=batch, device="cuda:0")
            g = torch.Generator(device="cuda").manual_seed(1)
            soft = (1.0 - 2.0 * s.double()) * 2.0 + 2.5 * (torch.rand(s.shape, generator=g, device="cuda", dtype=torch.float64)
                                                           + torch.rand(s.shape, generator=g, device="cuda", dtype=torch.float64) - 1.0)
            out = eng.soft_info_decode_batch(soft, 2.0, 0.7)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                out = eng.soft_info_decode_batch(soft, 2.0, 0.7)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 3 * 1e3
            print(json.dumps({"config": f"soft-info: {name} serial min_sum {max_iter} it, cutoff 2, sigma 0.7, B={batch}", "batch": batch,
                              "syndromes_per_s": batch / ms * 1e3, "ms_per_decode": ms, "mean_iterations": float(out[2].float().mean()),
                              "bp_converged": float(out[3].float().mean())}), flush=True)


def stateful():
    """The two schedules that keep state in the decoder object (bp.hpp:467-483): serial_relative (every lane re-sorts its own bit
    order before every iteration) and the random serial order (one shuffled order per iteration for the whole call), next to
    the fixed-order serial schedule on the same workload."""
    from ldpc_amd import codes
    for name, h, p, it, method, alpha in (("BB144 product_sum 50 it", codes.bivariate_bicycle_hx(), 0.05, 50, 0, 1.0),
                                          ("surface d=21 min_sum 30 it", codes.rotated_surface_code_x(21), 0.05, 30, 1, 0.625)):
        run(f"serial (fixed order): {name} p={p}", h, p, it, method, alpha, 65536, False, steps=2, schedule="serial")
        run(f"serial, random order per iteration: {name} p={p}", h, p, it, method, alpha, 65536, False, steps=2, schedule="serial", random_serial=1234)
        run(f"serial_relative: {name} p={p}", h, p, it, method, alpha, 65536, False, steps=2, schedule="serial_relative")
        if os.environ.get("LDPC_BENCH_REL_VARIANTS"):  # the on-chip kernel with 64 / 16 lanes per syndrome, and the per-lane kernel
            run(f"serial_relative [bit by bit, REL_LEVELS=0]: {name} p={p}", h, p, it, method, alpha, 65536, False, steps=1, schedule="serial_relative", switches=(("REL_LEVELS", 0),))
            for gs in (16, 0):
                run(f"serial_relative [REL_LDS={gs}]: {name} p={p}", h, p, it, method, alpha, 65536 if gs else 8192, False, steps=1, schedule="serial_relative", switches=(("REL_LDS", gs),))


def serial_big():
    """The serial kernel is one wavefront per 64-syndrome tile: it needs a large batch to fill the chip."""
    from ldpc_amd import codes
    h = codes.bivariate_bicycle_hx()
    for b in (262144, 1048576):
        run(f"serial: BB144 product_sum 50 it p=0.05, B={b}", h, 0.05, 50, 0, 1.0, b, False, schedule="serial")
    h = codes.rotated_surface_code_x(21)
    for b in (262144, 1048576):
        run(f"serial: surface d=21 min_sum 30 it p=0.05, B={b}", h, 0.05, 30, 1, 0.625, b, False, schedule="serial")


def serial():
    from ldpc_amd import codes
    h = codes.bivariate_bicycle_hx()
    run("serial: BB144 product_sum 50 it + OSD-0 p=0.05", h, 0.05, 50, 0, 1.0, 65536, True, schedule="serial")
    run("serial: BB144 min_sum(0.625) 50 it p=0.05", h, 0.05, 50, 1, 0.625, 65536, False, schedule="serial")
    h = codes.rotated_surface_code_x(21)
    run("serial: surface d=21 min_sum 30 it p=0.05", h, 0.05, 30, 1, 0.625, 65536, False, schedule="serial")
    h = codes.regular_ldpc_code(10000, 3, 6, seed=1)
    run("serial: (3,6) n=10000 product_sum 50 it p=0.05", h, 0.05, 50, 0, 1.0, 65536, False, steps=1, schedule="serial")


if __name__ == "__main__":
    main()
