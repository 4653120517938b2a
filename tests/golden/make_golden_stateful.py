#!/usr/bin/env python3
"""Golden fixtures for the schedules that keep state in the decoder object (bp.hpp:467-483): schedule = serial_relative
(the bit order is re-sorted by std::sort at the start of every iteration and stays rearranged) and the random serial
schedule (std::shuffle on a std::mt19937 before every iteration), through the REAL reference (oracle/_ref/libref_bp.so).
Build container only:   make -C oracle ref && python tests/golden/make_golden_stateful.py

Two ways of running the same syndromes are recorded:
  fresh    a NEW decoder object per syndrome  -- what row b of a decode_batch means on the device;
  carried  ONE decoder object, syndromes one after the other -- what a loop of BpDecoder.decode calls does.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import oracle  # noqa: E402
from oracle import csr_arrays  # noqa: E402
from ldpc_amd import codes  # noqa: E402
from make_golden import bsc_syndromes, h_crc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def run(name, h, p, *, schedule, max_iter, bp_method, alpha, rows, random_serial=False, seed=0, varied=False, order0=None):
    h = sp.csr_matrix(h, dtype=np.uint8)
    m, n, rp, ci = csr_arrays(h)
    s = bsc_syndromes(h, 31, p, 0, rows)
    probs = np.full(n, p)
    if varied:
        probs = np.clip(p * np.random.default_rng(9).uniform(0.5, 1.5, n), 1e-3, 0.4)
    kw = dict(schedule=schedule, error_channel=probs, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=alpha,
              random_serial=random_serial, seed=seed, order0=order0)
    extra = {} if order0 is None else {"order0": np.asarray(order0, np.int32)}  # the serial_schedule_order given to the constructor
    fd, fl, fi, fc, fo = oracle.ref_decode_stateful(h, s, fresh=True, **kw)
    cd, cl, ci_, cc, co = oracle.ref_decode_stateful(h, s, fresh=False, **kw)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, name=name, m=m, n=n, h_crc=np.uint32(h_crc(h)), row_ptr=rp, col_idx=ci, channel_probs=probs,
                        max_iter=np.int32(max_iter), bp_method=np.int32(0 if bp_method == "product_sum" else 1), ms_scaling_factor=np.float64(alpha),
                        schedule=np.int32({"serial": 0, "serial_relative": 2}[schedule]), random_serial=np.bool_(random_serial), seed=np.int32(seed),
                        syndromes=np.packbits(s, axis=1),
                        fresh_decoding=np.packbits(fd, axis=1), fresh_llr=fl, fresh_iterations=fi, fresh_converge=fc, fresh_order_last=fo[-1],
                        carried_decoding=np.packbits(cd, axis=1), carried_llr=cl, carried_iterations=ci_, carried_converge=cc,
                        carried_orders=co.astype(np.int16 if n < 32768 else np.int32), **extra)
    differ = int((fd != cd).any(axis=1).sum())
    print(f"{name:34s} {m} x {n} rows={rows} converged={int(fc.sum())} rows where carried != fresh: {differ}  {os.path.getsize(path) / 1024:.1f} KiB")


def run_soft(name, h, soft, p, *, max_iter, alpha, cutoff, sigma, seed):
    """soft_info_decode_serial with random_serial_schedule (bp.hpp:573-577: the order is reshuffled by a NEW
    std::default_random_engine(seed) at the top of every iteration that runs), fresh and carried as above."""
    h = sp.csr_matrix(h, dtype=np.uint8)
    m, n, rp, ci = csr_arrays(h)
    soft = np.ascontiguousarray(soft, np.float64).reshape(-1, m)
    kw = dict(error_rate=p, max_iter=max_iter, ms_scaling_factor=alpha, seed=seed)
    fd, fl, fi, fc, fs, fo = oracle.ref_soft_random(h, soft, cutoff, sigma, fresh=True, **kw)
    cd, cl, ci_, cc, cs, co = oracle.ref_soft_random(h, soft, cutoff, sigma, fresh=False, **kw)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, name=name, m=m, n=n, h_crc=np.uint32(h_crc(h)), row_ptr=rp, col_idx=ci, channel_probs=np.full(n, p),
                        max_iter=np.int32(max_iter), ms_scaling_factor=np.float64(alpha), cutoff=np.float64(cutoff), sigma=np.float64(sigma),
                        seed=np.int32(seed), soft_syndromes=soft,
                        fresh_decoding=np.packbits(fd, axis=1), fresh_llr=fl, fresh_iterations=fi, fresh_converge=fc, fresh_soft_out=fs,
                        fresh_order_last=fo[-1],
                        carried_decoding=np.packbits(cd, axis=1), carried_llr=cl, carried_iterations=ci_, carried_converge=cc, carried_soft_out=cs,
                        carried_orders=co.astype(np.int16 if n < 32768 else np.int32))
    differ = int((fd != cd).any(axis=1).sum())
    print(f"{name:34s} {m} x {n} rows={len(soft)} converged={int(fc.sum())} mean iterations {fi.mean():.2f} rows where carried != fresh: {differ}  "
          f"{os.path.getsize(path) / 1024:.1f} KiB")


def main():
    bb = codes.bivariate_bicycle_hx()
    run("stateful_rel_bb144_ps", bb, 0.06, schedule="serial_relative", max_iter=12, bp_method="product_sum", alpha=1.0, rows=96)
    run("stateful_rel_bb144_ms_varied", bb, 0.06, schedule="serial_relative", max_iter=12, bp_method="minimum_sum", alpha=0.8, rows=70, varied=True)
    run("stateful_rel_ham4_ms", codes.hamming_code(4), 0.08, schedule="serial_relative", max_iter=9, bp_method="minimum_sum", alpha=0.0, rows=40)
    run("stateful_rel_surf7_ps", codes.rotated_surface_code_x(7), 0.06, schedule="serial_relative", max_iter=10, bp_method="product_sum", alpha=1.0, rows=64)
    run("stateful_rel_ldpc600_ms", codes.regular_ldpc_code(600, 3, 6, seed=3), 0.06, schedule="serial_relative", max_iter=8, bp_method="minimum_sum", alpha=0.75,
        rows=48)
    run("stateful_rnd_bb144_ps_s7", bb, 0.06, schedule="serial", max_iter=12, bp_method="product_sum", alpha=1.0, rows=96, random_serial=True, seed=7)
    run("stateful_rnd_surf7_ms_s123", codes.rotated_surface_code_x(7), 0.06, schedule="serial", max_iter=10, bp_method="minimum_sum", alpha=0.625, rows=64,
        random_serial=True, seed=123)
    run("stateful_rnd_ldpc600_ps_s1", codes.regular_ldpc_code(600, 3, 6, seed=3), 0.06, schedule="serial", max_iter=8, bp_method="product_sum", alpha=1.0, rows=48,
        random_serial=True, seed=1)
    # a serial_schedule_order from the caller (bp.hpp:110-111: taken as it is): a permutation, and one with bits twice and bits never
    og = np.random.default_rng(77)
    perm = og.permutation(144).astype(np.int32)
    run("stateful_rel_bb144_ms_startperm", bb, 0.06, schedule="serial_relative", max_iter=10, bp_method="minimum_sum", alpha=0.625, rows=64, order0=perm)
    rep = perm.copy()
    rep[20:44] = rep[60:84]
    run("stateful_rel_bb144_ps_startrepeats", bb, 0.06, schedule="serial_relative", max_iter=10, bp_method="product_sum", alpha=1.0, rows=64, order0=rep)
    s7 = codes.rotated_surface_code_x(7)
    rep7 = og.integers(0, 49, 49).astype(np.int32)
    run("stateful_rel_surf7_ms_startrepeats", s7, 0.06, schedule="serial_relative", max_iter=10, bp_method="minimum_sum", alpha=0.0, rows=48, order0=rep7)
    # the random flag wins over serial_relative (bp.hpp:467-469)
    run("stateful_rnd_over_rel_ham4_s5", codes.hamming_code(4), 0.08, schedule="serial_relative", max_iter=9, bp_method="product_sum", alpha=1.0, rows=40,
        random_serial=True, seed=5)

    # a code whose state does not fit LDS (768 x 1600, the [[1600,64]] hypergraph product of tools/bench_configs.py hgp1600): serial_relative takes the
    # per-lane kernel with its state in HBM there
    hx = codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5))
    run("stateful_rel_hgp1600_ms", hx, 0.03, schedule="serial_relative", max_iter=8, bp_method="minimum_sum", alpha=0.625, rows=16)
    run("stateful_rel_hgp1600_ps", hx, 0.03, schedule="serial_relative", max_iter=6, bp_method="product_sum", alpha=1.0, rows=12)

    # SoftInfoBpDecoder with random_serial_schedule
    from make_golden_soft import noisy  # noqa: E402
    run_soft("stateful_softrnd_bb144_s7", bb, noisy(bb, 21, 0.02, 64, 3.0), 0.02, max_iter=12, alpha=0.8, cutoff=4.0, sigma=1.5, seed=7)
    hs = codes.rotated_surface_code_x(7)
    run_soft("stateful_softrnd_surf7_s0", hs, noisy(hs, 23, 0.08, 48, 3.0), 0.08, max_iter=10, alpha=0.75, cutoff=2.0, sigma=1.0, seed=0)
    hl = codes.regular_ldpc_code(120, 3, 6, seed=2)
    run_soft("stateful_softrnd_ldpc120_sneg", hl, noisy(hl, 25, 0.03, 40, 4.0), 0.03, max_iter=15, alpha=0.75, cutoff=4.0, sigma=1.5, seed=-3)


if __name__ == "__main__":
    main()
