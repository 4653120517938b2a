#!/usr/bin/env python3
"""Golden fixtures for higher-order OSD: BpOsdDecoder.decode with osd_method OSD_E / OSD_CS and osd_order > 0
(_bposd_decoder.pyx:125-134 -> ldpc::osd::OsdDecoder::decode, osd.hpp:119-187) through the REAL reference
(oracle/_ref/libref_bp.so).  Build container only:

    make -C oracle ref && python tests/golden/make_golden_osdw.py

``decoding`` = the swept solution (osdw_decoding; BP's decision for converged rows), ``osd0_decoding`` = the same
decoder at order 0, ``converge`` / ``iterations`` / ``llr_rowsum`` = BP's.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from oracle import RefBpOsd, csr_arrays  # noqa: E402
from ldpc_amd import codes  # noqa: E402
from ldpc_amd.prng import sm64  # noqa: E402
from make_golden import bsc_syndromes, h_crc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
METHODS = {"osd_e": 2, "osd_cs": 3}


def run(name, h, syndromes, *, osd_method, osd_order, max_iter, error_rate=None, error_channel=None,
        bp_method="product_sum", ms_scaling_factor=1.0, note=""):
    h = sp.csr_matrix(h, dtype=np.uint8)
    m, n, rp, ci = csr_arrays(h)
    kw = dict(error_rate=error_rate, error_channel=error_channel, max_iter=max_iter, bp_method=bp_method,
              ms_scaling_factor=ms_scaling_factor)
    ref = RefBpOsd(h, osd_method=METHODS[osd_method], osd_order=osd_order, **kw)
    ref0 = RefBpOsd(h, **kw)
    syndromes = np.ascontiguousarray(syndromes, np.uint8).reshape(-1, m)
    dec, llr, it, conv = ref.decode_batch(syndromes)
    dec0 = ref0.decode_batch(syndromes)[0]
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, name=name, note=note, m=m, n=n, h_crc=np.uint32(h_crc(h)), row_ptr=rp, col_idx=ci, recipe="",
                        channel_probs=ref.channel_probs, max_iter=np.int32(ref.max_iter),
                        bp_method=np.int32(0 if bp_method in ("product_sum", "ps") else 1),
                        ms_scaling_factor=np.float64(ms_scaling_factor), syndromes=np.packbits(syndromes, axis=1),
                        syndromes_packed=np.bool_(True), decoding=np.packbits(dec, axis=1), converge=conv, iterations=it,
                        llr=llr[:0], llr_rowsum=np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1),
                        osd_method=np.int32(METHODS[osd_method]), osd_order=np.int32(osd_order), k=np.int32(ref.k),
                        osd0_decoding=np.packbits(dec0, axis=1))
    improved = int(((dec != dec0).any(axis=1)).sum())
    print(f"{name:34s} k={ref.k:3d} rows={len(syndromes):4d} osd rows={int((~conv).sum()):4d} "
          f"sweep changed={improved:4d} {os.path.getsize(path) / 1024:8.1f} KiB")


def wide_orders(hx, s):
    """Round 4: OSD_CS orders past 64 (the device's former limit) -- pairs that reach beyond the first 64 non-pivot columns."""
    run("osdw_cs78_bb144_ps8", hx, s[:96], osd_method="osd_cs", osd_order=78, max_iter=8, error_rate=0.08,
        note="osd_order = k = 78: every pair of non-pivot columns, 78 + 3003 candidates (register-resident elimination)")
    run("osdw_cs70_bb144_nonuniform", hx, s[:64], osd_method="osd_cs", osd_order=70, max_iter=8,
        error_channel=0.02 + 0.12 * ((sm64(13, np.arange(144, dtype=np.uint64)) >> np.uint64(11)).astype(np.float64) / 2.0 ** 53),
        note="64 < osd_order < k with non-uniform priors")
    rng = np.random.default_rng(11)
    m, n = 120, 600
    h = sp.csr_matrix((np.ones(m * 12, np.uint8), (np.repeat(np.arange(m), 12), rng.integers(0, n, size=m * 12))), shape=(m, n))
    h.sum_duplicates()
    h.data[:] = 1
    run("osdw_cs90_random120x600_ps4", h, bsc_syndromes(h, 9, 0.03, 0, 40), osd_method="osd_cs", osd_order=90, max_iter=4,
        error_channel=rng.uniform(0.01, 0.06, size=n), note="120 x 600 (k >= 480): the one-wavefront kernel with [H | s] in LDS")


def main():
    hx = codes.bivariate_bicycle_hx()
    s = bsc_syndromes(hx, 11, 0.08, 0, 384)
    run("osdw_cs10_bb144_ps8", hx, s, osd_method="osd_cs", osd_order=10, max_iter=8, error_rate=0.08)
    run("osdw_cs1_bb144_ps8", hx, s[:192], osd_method="osd_cs", osd_order=1, max_iter=8, error_rate=0.08,
        note="order 1: the k weight-one strings only")
    run("osdw_cs64_bb144_ms8", hx, s[:128], osd_method="osd_cs", osd_order=64, max_iter=8, error_rate=0.08,
        bp_method="minimum_sum", ms_scaling_factor=0.75, note="largest order the device takes: 78 + 2016 candidates")
    run("osdw_e8_bb144_ps8", hx, s[:256], osd_method="osd_e", osd_order=8, max_iter=8, error_rate=0.08)
    run("osdw_e1_bb144_ps8", hx, s[:128], osd_method="osd_e", osd_order=1, max_iter=8, error_rate=0.08)
    chan = 0.02 + 0.12 * ((sm64(13, np.arange(144, dtype=np.uint64)) >> np.uint64(11)).astype(np.float64) / 2.0 ** 53)
    run("osdw_cs12_bb144_nonuniform", hx, s[:256], osd_method="osd_cs", osd_order=12, max_iter=8, error_channel=chan,
        note="non-uniform priors: candidate weights are order-sensitive FP64 sums of log(1/p_j)")
    run("osdw_e10_bb144_nonuniform", hx, s[:128], osd_method="osd_e", osd_order=10, max_iter=8, error_channel=chan)
    hs = codes.rotated_surface_code_x(7)
    run("osdw_cs6_surface7_ms5", hs, bsc_syndromes(hs, 7, 0.1, 0, 256), osd_method="osd_cs", osd_order=6, max_iter=5,
        error_rate=0.1, bp_method="minimum_sum", ms_scaling_factor=0.625)
    hm = codes.hamming_code(4)
    sh = bsc_syndromes(hm, 5, 0.2, 0, 128)
    run("osdw_e13_hamming4_ps2", hm, sh, osd_method="osd_e", osd_order=13, max_iter=2, error_rate=0.2,
        note="order > k = 11: bits of the candidate number beyond the k-th are dropped (util.hpp:12-38)")
    run("osdw_cs11_hamming4_ps2", hm, sh, osd_method="osd_cs", osd_order=11, max_iter=2, error_rate=0.2,
        note="order == k: every pair of non-pivot columns")
    wide_orders(hx, s)
    hr = codes.ring_code(40)
    run("osdw_cs1_ring40_ps3", hr, bsc_syndromes(hr, 7, 0.12, 0, 128), osd_method="osd_cs", osd_order=1, max_iter=3,
        error_rate=0.12, note="rank-deficient square matrix, k = 1")


if __name__ == "__main__":
    if "--wide-only" in sys.argv:
        hx_ = codes.bivariate_bicycle_hx()
        wide_orders(hx_, bsc_syndromes(hx_, 11, 0.08, 0, 384))
    else:
        main()
