#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REAL reference decoder (oracle/_ref/libref_bp.so).

Run in the build container only (needs /root/reference):

    make -C oracle ref && python tests/golden/make_golden.py

Every fixture is DATA: inputs (H as CSR or as a generator recipe, channel probabilities, decoder
parameters, syndromes) and the outputs ``ldpc::bp::BpDecoder::decode`` (bp.hpp:159-325) produced for
them here (decoding, converge, iterations, log_prob_ratios).  No reference source is stored.

Cases follow SURVEY.md §7 step 1: the reference's own known-answer tests (cpp_test/TestBPDecoder.cpp:
122-164, 166-231, 301-344; python_test/test_bp_decoder.py:175-211) re-run through the reference so
that LLRs and iteration counts are pinned too, BASELINE.json configs 1-3 and 5 (BP part) at fixture
size, and edge cases (infinite priors, p >= 0.5, syndrome bytes > 1, degree-1 checks, NaN paths,
all-zero syndromes, adaptive min-sum scaling).
"""
from __future__ import annotations

import os
import sys
import zlib

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import RefBp, RefBpOsd, csr_arrays, have_ref  # noqa: E402
from ldpc_amd import codes  # noqa: E402
from ldpc_amd.prng import bernoulli_threshold, sm64  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bsc_syndromes(h, seed, p, shot0, shots):
    """Host twin of the synthetic generator: e ~ Bernoulli(p) from sm64, s = H e."""
    m, n = h.shape
    idx = (np.arange(shot0, shot0 + shots, dtype=np.uint64)[:, None] * np.uint64(n)
           + np.arange(n, dtype=np.uint64)[None, :])
    e = ((sm64(seed, idx) >> np.uint64(11)) < np.uint64(bernoulli_threshold(p))).astype(np.uint8)
    s = (sp.csr_matrix(h).astype(np.int64) @ e.T.astype(np.int64)).T % 2
    return np.ascontiguousarray(s, np.uint8)


def h_crc(h):
    m, n, rp, ci = csr_arrays(h)
    return zlib.crc32(ci.tobytes(), zlib.crc32(rp.tobytes(), zlib.crc32(np.array([m, n], np.int64).tobytes())))


def run_case(name, h, syndromes, *, error_rate=None, error_channel=None, max_iter=0,
             bp_method="product_sum", ms_scaling_factor=1.0, recipe=None, full_llr=None, note=""):
    h = sp.csr_matrix(h, dtype=np.uint8)
    m, n, rp, ci = csr_arrays(h)
    ref = RefBp(h, error_rate=error_rate, error_channel=error_channel, max_iter=max_iter,
                bp_method=bp_method, ms_scaling_factor=ms_scaling_factor)
    syndromes = np.ascontiguousarray(syndromes, np.uint8).reshape(-1, m)
    dec, llr, it, conv = ref.decode_batch(syndromes)
    k = syndromes.shape[0]
    if full_llr is None:
        full_llr = k
    payload = dict(
        name=name, note=note, m=m, n=n, h_crc=np.uint32(h_crc(h)),
        channel_probs=ref.channel_probs, max_iter=np.int32(ref.max_iter),
        bp_method=np.int32(0 if bp_method in ("product_sum", "ps") else 1),
        ms_scaling_factor=np.float64(ms_scaling_factor),
        syndromes=np.packbits(syndromes, axis=1) if syndromes.max(initial=0) <= 1 else syndromes,
        syndromes_packed=np.bool_(syndromes.max(initial=0) <= 1),
        decoding=np.packbits(dec, axis=1), converge=conv, iterations=it,
        llr=llr[:full_llr], llr_rowsum=np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1),  # NaN/inf/DBL_MAX-scale entries skipped
    )
    if recipe is None:
        payload.update(row_ptr=rp, col_idx=ci, recipe="")
    else:
        payload.update(recipe=recipe)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **payload)
    print(f"{name:34s} k={k:4d} conv={conv.mean():.3f} iters={it.mean():6.2f} "
          f"{os.path.getsize(path) / 1024:8.1f} KiB")


def run_serial_case(name, h, syndromes, *, error_rate, max_iter, bp_method="product_sum", ms_scaling_factor=1.0,
                    order=None, full_llr=None, note=""):
    """bp_decode_serial (bp.hpp:451-545) through the real reference, default or custom serial_schedule_order."""
    h = sp.csr_matrix(h, dtype=np.uint8)
    m, n, rp, ci = csr_arrays(h)
    ref = RefBp(h, error_rate=error_rate, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=ms_scaling_factor,
                schedule="serial")
    if order is not None:
        ref.set_serial_order(order)
    syndromes = np.ascontiguousarray(syndromes, np.uint8).reshape(-1, m)
    dec, llr, it, conv = ref.decode_batch(syndromes)
    k = len(syndromes) if full_llr is None else full_llr
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, name=name, note=note, m=m, n=n, h_crc=np.uint32(h_crc(h)), row_ptr=rp, col_idx=ci, recipe="",
                        channel_probs=ref.channel_probs, max_iter=np.int32(ref.max_iter),
                        bp_method=np.int32(0 if bp_method in ("product_sum", "ps") else 1),
                        ms_scaling_factor=np.float64(ms_scaling_factor),
                        syndromes=np.packbits(syndromes, axis=1) if syndromes.max(initial=0) <= 1 else syndromes,
                        syndromes_packed=np.bool_(syndromes.max(initial=0) <= 1),
                        decoding=np.packbits(dec, axis=1), converge=conv, iterations=it, llr=llr[:k],
                        llr_rowsum=np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1),
                        order=np.asarray(order if order is not None else [], np.int32))
    print(f"{name:34s} k={len(syndromes):4d} conv={conv.mean():.3f} iters={it.mean():6.2f} "
          f"{os.path.getsize(path) / 1024:8.1f} KiB")


def run_osd_case(name, h, syndromes, *, error_rate, max_iter, bp_method="product_sum", ms_scaling_factor=1.0, note=""):
    """BpOsdDecoder.decode (OSD_0) per row through the real reference (_bposd_decoder.pyx:125-134, osd.hpp:110-117)."""
    h = sp.csr_matrix(h, dtype=np.uint8)
    m, n, rp, ci = csr_arrays(h)
    ref = RefBpOsd(h, error_rate=error_rate, max_iter=max_iter, bp_method=bp_method, ms_scaling_factor=ms_scaling_factor)
    syndromes = np.ascontiguousarray(syndromes, np.uint8).reshape(-1, m)
    dec, llr, it, conv = ref.decode_batch(syndromes)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, name=name, note=note, m=m, n=n, h_crc=np.uint32(h_crc(h)), row_ptr=rp, col_idx=ci, recipe="",
                        channel_probs=ref.channel_probs, max_iter=np.int32(ref.max_iter),
                        bp_method=np.int32(0 if bp_method in ("product_sum", "ps") else 1),
                        ms_scaling_factor=np.float64(ms_scaling_factor), syndromes=np.packbits(syndromes, axis=1),
                        syndromes_packed=np.bool_(True), decoding=np.packbits(dec, axis=1), converge=conv, iterations=it,
                        llr=llr[:0], llr_rowsum=np.sum(np.where(np.abs(llr) < 1e100, llr, 0.0), axis=1))
    print(f"{name:34s} k={len(syndromes):4d} conv={conv.mean():.3f} osd rows={int((~conv).sum()):4d} "
          f"{os.path.getsize(path) / 1024:8.1f} KiB")


def chain(n):  # the 'pcm' every cpp_test/TestBPDecoder.cpp case builds (e.g. :126-129)
    return codes.rep_code(n)


def main():
    if not have_ref():
        raise SystemExit("build oracle/_ref first: make -C oracle ref")

    # --- reference known-answer tests, re-run through the reference (hard decisions asserted there) ---
    s3 = [[0, 0], [0, 1], [1, 0], [1, 1]]
    s5 = [[0, 0, 0, 0], [0, 0, 0, 1], [0, 1, 0, 1], [1, 0, 1, 0], [1, 1, 1, 1]]
    run_case("kat_chain3_ps", chain(3), s3, error_rate=0.1, max_iter=3, bp_method="product_sum",
             ms_scaling_factor=79879879.0, note="TestBPDecoder.cpp:122-164")
    run_case("kat_rep5_ps", chain(5), s5, error_rate=0.1, max_iter=5, bp_method="product_sum",
             ms_scaling_factor=4324234.0, note="TestBPDecoder.cpp:166-197")
    run_case("kat_rep5_ms", chain(5), s5, error_rate=0.1, max_iter=5, bp_method="minimum_sum",
             ms_scaling_factor=1.0, note="TestBPDecoder.cpp:200-231")
    run_case("kat_chain3_ms", chain(3), s3, error_rate=0.1, max_iter=3, bp_method="minimum_sum",
             ms_scaling_factor=0.625, note="TestBPDecoder.cpp:301-344")
    run_case("kat_rep3_ps_infprior", codes.rep_code(3), [[1, 1]], error_channel=[0.1, 0.0, 0.1],
             max_iter=0, bp_method="product_sum", note="test_bp_decoder.py:175-192 (p=0 -> +inf prior)")
    run_case("kat_rep3_ms_infprior", codes.rep_code(3), [[1, 1]], error_channel=[0.1, 0.0, 0.1],
             max_iter=0, bp_method="minimum_sum", ms_scaling_factor=1.0, note="test_bp_decoder.py:195-211")

    # --- BASELINE.json config 1: hamming_code(5), product_sum, max_iter=20 ---
    h = codes.hamming_code(5)
    run_case("c1_hamming5_ps20", h, bsc_syndromes(h, 7, 0.1, 0, 64), error_rate=0.1, max_iter=20,
             bp_method="product_sum", note="BASELINE.json configs[0]")
    run_case("c1_hamming5_ms20", h, bsc_syndromes(h, 7, 0.1, 0, 64), error_rate=0.1, max_iter=20,
             bp_method="minimum_sum", ms_scaling_factor=0.9)

    # --- config 2 at fixture size: (3,6)-regular n=10000, product_sum, 50 iters ---
    h = codes.regular_ldpc_code(10_000, 3, 6, seed=1)
    rec = "regular_ldpc_code(10000,3,6,seed=1)"
    for p, tag in ((0.05, "p050"), (0.08, "p080"), (0.09, "p090")):
        run_case(f"c2_ldpc36_n10000_ps50_{tag}", h, bsc_syndromes(h, 7, p, 0, 12), error_rate=p,
                 max_iter=50, bp_method="product_sum", recipe=rec, full_llr=2,
                 note="BASELINE.json configs[1] shape; error seed 7, shots 0..11")
    run_case("c2_ldpc36_n10000_ms50_p050", h, bsc_syndromes(h, 7, 0.05, 0, 8), error_rate=0.05,
             max_iter=50, bp_method="minimum_sum", ms_scaling_factor=0.625, recipe=rec, full_llr=2)
    # a small sibling of the same family that the pure-CPU tests can sweep quickly
    hs = codes.regular_ldpc_code(600, 3, 6, seed=3)
    for p, tag in ((0.04, "p040"), (0.07, "p070")):
        run_case(f"ldpc36_n600_ps50_{tag}", hs, bsc_syndromes(hs, 11, p, 0, 96), error_rate=p,
                 max_iter=50, bp_method="product_sum", full_llr=16)
    run_case("ldpc36_n600_ms30_adaptive", hs, bsc_syndromes(hs, 11, 0.05, 0, 96), error_rate=0.05,
             max_iter=30, bp_method="minimum_sum", ms_scaling_factor=0.0, full_llr=16,
             note="ms_scaling_factor==0 -> alpha = 1 - 2^-it (bp.hpp:222-228)")

    # --- config 3: rotated surface code d=21 (and d=5), minimum_sum 0.625, 30 iters ---
    h = codes.rotated_surface_code_x(21)
    for p, tag in ((0.05, "p050"), (0.01, "p010")):
        run_case(f"c3_surface21_ms30_{tag}", h, bsc_syndromes(h, 7, p, 0, 128), error_rate=p,
                 max_iter=30, bp_method="minimum_sum", ms_scaling_factor=0.625, full_llr=32,
                 note="BASELINE.json configs[2] shape")
    h5 = codes.rotated_surface_code_x(5)
    run_case("surface5_ms30", h5, bsc_syndromes(h5, 7, 0.06, 0, 128), error_rate=0.06, max_iter=30,
             bp_method="minimum_sum", ms_scaling_factor=0.625)
    run_case("surface5_ps30", h5, bsc_syndromes(h5, 7, 0.06, 0, 128), error_rate=0.06, max_iter=30,
             bp_method="product_sum")

    # --- config 5 (BP part): BB [[144,12,12]] hx, product_sum, 50 iters ---
    h = codes.bivariate_bicycle_hx()
    run_case("c5_bb144_ps50_p050", h, bsc_syndromes(h, 7, 0.05, 0, 256), error_rate=0.05, max_iter=50,
             bp_method="product_sum", full_llr=64, note="BASELINE.json configs[4], BP stage")
    run_case("c5_bb144_ms50_p050", h, bsc_syndromes(h, 7, 0.05, 0, 256), error_rate=0.05, max_iter=50,
             bp_method="minimum_sum", ms_scaling_factor=0.625, full_llr=64)

    # --- config 5: BP-50 + OSD-0 on BB [[144,12,12]] (and siblings whose H is rank deficient / irregular) ---
    h = codes.bivariate_bicycle_hx()
    run_osd_case("osd_c5_bb144_ps50_p050", h, bsc_syndromes(h, 7, 0.05, 0, 1024), error_rate=0.05, max_iter=50,
                 note="BASELINE.json configs[4]")
    run_osd_case("osd_bb144_ps50_p070", h, bsc_syndromes(h, 9, 0.07, 0, 512), error_rate=0.07, max_iter=50)
    run_osd_case("osd_bb144_ms50_p060", h, bsc_syndromes(h, 9, 0.06, 0, 512), error_rate=0.06, max_iter=50,
                 bp_method="minimum_sum", ms_scaling_factor=0.625)
    hs = codes.rotated_surface_code_x(7)
    run_osd_case("osd_surface7_ms30", hs, bsc_syndromes(hs, 7, 0.08, 0, 512), error_rate=0.08, max_iter=30,
                 bp_method="minimum_sum", ms_scaling_factor=0.625, note="many exact LLR ties: the stable column order matters")
    hh = codes.hamming_code(6)
    run_osd_case("osd_hamming6_ps10", hh, bsc_syndromes(hh, 7, 0.06, 0, 256), error_rate=0.06, max_iter=10)
    hr = codes.ring_code(40)
    run_osd_case("osd_ring40_ps3", hr, bsc_syndromes(hr, 7, 0.12, 0, 256), error_rate=0.12, max_iter=3,
                 note="rank-deficient H (m = n, rank n-1), BP cut short so OSD runs often")

    # --- serial schedule, fixed order ---
    h = codes.bivariate_bicycle_hx()
    perm = (sm64(13, np.arange(144, dtype=np.uint64)) % np.uint64(1 << 40)).argsort().astype(np.int32)
    run_serial_case("serial_bb144_ps30", h, bsc_syndromes(h, 7, 0.06, 0, 192), error_rate=0.06, max_iter=30, full_llr=48)
    run_serial_case("serial_bb144_ms30_custom_order", h, bsc_syndromes(h, 7, 0.06, 0, 192), error_rate=0.06, max_iter=30,
                    bp_method="minimum_sum", ms_scaling_factor=0.625, order=perm, full_llr=48)
    run_serial_case("serial_bb144_ms30_adaptive", h, bsc_syndromes(h, 8, 0.06, 0, 128), error_rate=0.06, max_iter=30,
                    bp_method="minimum_sum", ms_scaling_factor=0.0, full_llr=32)
    hs = codes.rotated_surface_code_x(7)
    run_serial_case("serial_surface7_ps20", hs, bsc_syndromes(hs, 7, 0.07, 0, 192), error_rate=0.07, max_iter=20, full_llr=48)
    hl = codes.regular_ldpc_code(600, 3, 6, seed=3)
    run_serial_case("serial_ldpc36_n600_ps20", hl, bsc_syndromes(hl, 11, 0.08, 0, 96), error_rate=0.08, max_iter=20, full_llr=16)
    hm6 = codes.hamming_code(6)  # row weight 32: beyond the register bounds -> streaming path of the kernel
    sb = bsc_syndromes(hm6, 7, 0.05, 0, 96)
    sb[::7, 2] = 2
    sb[3::9, 4] = 3  # syndrome bytes > 1: pow(-1, byte) sign, never converges (bp.hpp:499, 540)
    run_serial_case("serial_hamming6_ps10_bytes", hm6, sb, error_rate=0.05, max_iter=10, full_llr=24)
    run_serial_case("serial_hamming6_ms10", hm6, bsc_syndromes(hm6, 7, 0.05, 0, 96), error_rate=0.05, max_iter=10,
                    bp_method="minimum_sum", ms_scaling_factor=0.75, full_llr=24)

    # --- edge cases (SURVEY.md §7 "Inf/NaN semantics", §8a a7/a9) ---
    rng_idx = np.arange(31, dtype=np.uint64)
    hm = codes.hamming_code(5)
    chan = 0.02 + 0.3 * ((sm64(5, rng_idx) >> np.uint64(11)).astype(np.float64) / 2.0 ** 53)
    run_case("edge_nonuniform_channel_ps", hm, bsc_syndromes(hm, 3, 0.1, 0, 32), error_channel=chan,
             max_iter=20, bp_method="product_sum")
    chan2 = chan.copy()
    chan2[[0, 5, 9]] = [0.5, 0.7, 1.0]  # llr0 = 0, < 0, -inf
    chan2[[12]] = 0.0  # +inf
    run_case("edge_extreme_priors_ps", hm, bsc_syndromes(hm, 3, 0.1, 0, 32), error_channel=chan2,
             max_iter=20, bp_method="product_sum", note="priors 0, .5, .7, 1 -> +inf, 0, negative, -inf LLRs; NaN paths")
    run_case("edge_extreme_priors_ms", hm, bsc_syndromes(hm, 3, 0.1, 0, 32), error_channel=chan2,
             max_iter=20, bp_method="minimum_sum", ms_scaling_factor=0.75)
    s = bsc_syndromes(hm, 3, 0.1, 0, 16)
    s[::2, 1] = 2  # even byte: ms parity ignores it, ps sign sees != 0, never converges
    s[1::4, 3] = 3
    run_case("edge_syndrome_bytes_gt1_ps", hm, s, error_rate=0.1, max_iter=10, bp_method="product_sum",
             note="syndrome[i] != 0 flips the sign (bp.hpp:213); bytes > 1 can never converge (bp.hpp:300)")
    run_case("edge_syndrome_bytes_gt1_ms", hm, s, error_rate=0.1, max_iter=10, bp_method="minimum_sum",
             ms_scaling_factor=1.0, note="total_sgn = syndrome[i] (bp.hpp:236)")
    # degree-1 checks and an isolated (weight-0) column and an empty row
    hd = sp.csr_matrix(np.array([[1, 0, 0, 0, 0, 0],
                                 [1, 1, 0, 0, 0, 0],
                                 [0, 1, 1, 1, 0, 0],
                                 [0, 0, 0, 0, 0, 0],
                                 [0, 0, 0, 1, 1, 0]], dtype=np.uint8))
    sd = np.array([[a, b, c, 0, d] for a in (0, 1) for b in (0, 1) for c in (0, 1) for d in (0, 1)]
                  + [[0, 0, 0, 1, 0]], dtype=np.uint8)
    run_case("edge_degree1_empty_ps", hd, sd, error_rate=0.15, max_iter=8, bp_method="product_sum",
             note="weight-1 row (c2b = log(2/0)... = +-inf? no: log((1+1)/(1-1))), empty row, weight-0 column")
    run_case("edge_degree1_empty_ms", hd, sd, error_rate=0.15, max_iter=8, bp_method="minimum_sum",
             ms_scaling_factor=0.5, note="weight-1 row -> magnitude DBL_MAX * alpha (bp.hpp:237,252)")
    # all-zero syndromes THROUGH the C++ decoder (the Python zero shortcut, pyx:679-681, is host logic)
    hz = codes.ring_code(7)
    run_case("edge_zero_syndrome_ps", hz, np.zeros((2, 7), np.uint8), error_rate=0.1, max_iter=7)
    run_case("edge_saturation_ps", codes.rep_code(9), bsc_syndromes(codes.rep_code(9), 2, 0.2, 0, 32),
             error_rate=1e-9, max_iter=9, bp_method="product_sum",
             note="tiny p: |b2c|/2 > 19 -> tanh == 1.0 -> log(2/0) = inf")


if __name__ == "__main__":
    main()
