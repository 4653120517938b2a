#!/usr/bin/env python3
"""Differential run of the stateful schedules (by hand for as long as wanted; a bounded slice with fixed seeds is collected by
tests/test_gpu_fuzz_soak.py): random regular and irregular codes of 24 .. 700 bits, random priors (uniform or per bit), methods, iteration
limits, batch sizes and STARTING ORDERS (identity, a permutation, an order with repeated bits), schedule = serial_relative through every
form of the on-chip kernel -- level by level with the scratch in the posterior array / apart, bit by bit with 64 / 16 lanes per syndrome --
and through the per-lane kernel, plus the fixed-order serial schedule and the random one (walking kernel / level kernel), and soft-syndrome decoding with both: decisions, iteration counts, flags, log-ratio BITS and the order
left behind, every row against the CPU checker (oracle/, pinned to the reference and to the host's std::sort).
    python tests/fuzz_schedules.py <seconds> <seed>"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, scipy.sparse as sp
import oracle
from ldpc_amd.engine import HipBpEngine
from ldpc_amd import codes

FORMS = ((), (("REL_SCRATCH_IN_L", 0),), (("REL_LEVELS", 0),), (("REL_LDS", 16),), (("REL_LDS", 0),))


def random_code(rng):
    kind = rng.random()
    if kind < 0.55:
        n = int(rng.choice([24, 48, 96, 144, 200, 330, 441, 520, 700]))
        dv, dc = (3, 6) if rng.random() < 0.5 else (4, 8) if rng.random() < 0.4 else (2, 4)
        n -= n % dc
        return sp.csr_matrix(codes.regular_ldpc_code(n, dv, dc, seed=int(rng.integers(1, 1000))))
    if kind < 0.8:
        n = int(rng.choice([60, 120, 300, 480]))
        return sp.csr_matrix(codes.irregular_ldpc_code(n, n // 2, seed=int(rng.integers(1, 1000)), row_weights=(3, 4, 5, 6, 7, 8, 9, 10, 12, 16),
                                                       col_weights=((2, 0.20), (3, 0.50), (6, 0.15), (8, 0.15))))
    if kind < 0.9:
        return sp.csr_matrix(codes.rotated_surface_code_x(int(rng.choice([5, 9, 13, 21]))))
    return sp.csr_matrix(codes.bivariate_bicycle_hx())


def run(seconds=120.0, seed=1, max_cases=None):
    """Random cases until `seconds` have passed or `max_cases` are done; returns the number of cases (asserts on any mismatch)."""
    oracle.build(ref=False)
    t_end = time.time() + float(seconds)
    rng = np.random.default_rng(int(seed))
    n_ok = 0
    while time.time() < t_end and (max_cases is None or n_ok < max_cases):
        h = random_code(rng)
        m, n = h.shape
        method = 0 if rng.random() < 0.4 else 1
        alpha = float(rng.choice([0.0, 0.625, 1.0]))
        max_iter = int(rng.integers(1, 12))
        p = float(rng.choice([0.02, 0.05, 0.09]))
        probs = np.full(n, p) if rng.random() < 0.6 else rng.uniform(0.005, 0.2, n)
        B = int(rng.choice([1, 5, 64, 300, 1500]))
        order = None
        r = rng.random()
        if r < 0.3:
            order = rng.permutation(n).astype(np.int32)
        elif r < 0.45:
            order = rng.integers(0, n, n).astype(np.int32)  # repeats: the bit-by-bit walk
        e = (rng.random((B, n)) < probs).astype(np.uint8)
        s = np.asarray((h @ e.T % 2).T, dtype=np.uint8)
        if rng.random() < 0.3:
            s[rng.integers(0, B)] = rng.integers(0, 3, m)  # bytes > 1, out of image
        o = oracle.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
        want = o.decode_serial_relative_batch(s, order_state=order, fresh=True)
        tag0 = f"{m}x{n} nnz={h.nnz} method={method} a={alpha} it={max_iter} B={B} order={'id' if order is None else 'perm' if r < 0.3 else 'repeats'}"
        for sw in FORMS:
            eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, method, alpha)
            eng.set_schedule("serial_relative", order)
            for k, v in sw:
                eng.set_debug_switch(k, v)
            got = eng.decode_batch(s) + (eng.schedule_order(),)
            eng.close()
            tag = f"{tag0} switches={sw}"
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(np.asarray(got[3], bool), want[3]), tag
            assert oracle.bits_equal(got[1], want[1]), "llr " + tag
            assert np.array_equal(got[4], want[4]), "order " + tag
        # the fixed-order serial schedule on the same inputs (a permutation or the identity)
        if order is None or r < 0.3:
            eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, method, alpha)
            eng.set_schedule("serial", order)
            got = eng.decode_batch(s)
            eng.close()
            ws = o.decode_serial_batch(s, order)
            assert np.array_equal(got[0], ws[0]) and np.array_equal(got[2], ws[2]) and oracle.bits_equal(got[1], ws[1]), "serial " + tag0
        # the random serial schedule (a new order per iteration, the same for every row): one wavefront per tile walking the order, and the
        # level kernel on the per-iteration level tables -- rows of the call start from the handle's state, which moves on call by call
        if order is None:
            seed_r = int(rng.integers(0, 2**31 - 1))
            wr = o.decode_random_serial_batch(s, seed_r)
            for mode in (0, 1, -1):
                eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, method, alpha)
                eng.set_schedule("serial")
                eng.set_random_serial(True, seed_r)
                eng.set_serial_kernel(mode)
                got = eng.decode_batch(s)
                eng.close()
                assert np.array_equal(got[0], wr[0]) and np.array_equal(got[2], wr[2]) and np.array_equal(np.asarray(got[3], bool), wr[3]), f"random serial {tag0} kernel={mode}"
                assert oracle.bits_equal(got[1], wr[1]), f"random serial llr {tag0} kernel={mode}"
        # soft-syndrome decoding (serial min-sum, bp.hpp:547-660), fixed order and with the random schedule (a NEW engine seeded alike per
        # shuffle, bp.hpp:573-577): walking kernel / level kernel
        if method == 1 and order is None and rng.random() < 0.5:
            Bs = min(B, 300)
            soft = (1 - 2 * s[:Bs].astype(np.float64) % 2) * 2 + rng.uniform(-3.0, 3.0, (Bs, m))
            cutoff, sigma = float(rng.choice([2.0, 4.0])), float(rng.choice([1.0, 1.5]))
            seed_s = int(rng.integers(-5, 2**31 - 1))
            for shuffled in (False, True):
                ws = o.soft_info_decode_random_batch(soft, cutoff, sigma, seed_s) if shuffled else o.soft_info_decode_batch(soft, cutoff, sigma)
                for mode in (0, 1, -1):
                    eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, 1, alpha)
                    eng.set_schedule("serial")
                    if shuffled:
                        eng.set_random_serial(True, seed_s & 0xffffffff)
                    eng.set_serial_kernel(mode)
                    got = eng.soft_info_decode_batch(soft, cutoff, sigma)
                    eng.close()
                    t = f"soft {tag0} shuffled={shuffled} kernel={mode} cutoff={cutoff} sigma={sigma}"
                    assert np.array_equal(got[0], ws[0]) and np.array_equal(got[2], ws[2]) and np.array_equal(np.asarray(got[3], bool), ws[3]), t
                    assert oracle.bits_equal(got[1], ws[1]) and oracle.bits_equal(got[4], ws[4]), "values " + t
        n_ok += 1
    return n_ok


if __name__ == "__main__":
    print("cases passed:", run(float(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
