#!/usr/bin/env python3
"""Extended differential run of the STREAMED SERIAL decode (host_serial.h: decode_serial_streamed; bp_serial_stream_kernel,
bp_serial_lane_kernel, the lane compaction between passes), by hand for as long as wanted: (3,6)-regular codes of 1200 .. 10000 bits,
product-sum and min-sum, fixed order / a caller's permutation / an order with bits repeated and missing, batches from a few rows (lane
kernel from the start) to tens of tiles, random pass lengths, ring depths, wavefront counts, lane limits and round sizes
(ldpc_hip_bp_set_debug_switch) -- decisions, iteration counts, flags and LOG-RATIO BITS of a row sample against the CPU checker
(oracle/: decode_serial_batch, bp.hpp:451-545), and every variant against the first bit for bit over ALL rows.
      python tests/fuzz_serial_stream.py <seconds> <seed>"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import oracle
from ldpc_amd.engine import HipBpEngine
from ldpc_amd import codes
from ldpc_amd.noise_models import generate_bsc_batch


def run(seconds=120.0, seed=1, max_cases=None, shapes=True):
    oracle.build(ref=False)
    t_end = time.time() + float(seconds)
    rng = np.random.default_rng(int(seed))
    n_ok = 0
    while time.time() < t_end and (max_cases is None or n_ok < max_cases):
        n = int(rng.choice([1200, 2400, 4800, 10000]))
        # (round 6: any degree profile -- the item form, csrc/bp_serial_var_kernel.h; (3,6) codes take the form built around their record, or,
        # with SER_VAR 1, the item form too)
        shape = str(rng.choice(["ldpc36", "ldpc36", "ldpc48", "ldpc34", "ldpc510", "irregular"])) if shapes else "ldpc36"
        if shape == "irregular":
            h = codes.irregular_ldpc_code(n, n // 2, seed=int(rng.integers(1, 1000)))
        else:
            dv, dc = {"ldpc36": (3, 6), "ldpc48": (4, 8), "ldpc34": (3, 4), "ldpc510": (5, 10)}[shape]
            if shape == "ldpc510": n = n // 10 * 10
            h = codes.regular_ldpc_code(n, dv, dc, seed=int(rng.integers(1, 1000)))
        method = "product_sum" if rng.random() < 0.6 else "minimum_sum"
        alpha = 1.0 if method == "product_sum" else float(rng.choice([0.0, 0.625, 0.9]))
        max_iter = int(rng.choice([3, 9, 20, 40]))
        p = float(rng.choice([0.04, 0.06, 0.075, 0.09])) * {"ldpc36": 1.0, "ldpc48": 0.8, "ldpc34": 1.6, "ldpc510": 0.6, "irregular": 0.7}[shape]
        B = int(rng.choice([5, 64, 200, 700, 3000]))
        err = generate_bsc_batch(n, p, seed=int(rng.integers(1, 10000)), shot0=0, shots=B)
        synd = np.ascontiguousarray((h.astype(np.int64) @ err.T.astype(np.int64)).T % 2, np.uint8)
        if rng.random() < 0.3: synd[int(rng.integers(B)), int(rng.integers(h.shape[0]))] = 2  # a byte > 1: that row never converges
        kind = int(rng.integers(3))
        order = None
        if kind >= 1:
            order = rng.permutation(n).astype(np.int32)
        if kind == 2:
            k = len(order[1:n:11]); order[0:11 * k:11] = order[1:n:11]  # some bits twice, some never
        rows = np.arange(B) if B <= 200 else np.sort(rng.choice(B, size=120, replace=False))
        want = oracle.BpOracle(h, error_rate=p, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha).decode_serial_batch(synd[rows], order)
        eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 0 if method == "product_sum" else 1, alpha)
        eng.set_schedule("serial", order)
        eng.set_serial_kernel(2)
        variants = [dict(),
                    dict(repack=0),
                    dict(repack=int(rng.choice([1, 2, 3, 5])), SER_LANE_MAX=int(rng.choice([0, 16, 300]))),
                    dict(SER_RING=2, SER_WAVES=int(rng.choice([1, 4, 8, 16]))),
                    dict(SER_ROUND_TILES=int(rng.choice([1, 2, 8])), SER_WAVES2=int(rng.choice([4, 8, 16]))),
                    dict(SER_NO_REMAINDER=1, EXPLICIT_INIT=1, SER_LANE_THREADS=int(rng.choice([256, 512, 1024]))),
                    dict(SER_VAR=1, SER_VAR_UNITS=int(rng.choice([8, 9, 12, 16])), SER_WAVES=int(rng.choice([1, 3, 8, 16])), repack=int(rng.choice([-1, 0, 2]))),
                    dict(SER_VAR=0)]
        if rng.random() < 0.3:
            variants.append(dict(chunk=int(rng.choice([2, 70]))))  # the state of the batch not resident at once: pieces, or the one-pass chunk loop
        first = None
        for v in variants:
            eng.set_repack(v.get("repack", -1))
            eng.set_tuning(max_chunk_tiles=v.get("chunk", 0))
            for k, val in v.items():
                if k not in ("repack", "chunk"): eng.set_debug_switch(k, val)
            for want_llr in (True, False):
                got = eng.decode_batch(synd, want_llr=want_llr)
                tag = f"{shape} n={n} {method} a={alpha} it={max_iter} p={p} B={B} order={kind} {v} llr={want_llr}"
                assert np.array_equal(got[0][rows], want[0]) and np.array_equal(got[2][rows], want[2]) and np.array_equal(got[3][rows].astype(bool), want[3].astype(bool)), tag
                if want_llr:
                    assert oracle.bits_equal(got[1][rows], want[1]), "llr " + tag
                    if first is None: first = got
                    else: assert oracle.bits_equal(got[1], first[1]), "llr vs first variant " + tag
                if first is not None:
                    assert np.array_equal(got[0], first[0]) and np.array_equal(got[2], first[2]) and np.array_equal(got[3], first[3]), "vs first variant " + tag
            for k in v:
                if k not in ("repack", "chunk"): eng.set_debug_switch(k, -1)
        eng.close()
        n_ok += 1
    return n_ok


if __name__ == "__main__":
    print("cases passed:", run(float(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
