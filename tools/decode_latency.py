#!/usr/bin/env python3
"""Latency of ONE decode() call through the Python mirror (Cython and ctypes back ends): hamming(5), BB144, the n = 10 000 code --
through the resident workgroup (default, host_onchip.h: decode_onchip_resident) and with a launch per call (LDPC_HIP_RESIDENT=0).
Run on an MI355X:   python tools/decode_latency.py   (profiles/r5_single_decode_latency.txt)"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from ldpc_amd.bp_decoder import BpDecoder
    from ldpc_amd import codes
    for name, h, p, mi in (("hamming(5) 5x31", codes.hamming_code(5), 0.05, 20), ("BB144", codes.bivariate_bicycle_hx(), 0.05, 50), ("ldpc n=10000", codes.regular_ldpc_code(10000, 3, 6, seed=1), 0.05, 50)):
        m, n = h.shape
        rng = np.random.default_rng(0)
        e = (rng.random(n) < p).astype(np.uint8)
        s = (h @ e % 2).astype(np.uint8)
        for backend in ("cython", "ctypes"):
            d = BpDecoder(h, error_rate=p, max_iter=mi, bp_method="product_sum", _backend=backend)
            d.decode(s)
            t0 = time.perf_counter()
            for _ in range(500):
                d.decode(s)
            dt = (time.perf_counter() - t0) / 500
            print(f"{name:16s} {backend:7s} decode() {dt*1e6:8.1f} us per call, iterations {d.iter}   [LDPC_HIP_RESIDENT={os.environ.get('LDPC_HIP_RESIDENT', 'default (on)')}]", flush=True)
else:
    for res in (None, "0"):
        env = dict(os.environ)
        if res is not None:
            env["LDPC_HIP_RESIDENT"] = res
        subprocess.run([sys.executable, __file__, "child"], env=env, check=False)
