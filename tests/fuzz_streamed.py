#!/usr/bin/env python3
"""Extended differential run of the STREAMED kernels (by hand for as long as wanted; a bounded slice with fixed seeds is collected as tests/test_gpu_fuzz_soak.py): regular codes of 600 .. 3000 bits forced
off the on-chip kernels (ldpc_hip_bp_set_small_code_kernel 0), operating points where syndromes converge at different iterations, batches
from one tile to a few hundred (per-pass launches from the start, or the persistent kernel handing its last tiles over), hand-off
thresholds, chunked workspaces and the two-pass decode with lane compaction; a third of the cases on IRREGULAR codes (rows of 3 .. 16
entries, columns of 2 .. 8: per-pass kernels from the start by default for product-sum, the register variant of the persistent kernel,
its variable-degree LDS ring with queues of 8 .. 20 units) -- decisions, iteration counts, flags and LOG-RATIO BITS of
every row against the CPU checker.      python tests/fuzz_streamed.py <seconds> <seed>"""
import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, scipy.sparse as sp
import oracle
from ldpc_amd.engine import HipBpEngine
from ldpc_amd import codes


def run(seconds=120.0, seed=1, max_cases=None):
    """Random cases until `seconds` have passed or `max_cases` are done; returns the number of cases (asserts on any mismatch)."""
    oracle.build(ref=False)
    t_end = time.time() + float(seconds)
    rng = np.random.default_rng(int(seed))
    n_ok = 0
    while time.time() < t_end and (max_cases is None or n_ok < max_cases):
        n = int(rng.choice([600, 1200, 2400, 3000]))
        dv, dc = (3, 6) if rng.random() < 0.7 else (4, 8)
        irregular = rng.random() < 0.34
        if irregular: h = sp.csr_matrix(codes.irregular_ldpc_code(n, n // 2, seed=int(rng.integers(1, 1000))))
        else: h = sp.csr_matrix(codes.regular_ldpc_code(n, dv, dc, seed=int(rng.integers(1, 1000))))
        method = 0 if rng.random() < 0.6 else 1
        alpha = 1.0 if method == 0 else float(rng.choice([0.0, 0.625, 0.9]))
        max_iter = int(rng.choice([3, 8, 15, 30]))
        p = float(rng.choice([0.02, 0.03, 0.04]) if irregular else rng.choice([0.03, 0.045, 0.06, 0.08]))
        B = int(rng.choice([64, 300, 1000, 4096, 20000]))
        probs = np.full(n, p)
        o = oracle.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
        eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, method, alpha)
        eng.set_small_code_kernel(0)
        s = eng.gen_bsc_syndromes(int(rng.integers(1, 1000)), p, shot0=0, shots=B, device="cuda:0")
        s_host = s.cpu().numpy()
        rows = np.arange(B) if B <= 1000 else np.sort(rng.choice(B, size=700, replace=False))
        want = o.decode_batch(s_host[rows])
        variants = [dict(), dict(handoff=int(rng.choice([0, 4, 64]))), dict(chunk=int(rng.choice([3, 17]))), dict(repack=int(rng.choice([1, 2, 4]))), dict(ring=int(rng.choice([0, 2, 3])))]
        if irregular:
            variants = [dict(), dict(handoff=int(rng.choice([0, 64, 256]))), dict(var_ring=1, handoff=int(rng.choice([-1, 0, 7]))),
                        dict(var_ring=1, units=int(rng.choice([8, 9, 14, 20])), waves=int(rng.choice([1, 4, 9, 12])), repack=int(rng.choice([0, 2]))), dict(chunk=int(rng.choice([3, 17])))]
        for v in variants:
            eng.set_handoff(v.get("handoff", -1))
            eng.set_repack(v.get("repack", -1))
            eng.set_ring(v.get("ring", 1))
            eng.set_debug_switch("VAR_RING", v.get("var_ring", -1))
            eng.set_debug_switch("VAR_RING_UNITS", v.get("units", -1))
            eng.set_tuning(waves_per_workgroup=v.get("waves", 0), max_chunk_tiles=v.get("chunk", 0))
            for want_llr in (True, False):
                got = eng.decode_batch(s, want_llr=want_llr)
                g = [x.cpu().numpy()[rows] if x is not None else None for x in got]
                tag = f"n={n} irregular={irregular} dv={dv} method={method} a={alpha} it={max_iter} p={p} B={B} {v} llr={want_llr}"
                assert np.array_equal(g[0], want[0]) and np.array_equal(g[2], want[2]) and np.array_equal(g[3].astype(bool), want[3].astype(bool)), tag
                if want_llr: assert oracle.bits_equal(g[1], want[1]), "llr " + tag
            eng.set_tuning(waves_per_workgroup=0, max_chunk_tiles=0)
        eng.close()
        n_ok += 1
    return n_ok


if __name__ == "__main__":
    print("cases passed:", run(float(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
