#!/usr/bin/env python3
"""Extended differential run (by hand for as long as wanted; a bounded slice with fixed seeds is collected as tests/test_gpu_fuzz_soak.py): random regular codes of 96 .. 1300 bits, random priors /
methods / iteration limits / batch sizes, through every form of the on-chip BP kernels (one wavefront or a workgroup per syndrome,
lane = node, lane = entry or lane = edge) and through the workgroup OSD kernel (blocked, one pivot per step, one staged plane; rows
outside the image included since round 3), against the CPU checker bit for bit.      python tests/fuzz_differential.py <seconds> <seed>
End of round 2: 3 242 cases over two seeds, no mismatch (gpurun, 7 minutes).  Round 3: the kernel-shape choices are handle switches
(ldpc_hip_bp_set_debug_switch), mode 6 (the lane = edge kernels) joins wherever it applies, sizes 48 .. 1300."""
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
        n = int(rng.choice([48, 96, 200, 330, 520, 800, 1300]))
        dv, dc = (3, 6) if rng.random() < 0.6 else (4, 8) if rng.random() < 0.5 else (2, 4)
        n -= n % dc
        h = sp.csr_matrix(codes.regular_ldpc_code(n, dv, dc, seed=int(rng.integers(1, 1000))))
        m = h.shape[0]
        method = "product_sum" if rng.random() < 0.4 else "minimum_sum"
        alpha = float(rng.choice([0.0, 0.7, 1.0]))
        max_iter = int(rng.integers(1, 14))
        p = float(rng.choice([0.02, 0.05, 0.09]))
        B = int(rng.choice([1, 7, 64, 200, 513, 3000, 9000]))
        probs = np.full(n, p)
        e = (rng.random((B, n)) < p).astype(np.uint8)
        s = np.asarray((h @ e.T % 2).T, dtype=np.uint8)
        if rng.random() < 0.3: s[rng.integers(0, B)] = rng.integers(0, 3, m)  # bytes > 1, out of image
        o = oracle.BpOracle(h, error_channel=probs, max_iter=max_iter, bp_method=method, ms_scaling_factor=alpha)
        want = o.decode_batch(s)
        eng = HipBpEngine(h.indptr, h.indices, n, probs, max_iter, 0 if method == "product_sum" else 1, alpha)
        for mode in (4, 5, 1, 6, -1):
            eng.set_small_code_kernel(mode)
            if mode == 1 and method == "product_sum": variants = ((), (("PS_TEAM", 0),), (("PS_TEAM", 1),))
            elif mode == 6: variants = ((), (("EDGE_STATIC_PCT", 50), ("EDGE_CHUNK", 3)))
            else: variants = ((),)
            for sw in variants:
                for k, v in sw: eng.set_debug_switch(k, v)
                got = eng.decode_batch(s)
                for k, v in sw: eng.set_debug_switch(k, -1)
                tag = f"n={n} dv={dv} {method} a={alpha} it={max_iter} B={B} mode={mode} switches={sw}"
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3]), tag
                assert oracle.bits_equal(got[1], want[1]), "llr " + tag
        # OSD through the workgroup kernel and the automatic choice (rows outside the image included: the exact second pass)
        s2 = np.asarray((h @ e.T % 2).T, dtype=np.uint8)[:min(B, 24 if rng.random() < 0.7 else 600)].copy()  # (beyond the resident teams: the BP kernel's own list of rows for OSD)
        s2[0, int(rng.integers(0, m))] ^= 1  # (outside the image where H is rank-deficient: every (2,4) and (4,8) code)
        meth, order = [(1, 0), (3, int(rng.integers(1, 12))), (2, int(rng.integers(1, 8)))][int(rng.integers(0, 3))]
        wo = o.bposd_decode_batch(s2, meth, order, want_llr=False)
        eng.set_small_code_kernel(-1)
        eng.set_osd(meth, order)
        hd = h.toarray().astype(np.int64)
        for kern, sw in ((2, ()), (2, (("OSD_UNBLOCKED", 1),)), (2, (("OSD_PLANES", 1),)), (-1, ()), (-1, (("OSD_COLLECT_AFTER", 1),)), (-1, (("OSD_NO_FLAT", 1),))):
            eng.set_osd_kernel(kern)
            for k, v in sw: eng.set_debug_switch(k, v)
            g = eng.decode_batch(s2, want_llr=False, osd=True)
            st = eng.osd_status(len(s2))
            for k, v in sw: eng.set_debug_switch(k, -1)
            tag = f"osd n={n} dv={dv} {method} method={meth} order={order} rows={len(s2)} kern={kern} switches={sw}"
            assert np.array_equal(g[0], wo[0]), tag
            # the status array (round 6: written by the BP kernel / the register OSD-0 kernel themselves): 0 = BP converged, 1 = the returned x
            # solves H x = s, 2 = it does not (s outside the image: x solves the reference's pivot rows only)
            solves = np.all((g[0].astype(np.int64) @ hd.T) % 2 == (s2 != 0), axis=1)
            want_st = np.where(g[3] != 0, 0, np.where(solves, 1, 2)).astype(np.uint8)
            assert np.array_equal(st, want_st), "status " + tag
        n_ok += 1
    return n_ok


if __name__ == "__main__":
    print("cases passed:", run(float(sys.argv[1]) if len(sys.argv) > 1 else 120, int(sys.argv[2]) if len(sys.argv) > 2 else 1))
