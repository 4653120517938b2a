"""One decode() through the C ABI alone (1000 calls): total, the part between the timing events, and the cost of set_params.
    python tools/decode_latency_breakdown.py [keep-timing-events]        (on an MI355X)"""
import sys, os, time, ctypes as C, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ldpc_amd import codes, _lib
from ldpc_amd.engine import HipBpEngine
lib = _lib.load()
for name, h, p, mi in (("hamming5", codes.hamming_code(5), 0.05, 20), ("BB144", codes.bivariate_bicycle_hx(), 0.05, 50)):
    import scipy.sparse as sp
    h = sp.csr_matrix(h); m, n = h.shape
    rng = np.random.default_rng(0)
    e = (rng.random(n) < p).astype(np.uint8)
    s = np.ascontiguousarray((h @ e % 2).astype(np.uint8)[None, :])
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), mi, 0, 1.0)
    if len(sys.argv) > 1: eng.set_debug_switch('TIME_SMALL_CALLS', 1)
    dec = np.empty((1, n), np.uint8); llr = np.empty((1, n)); it = np.empty(1, np.int32); cv = np.empty(1, np.uint8)
    def raw(want_llr=True):
        return lib.ldpc_hip_bp_decode_batch(eng._h, s.ctypes.data, 1, dec.ctypes.data, llr.ctypes.data if want_llr else None, it.ctypes.data, cv.ctypes.data)
    for tag, f in (("C ABI, llr", lambda: raw(True)), ("C ABI, no llr", lambda: raw(False)), ("engine.decode_batch numpy", lambda: eng.decode_batch(s))):
        f()
        t0 = time.perf_counter()
        for _ in range(1000): f()
        print(f"{name:9s} {tag:28s} {(time.perf_counter() - t0) / 1000 * 1e6:7.1f} us  kernel {eng.last_kernel_ms() * 1e3:6.1f} us  iters {int(it[0])}")
    # pieces of the host path
    t0 = time.perf_counter()
    for _ in range(1000): lib.ldpc_hip_bp_set_params(eng._h, mi, 0, C.c_double(1.0))
    print(f"{name:9s} set_params {(time.perf_counter() - t0) / 1000 * 1e6:6.2f} us")
