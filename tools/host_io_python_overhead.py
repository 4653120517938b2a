"""What a `BpDecoder.decode_batch(numpy)` call costs outside the C ABI: freeing the previous 5.9 GB of results, np.empty, the profile of one call.
    python tools/host_io_python_overhead.py        (on an MI355X)"""
import sys, os, time, numpy as np, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ldpc_amd.codes import regular_ldpc_code
from ldpc_amd.engine import HipBpEngine
from ldpc_amd.bp_decoder import BpDecoder
h = regular_ldpc_code(10000, 3, 6, seed=1)
eng = HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, 0.09), 50, 0, 1.0)
B = 65536
s = eng.gen_bsc_syndromes(7, 0.09, shot0=0, shots=B, device="cuda:0").cpu().numpy()
keep = []
keep.append(eng.decode_batch(s, want_llr=True))
t0 = time.perf_counter(); keep.append(eng.decode_batch(s, want_llr=True)); t1 = time.perf_counter()
print(f"engine, results kept alive (no free inside): {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); x = keep.pop(); del x; t1 = time.perf_counter()
print(f"freeing one result set (5.9 GB): {(t1 - t0) * 1e3:.1f} ms")
t0 = time.perf_counter(); a = np.empty((B, 10000), np.float64); t1 = time.perf_counter(); print(f"np.empty 5.2 GB: {(t1 - t0) * 1e3:.2f} ms")
del a
dec = BpDecoder(h, error_rate=0.09, max_iter=50, bp_method="product_sum", input_vector_type="syndrome")
dec.decode_batch(s, want_log_prob_ratios=True)
t0 = time.perf_counter(); out = dec.decode_batch(s, want_log_prob_ratios=True); t1 = time.perf_counter()
print(f"BpDecoder.decode_batch with llr (frees the previous llr inside): {(t1 - t0) * 1e3:.1f} ms")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); out = dec.decode_batch(s, want_log_prob_ratios=True); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
