"""The pinned-chunk host path at several chunk sizes, with and without log-ratios, the HOST_PIPE_TIMING breakdown, and the device-resident rate of one chunk.
    python tools/host_io_chunks.py        (on an MI355X)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ldpc_amd.codes import regular_ldpc_code
from ldpc_amd.engine import HipBpEngine
h = regular_ldpc_code(10000, 3, 6, seed=1)
eng = HipBpEngine(h.indptr, h.indices, 10000, np.full(10000, 0.09), 50, 0, 1.0)
B = 65536
s = eng.gen_bsc_syndromes(7, 0.09, shot0=0, shots=B, device="cuda:0").cpu().numpy()
eng.set_debug_switch('HOST_PIPE_TIMING', 1)
for rows in (2816, 8192):
    eng.set_debug_switch("HOST_CHUNK_ROWS", rows)
    for want in (True, False):
        eng.decode_batch(s, want_llr=want)
        t0 = time.perf_counter(); eng.decode_batch(s, want_llr=want); dt = time.perf_counter() - t0
        print(f"chunk rows {rows:6d} llr {want!s:5s}: {dt*1e3:8.1f} ms  {B/dt:9.0f} syndromes/s", flush=True)
# where does the time go at the default chunk: device-resident decode of one chunk
import torch
for rows in (2816, 8192, 16384):
    sd = torch.from_numpy(s[:rows]).cuda()
    eng.decode_batch(sd, want_llr=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.decode_batch(sd, want_llr=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"device-resident {rows} rows: {dt*1e3:.1f} ms  {rows/dt:.0f} syndromes/s")
# host copy rate into a fresh array
a = np.empty((2816, 10000), np.float64); src = np.random.rand(2816, 10000)
t0 = time.perf_counter(); a[:] = src; print("numpy copy 225 MB into fresh array", round(225e6 / (time.perf_counter() - t0) / 1e9, 2), "GB/s")
