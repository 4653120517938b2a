#!/usr/bin/env python3
"""Experiment (round 3): config 5's BP stage in two phases -- K iterations for everybody, then the unconverged rows again from the start
with the full limit on large teams -- emulated with two handles.  Slower than one phase at every K (the second phase alone is the
200 us of 50 dependent iterations).  Run on an MI355X:   python tools/two_phase_c5_experiment.py"""
import os, sys, json, time
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.getcwd())
from ldpc_amd import codes
from ldpc_amd.engine import HipBpEngine
h = sp.csr_matrix(codes.bivariate_bicycle_hx()); n = h.shape[1]; p = 0.05; B = 8192
def eng(max_iter):
    e = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), max_iter, 0, 1.0)
    return e
def timed(e, s, reps=7, **kw):
    out = e.decode_batch(s, **kw)
    ks = []
    for _ in range(reps):
        out = e.decode_batch(s, out=out, asynchronous=True, **kw); torch.cuda.synchronize(); ks.append(e.last_kernel_ms())
    return float(np.median(ks)), out
e50 = eng(50)
s = e50.gen_bsc_syndromes(7, p, shot0=0, shots=B, device="cuda:0")
base, out50 = timed(e50, s)
print(json.dumps({"one phase, max_iter 50, default form": base}))
for form in (0, 1):
    e50.set_debug_switch("PS_TEAM", form); t, _ = timed(e50, s); print(json.dumps({"one phase PS_TEAM": form, "ms": t}))
e50.set_debug_switch("PS_TEAM", -1)
for K in (8, 10, 12, 16):
    eK = eng(K)
    row = {"K": K}
    for form in (0, 1):
        eK.set_debug_switch("PS_TEAM", form)
        t1, outK = timed(eK, s)
        live = (~outK[3].bool()).nonzero().flatten()
        s2 = s[live].contiguous()
        best2 = None
        for tw in (-1, 4, 8, 12, 16):
            e50.set_debug_switch("PS_TEAM", 1); e50.set_debug_switch("PS_TEAM_WAVES", tw)
            t2, out2 = timed(e50, s2)
            if best2 is None or t2 < best2[0]: best2 = (t2, tw)
        e50.set_debug_switch("PS_TEAM", -1); e50.set_debug_switch("PS_TEAM_WAVES", -1)
        row[f"phase1 form {form}"] = round(t1, 4); row[f"phase2 after form {form} (best team waves)"] = (round(best2[0], 4), best2[1]); row["live"] = int(live.numel())
        ok = bool(torch.equal(out2[0], out50[0][live]) and torch.equal(out2[2], out50[2][live]))
        row["identical"] = ok
    print(json.dumps(row), flush=True)
