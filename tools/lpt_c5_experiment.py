#!/usr/bin/env python3
"""Experiment (round 3): does handing the on-chip product-sum kernel its heaviest syndromes FIRST shorten config 5's BP stage?
The 7 % that never converge run 50 iterations (200 us of dependent work); a workgroup that pulls one of them late ends late.
Run on an MI355X:   python tools/lpt_c5_experiment.py"""
import os, sys, json
import numpy as np, scipy.sparse as sp, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ldpc_amd import codes
from ldpc_amd.engine import HipBpEngine
h = sp.csr_matrix(codes.bivariate_bicycle_hx()); n = h.shape[1]; p = 0.05
for B in (8192, 65536):
    e = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), 50, 0, 1.0)
    s = e.gen_bsc_syndromes(7, p, shot0=0, shots=B, device="cuda:0")
    def timed(s_):
        out = e.decode_batch(s_)
        ks = []
        for _ in range(9):
            out = e.decode_batch(s_, out=out, asynchronous=True); torch.cuda.synchronize(); ks.append(e.last_kernel_ms())
        return float(np.median(ks)), out
    t0, out = timed(s)
    w = s.sum(dim=1, dtype=torch.int32)
    conv = out[3].bool()
    order = torch.argsort(w, descending=True, stable=True)
    t1, _ = timed(s[order].contiguous())
    oracle_order = torch.argsort(out[2], descending=True, stable=True)  # (the unattainable optimum: longest decode first)
    t2, _ = timed(s[oracle_order].contiguous())
    hard = (~conv).nonzero().flatten()
    rank = torch.empty_like(order); rank[order] = torch.arange(B, device=order.device)
    frac_in_first_quarter = float((rank[hard] < B // 4).float().mean())
    print(json.dumps({"batch": B, "kernel_ms": {"as generated": round(t0, 4), "heaviest syndrome first": round(t1, 4), "longest decode first (oracle)": round(t2, 4)},
                      "unconverged": int(hard.numel()), "share of them among the heaviest quarter": round(frac_in_first_quarter, 3),
                      "mean syndrome weight": {"converged": float(w[conv].float().mean()), "unconverged": float(w[~conv].float().mean())}}), flush=True)
    e.close()
