#!/usr/bin/env python3
"""Cycles per phase of osdw_reg_kernel (higher-order OSD with the elimination in registers).

Builds a second copy of the library with -DLDPC_HIP_OSD_CLOCKS (tools/_dbg/, git-ignored; build it in the container:
`python tools/osd_phase_clocks.py --build`), loads THAT copy, decodes one batch of the [[400,16,6]] HGP code (or BB144
with --bb) with OSD_CS and prints the share of each phase.  A profiling aid, not part of the product path."""
import argparse
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
DBG = os.path.join(ROOT, "tools", "_dbg", "libldpc_hip.so")
PHASES = ["load rows", "sort columns", "eliminate", "number non-pivot columns", "gather reduced rows", "weigh candidates", "pick + write"]
# (the workgroup kernel reports: slot 1 copy + sort, 2 elimination, 3 numbering, 4 gather, 5 weighing)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--bb", action="store_true")
    ap.add_argument("--hgp1600", action="store_true", help="the [[1600,64]] code of bench_configs.py: workgroup kernel, H in HBM")
    ap.add_argument("--osd0", action="store_true", help="OSD-0 through osd0_reg_kernel (with --bb and --shots 8192: BASELINE config 5's OSD stage)")
    ap.add_argument("--order", type=int, default=10)
    ap.add_argument("--shots", type=int, default=65536, help="batch size (few rows through OSD = the latency regime: little contention between wavefronts)")
    ap.add_argument("--mask", type=lambda v: int(v, 0), default=0xffff,
                    help="workgroup kernel: the probes that are live (bit = slot); every probe serialises, so a few at a time perturb "
                         "least -- the time between two live probes goes to the later one")
    args = ap.parse_args()
    global DBG
    if args.mask != 0xffff:
        DBG = DBG[:-3] + "_%x.so" % args.mask
    if args.build:
        os.makedirs(os.path.dirname(DBG), exist_ok=True)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "ldpc_amd", "csrc"), "OUT=" + DBG, "OBJDIR=" + os.path.join(ROOT, "build", "csrc_osd_clocks_%x" % args.mask),
                               "CFLAGS=-O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wall "
                               "-Wno-unused-function -DLDPC_HIP_OSD_CLOCKS -DLDPC_HIP_OSD_CLOCK_MASK=0x%x" % args.mask])
        return
    import ldpc_amd._lib as lib
    lib.LIB_PATH = DBG
    from ldpc_amd.engine import HipBpEngine
    from ldpc_amd import codes
    import scipy.sparse as sp
    import torch
    if args.hgp1600:
        h = codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5))
        p, it, method, alpha = 0.02, 30, 1, 0.625
    elif args.bb:
        h, p, it, method, alpha = codes.bivariate_bicycle_hx(), 0.05, 50, 0, 1.0
    else:
        z = np.load(os.path.join(ROOT, "tests", "golden", "qcodes_400_16_6_ms_par_osd0.npz"))
        h = sp.csr_matrix((np.ones(len(z["col_idx"]), np.uint8), z["col_idx"], z["row_ptr"]), shape=(int(z["m"]), int(z["n"])))
        p, it, method, alpha = 0.02, 30, 1, 0.625
    n = h.shape[1]
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, p), it, method, alpha)
    s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=args.shots, device="cuda:0")
    eng.set_osd(1 if args.osd0 else 3, 0 if args.osd0 else args.order)
    out = eng.decode_batch(s, osd=True)
    torch.cuda.synchronize()
    dll = lib.load()
    buf = (C.c_ulonglong * 16)()
    dll.ldpc_hip_debug_osd_clocks(buf, 1)
    eng.decode_batch(s, out=out, osd=True)
    torch.cuda.synchronize()
    dll.ldpc_hip_debug_osd_clocks(buf, 0)
    rows = int((out[3].cpu().numpy() == 0).sum())
    tot = sum(buf[:7])
    print(f"{rows} rows through OSD; cycles per row (s_memtime, 100 MHz-class counter units):")
    names = PHASES
    if args.hgp1600:  # the workgroup kernel's slots (osd_big_kernel); 0 / 6 / 7 split its blocked elimination
        names = ["elimination: the block's plane", "sort + working copy", "(columns, pivots: 1e6, 1e9 digits)", "number non-pivot columns", "T planes",
                 "weigh candidates", "elimination: block pivots", "elimination: rest of the update", "update: list of the rows", "update: table build",
                 "update: barrier", "update: items", "update: closing barrier"]
        tot = sum(buf[:14]) - buf[2]
    if args.osd0:
        print(f"  (osd0_reg_kernel / osd0_flat_kernel: load, sort, eliminate, 'number non-pivot columns' = the rows built with their columns in sorted order (flat kernel), "
              f"'pick + write' = status + decisions out; pivots per row {buf[8] / max(buf[9], 1):.1f} over {buf[9]} rows)")
    for name, c in zip(names, buf[:16]):
        print(f"  {name:52s} {c / max(rows, 1):10.0f}  {100.0 * c / max(tot, 1):5.1f} %")


if __name__ == "__main__":
    main()
