#!/usr/bin/env python3
"""Where the wavefronts of bp_serial_stream_kernel spend their cycles (a measurement build with -DLDPC_SER_PROF under tools/_dbg/;
build it in the container:  python tools/serial_stream_phases.py --build ; then on the GPU:  python tools/serial_stream_phases.py)."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "tools", "_dbg", "libldpc_hip_serprof.so")
NAMES = ["level prologue", "wait for segments", "LDS reads", "issue", "arithmetic", "scalar loads + stores", "drain", "level barrier"]

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(DBG), exist_ok=True)
    src = os.path.join(ROOT, "ldpc_amd", "csrc")
    units = ["bp_hip", "tu_stream", "tu_serial", "tu_onchip", "tu_osd"]
    flags = "-O3 -std=c++17 -ffp-contract=off -fPIC --offload-arch=gfx950 -Wno-unused-function -DLDPC_SER_PROF".split()
    objs = []
    procs = []
    for u in units:
        o = os.path.join(ROOT, "build", "csrc", u + ".o") if u != "tu_serial" else os.path.join(ROOT, "build", "csrc", "tu_serial_prof.o")
        objs.append(o)
        if u == "tu_serial":
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", "-o", o, os.path.join(src, u + ".hip")]))
    for p in procs:
        assert p.wait() == 0
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", DBG, *objs])
    print("built", DBG)
    sys.exit(0)

os.environ["LDPC_HIP_LIB"] = DBG
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
from ldpc_amd import codes  # noqa: E402
from ldpc_amd.engine import HipBpEngine  # noqa: E402

lib = C.CDLL(DBG)
h = codes.regular_ldpc_code(10000, 3, 6, seed=1)
n = h.shape[1]
FORMS = (("one pass", 0, ()), ("one pass, 8 waves", 0, (("SER_WAVES", 8),)), ("one pass, ring 2 x 8 waves", 0, (("SER_WAVES", 8), ("SER_RING", 2))))
for label, repack, switches in (FORMS[:1] if "--timeline" in sys.argv else FORMS):
    eng = HipBpEngine(h.indptr, h.indices, n, np.full(n, 0.05), 50, 0, 1.0)
    eng.set_schedule("serial")
    eng.set_serial_kernel(2)
    eng.set_repack(repack)
    for k, v in switches:
        eng.set_debug_switch(k, v)
    s = eng.gen_bsc_syndromes(7, 0.05, shot0=0, shots=65536, device="cuda:0")
    out = eng.decode_batch(s)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (8 + 4096 * 4))()
    lib.ldpc_hip_debug_serial_stream_clocks(buf, 1)
    out = eng.decode_batch(s, out=out)
    torch.cuda.synchronize()
    lib.ldpc_hip_debug_serial_stream_clocks(buf, 2)
    v = np.array(list(buf[:8]), float)
    tr = np.array(list(buf[8:]), np.uint64).reshape(4096, 4)[:1024]
    t0, t1 = tr[:, 0].astype(np.int64), tr[:, 1].astype(np.int64)
    base = t0.min()
    ev = sorted([(int(a - base), 1) for a in t0] + [(int(b - base), -1) for b in t1])
    cur = peak = 0
    area = 0
    last = 0
    for t, d in ev:
        area += cur * (t - last)
        last = t
        cur += d
        peak = max(peak, cur)
    span = int(t1.max() - base)
    hw = tr[:, 2]
    cu = (hw >> np.uint64(8)) & np.uint64(0xF)
    se = (hw >> np.uint64(13)) & np.uint64(0x7)
    sh = (hw >> np.uint64(12)) & np.uint64(0x1)
    xcc = tr[:, 3] & np.uint64(0xF)
    places = {(int(x), int(a), int(b), int(c)) for x, a, b, c in zip(xcc, se, sh, cu)}
    if "--timeline" in sys.argv:
        by = {}
        for i in range(1024):
            by.setdefault((int(xcc[i]), int(se[i]), int(sh[i]), int(cu[i])), []).append((round(int(t0[i] - base) / 1e5, 2), round(int(t1[i] - base) / 1e5, 2), i))
        for k in sorted(by)[:6]:
            print("place", k, sorted(by[k]))
        print("starts per 10 ms:", np.bincount(((t0 - base) // 1000000).astype(np.int64)).tolist())
        print("ends per 10 ms:", np.bincount(((t1 - base) // 1000000).astype(np.int64)).tolist())
    print(json.dumps({"workgroups": 1024, "span_ms_at_100MHz": span / 1e5, "mean_lifetime_ms": float((t1 - t0).mean()) / 1e5, "resident_mean": area / max(span, 1), "resident_peak": peak,
                      "distinct_xcc_se_sh_cu": len(places), "per_xcc": np.bincount(xcc.astype(np.int64), minlength=8).tolist()}), flush=True)
    print(json.dumps({"form": label, "kernel_ms": round(eng.last_kernel_ms(), 2), "share_of_wavefront_cycles": {k: round(x / v.sum(), 4) for k, x in zip(NAMES, v)},
                      "cycles_total": int(v.sum())}), flush=True)
    eng.close()
