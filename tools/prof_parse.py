#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs under a directory: kernel stats, per-kernel counter sums, and -- for the
streaming BP kernels -- HBM traffic, effective clock and VALU issue fraction per decode.

    python tools/prof_parse.py gpurun_out/prof_<tag> [kernel-substring] [out-dir]

With an out-dir, writes hbm_traffic.json and valu_clock.json there, keyed with the workload read from the bench.py line in
<dir>/log.txt (bench.py copies them into `roofline.traffic` / `roofline.second_bound` when the workload matches).
"""
import glob
import json
import os
import sqlite3
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "bp_decode"
outdir = sys.argv[3] if len(sys.argv) > 3 else None
XCDS, SIMDS = 8, 1024

dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
for p in dbs:
    cur = sqlite3.connect(p).cursor()
    rel = os.path.relpath(p, root)
    try:
        rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    except sqlite3.Error:
        rows = []
    has_pmc = cur.execute("select count(*) from counters_collection").fetchone()[0] if rows is not None else 0
    if not has_pmc:
        print(f"# {rel}: kernel stats (name, calls, total_ns, avg_ns, pct)")
        for r in rows[:8]:
            print(f"  {r[0][:70]:70s} {r[1]:5d} {r[2]:14.0f} {r[3]:14.0f} {r[4]:6.2f}")
    else:
        print(f"# {rel}: counters summed over dispatches")
        for r in cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                             "group by kernel_name, counter_name order by kernel_name, counter_name"):
            if filt in r[0]:
                print(f"  {r[0][:44]:44s} {r[1]:28s} {r[2]:.6g}  (dispatches {r[3]})")

# ---- per decode: the dominant kernel (bp_decode_kernel) plus the per-pass kernels that finish its last tiles ----------
dom = "bp_decode" if filt.startswith("bp_") else filt
sums = {}      # counter -> {"dom": total over the dominant kernel, "all": dominant + bp_spread_*}
dom_ns = {}    # counter -> duration of the dominant kernel's dispatches in the SAME pass (clock = cycles / that)
n_dec = 0
for p in dbs:
    cur = sqlite3.connect(p).cursor()
    try:
        q = cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection "
                        "group by kernel_name, counter_name")
        for name, cname, total, cnt, ns in q:
            if not (dom in name or "bp_spread" in name):
                continue
            s = sums.setdefault(cname, {"dom": 0.0, "all": 0.0})
            s["all"] += total
            if dom in name:
                s["dom"] += total
                dom_ns[cname] = dom_ns.get(cname, 0.0) + ns
                n_dec = cnt
    except sqlite3.Error:
        pass

bench = None  # the bench.py line of the stats run: which workload these counters belong to
try:
    for line in open(os.path.join(root, "log.txt")):
        if line.startswith('{"metric"'):
            bench = json.loads(line)
            break
except OSError:
    pass
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import bench as _bench
    build_tag = _bench.kernel_sources_sha16()
except Exception:
    build_tag = None
key = {}
if bench:
    c = bench["config"]
    key = {"batch_per_gpu": c["batch_per_gpu"], "p": c["p"], "max_iter": c["max_iter"], "n": 10000 if "n=10000" in c["workload"] else None,
           "math": c.get("device_math", "libm_exact"), "mean_iterations": c["mean_iterations"], "kernel_sources_sha16": build_tag}

if n_dec and "FETCH_SIZE" in sums and "WRITE_SIZE" in sums:
    # corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B
    # request on wide coalesced reads -> x2; both counters are in KiB
    fetch, write = sums["FETCH_SIZE"]["all"] / n_dec, sums["WRITE_SIZE"]["all"] / n_dec
    traffic = (2.0 * fetch + write) * 1024.0
    print("# HBM traffic per launch (bytes) = (2*FETCH_SIZE + WRITE_SIZE) * 1024 =", f"{traffic:.6g}",
          f"(FETCH_SIZE {fetch:.6g} KiB, WRITE_SIZE {write:.6g} KiB per decode; all BP kernels of a decode)")
    if outdir:
        with open(os.path.join(outdir, "hbm_traffic.json"), "w") as f:
            json.dump({"kernel": dom + " + bp_spread_* (all BP kernels of one decode)", "hbm_bytes_per_launch": traffic,
                       "fetch_size_kib": fetch, "write_size_kib": write,
                       "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE x1; both KiB", "source": root, **key}, f, indent=1)

if n_dec and "GRBM_GUI_ACTIVE" in sums and "SQ_INSTS_VALU" in sums:
    # effective clock of the dominant kernel: GRBM_GUI_ACTIVE is summed over the 8 XCDs; duration from the same pass
    cycles_per_xcd = sums["GRBM_GUI_ACTIVE"]["dom"] / XCDS
    clock_ghz = cycles_per_xcd / dom_ns["GRBM_GUI_ACTIVE"]
    valu_dom, valu_all = sums["SQ_INSTS_VALU"]["dom"] / n_dec, sums["SQ_INSTS_VALU"]["all"] / n_dec
    # a wave-instruction occupies its SIMD for 4 cycles (64 lanes over 16-wide FP64 / 32-bit pipes); SIMD-cycles on offer
    # while the kernel ran = 1024 SIMDs x the cycles the chip clocked
    issue = valu_dom * 4.0 / (SIMDS * cycles_per_xcd / n_dec)
    out = {"kernel": dom, "clock_ghz": clock_ghz, "valu_issue_frac": issue, "valu_wave_insts_per_decode_dominant_kernel": valu_dom,
           "valu_wave_insts_per_decode_all_bp_kernels": valu_all, "grbm_gui_active_per_xcd_per_decode": cycles_per_xcd / n_dec,
           "dominant_kernel_ms_in_that_pass": dom_ns["GRBM_GUI_ACTIVE"] / n_dec / 1e6,
           "formulae": "clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the same pass; issue = SQ_INSTS_VALU x 4 cycles / (1024 SIMDs x cycles)",
           "source": root, **key}
    # What the hardware says about the same thing: SQ_ACTIVE_INST_VALU = quad-cycles the vector units spent executing (a wave64 instruction
    # holds its 16-lane SIMD for four cycles whatever its width; v_rcp_f64 and friends longer), against the quad-cycles the 1024 SIMDs offered
    if "SQ_ACTIVE_INST_VALU" in sums and "GRBM_GUI_ACTIVE" in sums:
        out["valu_busy_frac"] = sums["SQ_ACTIVE_INST_VALU"]["dom"] * 4.0 / (SIMDS * cycles_per_xcd)
        out["valu_busy_formula"] = "SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), dominant kernel"
    f64 = sum(sums[k]["all"] for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64") if k in sums) / n_dec
    if bench:
        c = bench["config"]
        wave_edge_iters = 30000.0 * c["mean_iterations"] * c["batch_per_gpu"] / 64.0  # E x iterations x 64-syndrome tiles
        out["valu_insts_per_edge_iteration"] = valu_all / wave_edge_iters
        if f64 > 0:
            out["fp64_arith_insts_per_edge_iteration"] = f64 / wave_edge_iters  # add / mul / fma F64 (the rest: selects, moves, compares, conversions, integer field work)
    print(f"# dominant kernel: effective clock {clock_ghz:.3f} GHz, VALU issue {issue:.3f} of the SIMD cycles on offer, "
          f"{out.get('valu_insts_per_edge_iteration', float('nan')):.1f} VALU instructions per edge-iteration (all BP kernels)")
    if outdir:
        with open(os.path.join(outdir, "valu_clock.json"), "w") as f:
            json.dump(out, f, indent=1)
