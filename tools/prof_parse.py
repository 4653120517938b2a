#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (.db) outputs under a directory: kernel stats and per-kernel counter sums.

    python tools/prof_parse.py gpurun_out/prof_<tag> [kernel-substring]
"""
import glob
import os
import sqlite3
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "bp_decode"
for p in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    con = sqlite3.connect(p)
    cur = con.cursor()
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


# HBM traffic per launch of the dominant kernel, corrected as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
# prescribes for gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads -> x2; both counters are in KiB.
fetch = write = None
nd = 0
dom = "bp_decode" if filt.startswith("bp_") else filt  # the kernel whose dispatch count = number of decodes
for p in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(p).cursor()
    try:
        for name, cname, total, cnt in cur.execute(
                "select kernel_name, counter_name, sum(value), count(*) from counters_collection group by kernel_name, counter_name"):
            # bytes of one decode: the dominant kernel's dispatches plus every per-pass kernel that finishes its tiles
            if (dom in name or "bp_spread" in name) and cname == "FETCH_SIZE":
                fetch = (fetch or 0.0) + total
                if dom in name:
                    nd = cnt
            if (dom in name or "bp_spread" in name) and cname == "WRITE_SIZE":
                write = (write or 0.0) + total
    except sqlite3.Error:
        pass
if fetch is not None and write is not None and nd:
    fetch, write = fetch / nd, write / nd
    import json
    traffic = (2.0 * fetch + write) * 1024.0
    print("# HBM traffic per launch (bytes) = (2*FETCH_SIZE + WRITE_SIZE) * 1024 =", f"{traffic:.6g}",
          f"(FETCH_SIZE {fetch:.6g} KiB, WRITE_SIZE {write:.6g} KiB per dispatch)")
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as f:
            json.dump({"kernel": dom, "hbm_bytes_per_launch": traffic, "fetch_size_kib": fetch, "write_size_kib": write,
                       "correction": "FETCH_SIZE x2 (gfx950, wide coalesced reads), WRITE_SIZE x1; both KiB", "source": root}, f, indent=1)
