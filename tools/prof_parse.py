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
