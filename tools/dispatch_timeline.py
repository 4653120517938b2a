#!/usr/bin/env python3
"""Per-dispatch timeline of the LAST decode in a rocprofv3 --kernel-trace run (rocpd .db): start offset, duration, kernel, grid.
    python tools/dispatch_timeline.py gpurun_out/prof_<tag>/stats [first_kernel_substring] [max_lines]"""
import glob
import sqlite3
import sys

root = sys.argv[1]
first = sys.argv[2] if len(sys.argv) > 2 else "bp_decode_kernel"
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 80
p = sorted(glob.glob(root + "/**/*.db", recursive=True))[0]
cur = sqlite3.connect(p).cursor()
rows = list(cur.execute("select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch d "
                        "join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if first in r[2]]
start = idx[-1] if idx else 0
t0 = rows[start][0]
total = {}
for r in rows[start:]:
    name = r[2].split("(")[0][:48]
    total[name] = total.get(name, 0.0) + (r[1] - r[0]) / 1e3
for r in rows[start:start + limit]:
    print(f"{(r[0] - t0) / 1e3:10.0f} us  +{(r[1] - r[0]) / 1e3:9.1f}  {r[2][:56]:56s} grid {r[3]} x {r[4]}")
print("-- totals from that dispatch on (us):")
for k, v in sorted(total.items(), key=lambda kv: -kv[1])[:12]:
    print(f"   {v:10.1f}  {k}")
print(f"   span {(rows[-1][1] - t0) / 1e3:.0f} us")
