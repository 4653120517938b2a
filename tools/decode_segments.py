"""Segments of the LAST decode in a rocprofv3 --kernel-trace run: consecutive dispatches of one kind merged (start .. end, busy time, count).
    python tools/decode_segments.py gpurun_out/<dir>        (profiles/r4_p050_timeline.txt)"""
import glob, sqlite3, sys
p = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[0]
cur = sqlite3.connect(p).cursor()
rows = list(cur.execute("select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
# last decode = from the second-to-last pack_syndromes (pass 1) on
packs = [i for i, r in enumerate(rows) if "pack_syndromes" in r[2]]
start = packs[-2]
t0 = rows[start][0]
seg = []
for r in rows[start:]:
    name = r[2].split("(")[0].replace("_Z", "")[:40]
    if seg and seg[-1][0] == name and "spread" not in name:
        seg[-1][1] += 1; seg[-1][2] += (r[1] - r[0]) / 1e3; seg[-1][4] = (r[1] - t0) / 1e3
    else:
        seg.append([name, 1, (r[1] - r[0]) / 1e3, (r[0] - t0) / 1e3, (r[1] - t0) / 1e3])
# merge runs of spread kernels
out = []
for s in seg:
    if "spread" in s[0] and out and out[-1][0] == "spread rounds":
        out[-1][1] += 1; out[-1][2] += s[2]; out[-1][4] = s[4]
    elif "spread" in s[0]:
        out.append(["spread rounds", 1, s[2], s[3], s[4]])
    else:
        out.append(s)
for s in out:
    print(f"{s[3]:10.0f} us .. {s[4]:10.0f}  busy {s[2]:9.1f} us  x{s[1]:4d}  {s[0]}")
