"""Timeline of the LAST streamed serial decode in a rocprofv3 --kernel-trace run (every dispatch from its serial_edge0_kernel on):
    rocprofv3 --kernel-trace -d gpurun_out/ser_tl -o tl -- python tools/bench_serial_stream.py --forms one --steps 1
    python tools/serial_stream_timeline.py gpurun_out/ser_tl"""
import glob, sqlite3, sys
p = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[0]
cur = sqlite3.connect(p).cursor()
rows = list(cur.execute("select d.start, d.end, s.kernel_name, d.grid_size_x, d.grid_size_y, d.workgroup_size_x from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if "serial_edge0" in r[2]]
start = marks[-1]
t0 = rows[start][0]
prev_end = t0
for r in rows[start:]:
    name = r[2].split("(")[0]
    print(f"{(r[0] - t0) / 1e3:10.0f} us .. {(r[1] - t0) / 1e3:10.0f}  busy {(r[1] - r[0]) / 1e3:9.1f} us  gap before {(r[0] - prev_end) / 1e3:7.1f}  grid {r[3] // max(r[5], 1)}x{r[4]} wg {r[5]}  {name[:90]}")
    prev_end = r[1]
