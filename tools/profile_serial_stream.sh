#!/bin/bash
# Run on the GPU box (through gpurun): kernel trace + PMC passes of the streamed serial decode of the headline code (tools/bench_serial_stream.py --forms one).
#   tools/profile_serial_stream.sh <tag>
# Outputs land under gpurun_out/ser_<tag>/ ; the text summary (timeline of the last decode + counters of bp_serial_stream_kernel) is what goes to profiles/.
set -u
TAG=${1:-r}
OUT=$PWD/gpurun_out/ser_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
CMD="python tools/bench_serial_stream.py --forms one --steps 1"
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- $CMD > "$OUT/log.txt" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INST_LEVEL_SMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "== pmc $grp" >> "$OUT/log.txt"
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc -- $CMD >> "$OUT/log.txt" 2>&1
done
{
  echo "# timeline of the last decode (tools/serial_stream_timeline.py)"
  python tools/serial_stream_timeline.py "$OUT/stats" | grep -v "rocclr_copyBuffer.kd" | cut -c1-175
  echo
  echo "# counters, per dispatch of bp_serial_stream_kernel / bp_serial_lane_kernel (last decode of each PMC pass)"
  python - "$OUT" <<'PY'
import glob, os, sqlite3, sys
root = sys.argv[1]
for p in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(p).cursor()
    rows = cur.execute("select kernel_name, counter_name, value, start, end, grid_size from counters_collection where kernel_name like '%bp_serial_%' order by start").fetchall()
    if not rows:
        continue
    # the last decode = the last three dispatches (first pass, second pass, lanes)
    starts = sorted({r[3] for r in rows})[-3:]
    for st in starts:
        sel = [r for r in rows if r[3] == st]
        name = sel[0][0].split("(")[0][:40]
        agg = {}
        for r in sel:
            agg[r[1]] = agg.get(r[1], 0.0) + r[2]
        print(f"{name:40s} grid {sel[0][5]:8d} {(sel[0][4] - sel[0][3]) / 1e6:9.3f} ms  " + "  ".join(f"{k}={v:.5g}" for k, v in sorted(agg.items())))
PY
} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
