#!/bin/bash
# Run on the GPU box (through gpurun): kernel stats + PMC passes of the streamed serial decode in its item form (bp_serial_var_kernel.h) on the
# (4,8)-regular or the irregular n = 10 000 code (tools/bench_serial_stream.py --code <code> --forms one), and of serial_relative on the
# [[1600,64]] hypergraph-product code (tools/bench_relative.py --forms one).
#   tools/profile_serial_var.sh <tag> <ldpc48|irregular|hgp1600>
# Outputs land under gpurun_out/var_<tag>_<code>/ ; summary.txt is what goes to profiles/.
set -u
TAG=${1:-r}; CODE=${2:-ldpc48}
OUT=$PWD/gpurun_out/var_${TAG}_$CODE
mkdir -p "$OUT"
export TMPDIR=/tmp
if [ "$CODE" = "hgp1600" ]; then CMD="python tools/bench_relative.py --forms one --steps 1"; PAT="%bp_relative%"
else CMD="python tools/bench_serial_stream.py --code $CODE --forms one --steps 1"; PAT="%bp_serial_%"; fi
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- $CMD > "$OUT/log.txt" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "== pmc $grp" >> "$OUT/log.txt"
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc -- $CMD >> "$OUT/log.txt" 2>&1
done
{
  echo "# $CMD   ($(date -u +%FT%TZ))"
  grep '^{' "$OUT/log.txt" | head -1 | cut -c1-600
  echo "# kernel stats (rocprofv3 --kernel-trace --stats): name, calls, total ns, average ns, percent"
  python - "$OUT" "$PAT" <<'PY'
import glob, os, sqlite3, sys
root, pat = sys.argv[1], sys.argv[2]
for p in sorted(glob.glob(os.path.join(root, "stats", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(p).cursor()
    for r in cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()[:10]:
        print(f"  {r[0][:90]:90s} {r[1]:5d} {r[2]:14.0f} {r[3]:14.0f} {r[4]:6.2f}")
print("# counters per dispatch (the dispatches of the LAST decode of each PMC pass: the timed one); FETCH_SIZE / WRITE_SIZE in KiB as the counters give them,")
print("# HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)")
for p in sorted(glob.glob(os.path.join(root, "pmc*", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(p).cursor()
    rows = cur.execute(f"select kernel_name, counter_name, value, start, end, grid_size from counters_collection where kernel_name like '{pat}' order by start").fetchall()
    if not rows:
        continue
    starts = sorted({r[3] for r in rows})
    half = starts[len(starts) // 2:] if len(starts) > 1 else starts
    for st in half:
        sel = [r for r in rows if r[3] == st]
        name = sel[0][0].split("(")[0][:48]
        agg = {}
        for r in sel:
            agg[r[1]] = agg.get(r[1], 0.0) + r[2]
        print(f"{name:48s} grid {sel[0][5]:8d} {(sel[0][4] - sel[0][3]) / 1e6:9.3f} ms  " + "  ".join(f"{k}={v:.6g}" for k, v in sorted(agg.items())))
PY
} > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
