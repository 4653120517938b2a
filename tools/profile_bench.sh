#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
#   tools/profile_bench.sh <tag> [bench args...]
# Outputs land under gpurun_out/prof_<tag>/ ; the summaries worth keeping are copied to profiles/ by hand.
# PMC passes are separate runs with --kernel-trace only (never combined with sys/hip/hsa traces).
set -u
TAG=${1:-r}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --cpu-sample 0 $*"
cd "$PWD"
echo "== stats" > "$OUT/log.txt"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python bench.py $ARGS >> "$OUT/log.txt" 2>&1
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $grp | tr ' ' '_' | cut -c1-40)
  echo "== pmc $grp" >> "$OUT/log.txt"
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc_$name" -o pmc -- python bench.py $ARGS >> "$OUT/log.txt" 2>&1
done
# compact summaries
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
with open(os.path.join(out, "summary.txt"), "w") as f:
    for p in sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True)):
        f.write(f"# {os.path.relpath(p, out)}\n" + open(p).read() + "\n")
    for p in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for row in csv.DictReader(open(p)):
            k = row.get("Kernel_Name", "?")[:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); cnt[(k, row["Counter_Name"])] += 1
        f.write(f"# {os.path.relpath(p, out)}\n")
        for k, d in agg.items():
            for c, v in d.items():
                f.write(f"{k:60s} {c:28s} total={v:.6g} dispatches={cnt[(k, c)]}\n")
print(open(os.path.join(out, "summary.txt")).read()[:6000])
PY
