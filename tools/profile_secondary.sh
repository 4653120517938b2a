#!/bin/bash
# Run on the GPU box (through gpurun): VALU instructions per entry-iteration of the on-chip product-sum kernel on BASELINE
# config 5 (BP stage), for bench.py's `secondary[c5].frac` (FP64 VALU issue bound).  Writes gpurun_out/prof_<tag>/secondary_valu.json
#   tools/profile_secondary.sh <tag>
set -u
TAG=${1:-secondary}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python tools/bench_configs.py c5bp > "$OUT/log.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d "$OUT/pmc1" -o pmc -- python tools/bench_configs.py c5bp >> "$OUT/log.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -d "$OUT/pmc2" -o pmc -- python tools/bench_configs.py c5bp >> "$OUT/log.txt" 2>&1
python - "$OUT" <<'PY'
import glob, json, os, sqlite3, sys
out = sys.argv[1]
sys.path.insert(0, os.getcwd())
try:
    import bench
    build_tag = bench.kernel_sources_sha16()
except Exception:
    build_tag = None
cfg = None
for line in open(os.path.join(out, "log.txt")):
    if line.startswith('{"config"'):
        cfg = json.loads(line)
        break
res = {}
for p in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(p).cursor()
    for name, cname, total, cnt, ns in cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection group by kernel_name, counter_name"):
        if "bp_wave_ps_kernel" in name:
            res[cname] = {"per_dispatch": total / cnt, "dispatches": cnt, "ms_per_dispatch": ns / cnt / 1e6}
print(json.dumps(res, indent=1))
if cfg and "SQ_INSTS_VALU" in res:
    entries_iters = 432.0 * cfg["mean_iterations"] * cfg["batch"]
    per = res["SQ_INSTS_VALU"]["per_dispatch"] * 64.0 / entries_iters
    d = {"c5": {"kernel": "bp_wave_ps_kernel<0, 6, 3>", "kernel_sources_sha16": build_tag, "valu_insts_per_entry_iteration": per,
                "sq_insts_valu_per_dispatch": res["SQ_INSTS_VALU"]["per_dispatch"], "batch": cfg["batch"], "mean_iterations": cfg["mean_iterations"],
                "note": "SQ_INSTS_VALU (wave-instructions) x 64 lanes / (432 entries x iterations executed x batch): padding lanes and the "
                        "prefix/suffix recomputation of the lane = entry layout are included, i.e. this is issue work per useful entry-iteration"}}
    if "GRBM_GUI_ACTIVE" in res:
        d["c5"]["clock_ghz"] = res["GRBM_GUI_ACTIVE"]["per_dispatch"] / 8.0 / (res["GRBM_GUI_ACTIVE"]["ms_per_dispatch"] * 1e6)
        d["c5"]["valu_issue_frac_measured"] = res["SQ_INSTS_VALU"]["per_dispatch"] * 4.0 / (1024.0 * res["GRBM_GUI_ACTIVE"]["per_dispatch"] / 8.0)
    json.dump(d, open(os.path.join(out, "secondary_valu.json"), "w"), indent=1)
    print(json.dumps(d, indent=1))
PY
python tools/prof_parse.py "$OUT" bp_wave > "$OUT/summary.txt" 2>&1
tail -20 "$OUT/summary.txt"
