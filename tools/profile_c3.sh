#!/bin/bash
# Run on the GPU box (through gpurun): issue-slot accounting of the on-chip min-sum kernel that runs BASELINE config 3
# (bp_edge_kernel; the lane = node kernel if LDPC_C3_MODE=5 is exported) -- VALU / SALU / LDS instructions per syndrome-iteration, effective clock,
# LDS bank-conflict share.  Writes gpurun_out/prof_<tag>/secondary_c3.json (bench.py's `secondary[c3]` bounds) and summary.txt.
#   tools/profile_c3.sh <tag>
set -u
TAG=${1:-c3}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python tools/bench_configs.py c3p05 > "$OUT/log.txt" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc -- python tools/bench_configs.py c3p05 >> "$OUT/log.txt" 2>&1
done
python tools/prof_parse.py "$OUT" bp_ > "$OUT/summary.txt" 2>&1
python - "$OUT" <<'PY' >> "$OUT/summary.txt" 2>&1
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
res, kern = {}, None
for p in sorted(glob.glob(os.path.join(out, "pmc*", "**", "*.db"), recursive=True)):
    cur = sqlite3.connect(p).cursor()
    for name, cname, total, cnt, ns in cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection group by kernel_name, counter_name"):
        if "bp_edge_kernel" in name or "bp_wave_kernel" in name:
            kern = name.split("(")[0].replace("void ", "")
            res[cname] = {"per_dispatch": total / cnt, "ms_per_dispatch": ns / cnt / 1e6}
if cfg and "SQ_INSTS_VALU" in res and "GRBM_GUI_ACTIVE" in res:
    synd_iters = cfg["mean_iterations"] * cfg["batch"]
    cyc = res["GRBM_GUI_ACTIVE"]["per_dispatch"] / 8.0  # per XCD
    d = {"c3": {"kernel": kern, "kernel_sources_sha16": build_tag, "batch": cfg["batch"], "mean_iterations": cfg["mean_iterations"],
                "valu_insts_per_syndrome_iteration": res["SQ_INSTS_VALU"]["per_dispatch"] / synd_iters,
                "salu_insts_per_syndrome_iteration": res["SQ_INSTS_SALU"]["per_dispatch"] / synd_iters,
                "lds_insts_per_syndrome_iteration": res["SQ_INSTS_LDS"]["per_dispatch"] / synd_iters,
                "clock_ghz": cyc / (res["GRBM_GUI_ACTIVE"]["ms_per_dispatch"] * 1e6),
                "valu_issue_frac_measured": res["SQ_INSTS_VALU"]["per_dispatch"] * 4.0 / (1024.0 * cyc),
                "salu_issue_frac_measured": res["SQ_INSTS_SALU"]["per_dispatch"] * 4.0 / (1024.0 * cyc),
                "lds_array_busy_frac_measured": res["SQ_LDS_IDX_ACTIVE"]["per_dispatch"] / (256.0 * cyc) if "SQ_LDS_IDX_ACTIVE" in res else None,
                "lds_bank_conflict_share": res["SQ_LDS_BANK_CONFLICT"]["per_dispatch"] / res["SQ_LDS_IDX_ACTIVE"]["per_dispatch"] if "SQ_LDS_IDX_ACTIVE" in res else None,
                "kernel_ms_in_the_clock_pass": res["GRBM_GUI_ACTIVE"]["ms_per_dispatch"],
                "note": "wave-instructions (SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS) per syndrome-iteration executed; a SIMD issues one vector and one "
                        "scalar instruction per 4-cycle turn: issue fraction = instructions x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); LDS: "
                        "SQ_LDS_IDX_ACTIVE / (256 CUs x cycles), conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"}}
    json.dump(d, open(os.path.join(out, "secondary_c3.json"), "w"), indent=1)
    print("# secondary_c3.json")
    print(json.dumps(d, indent=1))
PY
cat "$OUT/summary.txt"
