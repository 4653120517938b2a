#!/bin/bash
# PMC picture of bp_relative_lds_kernel on the surface code (serial_relative, min-sum) and BB144 (product-sum), and where its wavefronts spend their cycles
set -u
OUT=$PWD/gpurun_out/prof_serial_relative
mkdir -p "$OUT"
export TMPDIR=/tmp
cat > /tmp/rel_run.py <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from ldpc_amd import codes
from ldpc_amd.engine import HipBpEngine
which = sys.argv[1]
if which == "surface":
    h, p, it, meth, alpha = codes.rotated_surface_code_x(21), 0.05, 30, 1, 0.625
elif which == "hgp":  # state beyond LDS: the EXT form (round 6)
    h, p, it, meth, alpha = codes.hypergraph_product_hx(codes.regular_ldpc_code(n=32, dv=3, dc=4, seed=5)), 0.02, 30, 1, 0.625
else:
    h, p, it, meth, alpha = codes.bivariate_bicycle_hx(), 0.05, 50, 0, 1.0
eng = HipBpEngine(h.indptr, h.indices, h.shape[1], np.full(h.shape[1], p), it, meth, alpha)
eng.set_schedule("serial_relative")
if len(sys.argv) > 2: eng.set_debug_switch("REL_LDS", int(sys.argv[2]))
s = eng.gen_bsc_syndromes(7, p, shot0=0, shots=16384, device="cuda:0")
out = eng.decode_batch(s)
out = eng.decode_batch(s)
print(which, "kernel ms", eng.last_kernel_ms(), "mean it", float(out[2].float().mean()))
PY
for which in ${REL_WHICH:-surface bb hgp}; do
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/$which$i" -o pmc -- python /tmp/rel_run.py $which >> "$OUT/log.txt" 2>&1
done
done
python - "$OUT" > "$OUT/summary.txt" <<'PY'
import glob, os, sqlite3, sys
out = sys.argv[1]
allres = {}
for which in ("surface", "bb", "hgp"):
    res = {}
    for p in sorted(glob.glob(os.path.join(out, which + "*", "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(p).cursor()
        for name, cname, total, cnt, ns in cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection group by kernel_name, counter_name"):
            if "bp_relative_lds" in name:
                res[cname] = (total / cnt, ns / cnt / 1e6)
    print(which)
    for k, (v, ms) in sorted(res.items()):
        print(f"  {k:28s} {v:16.0f} per dispatch   ({ms:.2f} ms)")
    allres[which] = res
# what bench.py's f1_rel_* entries price their VALU bound with: wave-instructions per syndrome-iteration (16 384 syndromes a dispatch; mean
# iterations from the run's own line in log.txt), stamped with the kernel sources' fingerprint
import json, re
sys.path.insert(0, os.getcwd())
try:
    import bench
    tag = bench.kernel_sources_sha16()
except Exception:
    tag = None
its = {}
for line in open(os.path.join(out, "log.txt")):
    mm = re.match(r"(surface|bb|hgp) kernel ms [0-9.]+ mean it ([0-9.]+)", line)
    if mm:
        its[mm.group(1)] = float(mm.group(2))
doc = {"source": "tools/profile_serial_relative.sh: rocprofv3 --pmc over bp_relative_lds_kernel, 16 384 syndromes, p = 0.05 (hgp1600: 0.02)", "kernel_sources_sha16": tag}
for which, key in (("surface", "f1_rel_surface"), ("bb", "f1_rel_bb144"), ("hgp", "f1_rel_hgp1600")):
    r = allres.get(which, {})
    if which in its and "SQ_INSTS_VALU" in r and "GRBM_GUI_ACTIVE" in r:
        si = 16384.0 * its[which]
        doc[key] = {"valu_wave_insts_per_syndrome_iteration": r["SQ_INSTS_VALU"][0] / si, "salu_wave_insts_per_syndrome_iteration": r.get("SQ_INSTS_SALU", (0, 0))[0] / si,
                    "lds_wave_insts_per_syndrome_iteration": r.get("SQ_INSTS_LDS", (0, 0))[0] / si,
                    "valu_busy_frac_in_profile": r["SQ_ACTIVE_INST_VALU"][0] * 4.0 / (1024.0 * r["GRBM_GUI_ACTIVE"][0] / 8.0), "mean_iterations_in_profile": its[which]}
json.dump(doc, open(os.path.join(out, "f1_rel_valu.json"), "w"), indent=1)
PY
echo "phases (LDPC_HIP_REL_PROF=1, tools/serial_relative_phases.py; 65 536 syndromes):" >> "$OUT/summary.txt"
timeout 300 python tools/serial_relative_phases.py surface bb bbms 2>&1 | grep -v amdgpu.ids >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
