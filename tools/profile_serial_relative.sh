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
for which in surface bb; do
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
for which in ("surface", "bb"):
    res = {}
    for p in sorted(glob.glob(os.path.join(out, which + "*", "**", "*.db"), recursive=True)):
        cur = sqlite3.connect(p).cursor()
        for name, cname, total, cnt, ns in cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end - start) from counters_collection group by kernel_name, counter_name"):
            if "bp_relative_lds" in name:
                res[cname] = (total / cnt, ns / cnt / 1e6)
    print(which)
    for k, (v, ms) in sorted(res.items()):
        print(f"  {k:28s} {v:16.0f} per dispatch   ({ms:.2f} ms)")
PY
echo "phases (LDPC_HIP_REL_PROF=1, tools/serial_relative_phases.py; 65 536 syndromes):" >> "$OUT/summary.txt"
timeout 300 python tools/serial_relative_phases.py surface bb bbms 2>&1 | grep -v amdgpu.ids >> "$OUT/summary.txt"
cat "$OUT/summary.txt"
