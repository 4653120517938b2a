#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats + PMC passes of bench.py.
#   tools/profile_bench.sh <tag> [bench args...]
# Outputs (rocpd .db files) land under gpurun_out/prof_<tag>/ ; tools/prof_parse.py turns them into the text
# summaries committed under profiles/.  PMC passes are separate runs with --kernel-trace only.
set -u
TAG=${1:-r}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS="--steps 1 --warmup 1 --cpu-sample 0 --secondary 0 --host-io 0 $*"
echo "== stats: bench.py $ARGS" > "$OUT/log.txt"
timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python bench.py $ARGS >> "$OUT/log.txt" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  echo "== pmc $grp" >> "$OUT/log.txt"
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc -- python bench.py $ARGS >> "$OUT/log.txt" 2>&1
done
python tools/prof_parse.py "$OUT" bp_decode "$OUT" > "$OUT/summary.txt" 2>&1   # also writes hbm_traffic.json, valu_clock.json
cat "$OUT/summary.txt"
