#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel stats + SQ/LDS counters of the on-chip kernels on BASELINE config 3.
#   tools/profile_small.sh <tag> [bench_configs args, default c3]
set -u
TAG=${1:-small}; shift || true
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
ARGS=${*:-c3}
timeout 300 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats -- python tools/bench_configs.py $ARGS > "$OUT/log.txt" 2>&1
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d "$OUT/pmc$i" -o pmc -- python tools/bench_configs.py $ARGS >> "$OUT/log.txt" 2>&1
done
python tools/prof_parse.py "$OUT" bp_ > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
