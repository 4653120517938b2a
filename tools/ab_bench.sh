#!/bin/bash
# Same-box A/B of library builds, interleaved: tools/ab_bench.sh [bench args] -- variant ...   ("base" = the default build;
# other names are ldpc_amd/lib/variants/<name>.so, e.g. the libraries of earlier rounds built from `git archive <round-end commit>`).
# REPS (default 3) rounds over the variants in the order given, so box drift shows as a trend over the rounds, not as a difference.
ARGS=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done
shift
echo "# $(date -u +%FT%TZ) $(python -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null) bench args:${ARGS:- (defaults)}; columns: variant syndromes/s frac kernel_ms persistent_ms copy_GB/s frac_of_copy"
for rep in $(seq 1 ${REPS:-3}); do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/ldpc_amd/lib/variants/$v.so; fi
  python bench.py --cpu-sample 0 --secondary 0 --host-io 0 --steps 2 --warmup 1 $ARGS 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());r=d['roofline'];c=r.get('copy_GBps_this_run');print('rep$rep', '$v', round(d['value']), round(r['frac'],4), 'kernel_ms', round(r['kernel_ms'],1), 'persistent', round(r['kernel_ms_persistent'],1), 'copy', round(c) if c else None, 'frac_of_copy', round(r['frac_of_copy'],3) if c else None)"
done; done
