#!/bin/bash
# Same-box A/B of library builds: tools/ab_bench.sh [bench args] -- variant.so ...   ("base" = the default build)
ARGS=""
while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS="$ARGS $1"; shift; done
shift
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = "base" ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/ldpc_amd/lib/variants/$v.so; fi
  python bench.py --cpu-sample 0 --secondary 0 --host-io 0 --steps 2 --warmup 1 $ARGS | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v', round(d['value']), round(d['roofline']['frac'],4), 'kernel_ms', round(d['roofline']['kernel_ms'],1), 'persistent', round(d['roofline']['kernel_ms_persistent'],1))"
done; done
