#!/bin/bash
# Same-box A/B of BASELINE config 5's step between library builds (tools/bench_c5_step.py), interleaved REPS times, + a bit-for-bit comparison of
# one decode's outputs (decisions, log-ratio bits, iterations, convergence, OSD status) of every variant with the first one's.
#   tools/ab_c5.sh variant ...      ("base" = ldpc_amd/lib/libldpc_hip.so; other names: ldpc_amd/lib/variants/<name>.so;
#                                    "base+SWITCH" runs the default library with LDPC_HIP_<SWITCH>=1)
mkdir -p gpurun_out/ab_c5
echo "# $(date -u +%FT%TZ) $(python -c 'import torch;print(torch.cuda.get_device_name(0))' 2>/dev/null)"
run() {  # variant, extra args
  local v=$1; shift
  local lib=${v%%+*} sw=""
  [ "$lib" != "$v" ] && sw=${v#*+}
  ( if [ "$lib" = "base" ]; then unset LDPC_HIP_LIB; else export LDPC_HIP_LIB=$PWD/ldpc_amd/lib/variants/$lib.so; fi
    [ -n "$sw" ] && export LDPC_HIP_$sw=1
    python tools/bench_c5_step.py "$@" 2>&1 | tail -1 | sed "s/^/$v /" )
}
for v in "$@"; do run $v --steps 5 --rounds 1 --dump gpurun_out/ab_c5/$v.npz > /dev/null; done
python - "$@" <<'PY'
import sys, numpy as np
ref = np.load(f"gpurun_out/ab_c5/{sys.argv[1]}.npz")
for v in sys.argv[2:]:
    d = np.load(f"gpurun_out/ab_c5/{v}.npz")
    print("# outputs of", v, "vs", sys.argv[1], {k: bool(np.array_equal(ref[k], d[k])) for k in ref.files}, "rows through OSD:", int((d["status"] > 0).sum()))
PY
for rep in $(seq 1 ${REPS:-3}); do for v in "$@"; do echo -n "rep$rep "; run $v; done; done
