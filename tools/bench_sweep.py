#!/usr/bin/env python3
"""Run bench.py for several argument sets (one per command-line argument, quoted) and print one compact line each.

    python tools/bench_sweep.py "--bp-method minimum_sum --waves 8" "--math fast"
"""
import json
import subprocess
import sys

for argset in sys.argv[1:]:
    cmd = [sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--cpu-sample", "0", "--host-io", "0"] + argset.split()
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(f"[{argset}] FAILED rc={r.returncode}: {r.stderr[-400:]}")
        continue
    d = json.loads(line[-1])
    print(f"[{argset}] {d['value']:.0f} synd/s  {d['ms_per_step']:.1f} ms/step  kernel {d['roofline']['kernel_ms']:.1f} ms  "
          f"frac {d['roofline']['frac']:.3f}  achieved {d['roofline']['achieved']:.0f} GB/s  iters {d['config']['mean_iterations']:.2f}", flush=True)
