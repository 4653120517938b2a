#!/usr/bin/env python3
"""Where bp_wave_ps_kernel's wavefront 0 of every workgroup spends its cycles, phase by phase (a -DLDPC_WPS_PROF build of tu_onchip.hip linked as
ldpc_amd/lib/variants/wps_prof.so: the kernel adds up shader cycles per phase, the library prints and clears them after every decode).
    hipcc ... -DLDPC_WPS_PROF -c tu_onchip.hip ; link as the Makefile does ; python tools/wave_ps_phases.py [team_waves ...]
Config 5's BP stage (BB [[144,12,12]], product-sum 50 iterations, B = 8192, p = 0.05)."""
import json, os, re, subprocess, sys
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(here, "ldpc_amd", "lib", "variants", "wps_prof.so")
names = ["checkA", "checkB", "bitA", "bitB+synd", "close", "setup/out", "pull"]
for tw in (sys.argv[1:] or ["default"]):
    env = dict(os.environ, LDPC_HIP_LIB=lib)
    if tw != "default":
        env["LDPC_HIP_PS_TEAM_WAVES"] = tw
    r = subprocess.run([sys.executable, os.path.join(here, "tools", "bench_configs.py"), "c5bp"], env=env, capture_output=True, text=True)
    lines = [l for l in r.stderr.splitlines() if l.startswith("[wps_prof]")]
    if not lines:
        print("no [wps_prof] line: is the library a -DLDPC_WPS_PROF build?", r.stderr[-400:])
        continue
    l = lines[-1]
    v = {k: int(x) for k, x in re.findall(r"(checkA|checkB|bitA|bitB\+synd|close|setup/out|pull|iterations|syndromes) (\d+)", l)}
    head = re.search(r"team (\d+) waves (\d+) groups (\d+) batch (\d+)", l).groups()
    total = sum(v[k] for k in names)
    print(f"team_waves {tw}: team {head[0]} waves {head[1]} workgroups {head[2]} batch {head[3]}; per team-iteration (cycles of wavefront 0): " +
          ", ".join(f"{k} {v[k] / max(v['iterations'], 1):.0f}" for k in names[:5]) +
          f"; per syndrome: setup/out {v['setup/out'] / max(v['syndromes'], 1):.0f}, pull {v['pull'] / max(v['syndromes'], 1):.0f}; shares: " +
          ", ".join(f"{k} {100.0 * v[k] / total:.1f} %" for k in names) + f"; iterations {v['iterations']} syndromes {v['syndromes']}")
    for line in r.stdout.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            print(f"    bp_kernel_ms {d['bp_kernel_ms']:.4f} (with the stamps)")
