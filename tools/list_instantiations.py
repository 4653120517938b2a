#!/usr/bin/env python3
"""Every kernel instantiation of libldpc_hip.so with its register budget, grouped by template, and the dispatch rule that selects it.
    make -C ldpc_amd/csrc resource-usage > /tmp/ru.txt 2>&1 ; python tools/list_instantiations.py /tmp/ru.txt > profiles/r5_kernel_instantiations.txt
(no GPU needed: hipcc's -Rpass-analysis=kernel-resource-usage remarks)"""
import re
import subprocess
import sys

RULES = {  # template -> who selects which instantiation
    "bp_decode_kernel": "host_stream.h decode_device via pick_kernel(tu_stream.hip): <METHOD, MATH, DR, DC, RING> -- RING 2 (default) / 3 for exactly (6,3)- or (8,4)-regular H "
                        "(ldpc_hip_bp_set_ring picks the depth, 0 = register variant); else the smallest (DR, DC) of (4,3) (6,3) (8,4) (8,8) (16,8) (16,16) that "
                        "bounds the heaviest row / column (heavier nodes stream through memory inside the kernel); <., ., 16, 8, 9> = the variable-degree LDS ring, on request "
                        "(VAR_RING 1).  Since round 5 a product-sum batch WITHOUT a ring variant skips this kernel unless the caller sets a hand-off threshold: it takes "
                        "bp_spread_* from its first iteration (host_stream.h: per_pass_first)",
    "bp_spread_check_kernel": "host_stream.h pick_spread: <METHOD, MATH, DR in 8/16, NT, LOOP> (NT: tiles in flight outgrow the MALL; LOOP: the slots beyond the first 32 of a compacted list)",
    "bp_spread_bit_kernel": "host_stream.h pick_spread: <METHOD, MATH, DC in 4/8, NT, LOOP>",
    "bp_spread_init_kernel": "host_stream.h: batches of <= 256 tiles, and product-sum batches without a ring variant (per-pass kernels from the first iteration)",
    "bp_edge0_kernel": "host_stream.h: initial edge values of the ring variants",
    "bp_wave_kernel": "host_onchip.h plan_wave / pick_wave: <METHOD, MATH, DR, DC, TEAM> for (4,2) (4,4) (6,3) (8,4) (8,8) and, min-sum only, (16,8); TEAM where LDS leaves "
                      "few wavefronts per CU or the batch is small",
    "bp_wave_ps_kernel": "host_onchip.h plan_wave_ps / pick_wave_ps: product-sum, <MATH, DR, DC, TEAM> for (4,2) (4,4) (6,3) (8,4) (16,8) (32,8)",
    "bp_edge_kernel": "host_onchip.h plan_edge: min-sum, rows <= 4, columns <= 2, R = ceil(4 m / 64) rounds (1 .. 16), UNIFORM = one prior for all columns",
    "bp_edge8_kernel": "host_onchip.h plan_edge8: min-sum, rows <= 8, columns <= 4: <R rounds, DC, UNIFORM>",
    "bp_small_kernel": "host_onchip.h decode_small: small codes no wavefront kernel takes (rows > 32 or columns > 8)",
    "bp_serial_kernel": "host_serial.h pick_serial: serial schedule, one wavefront per tile",
    "bp_serial_level_kernel": "host_serial.h pick_serial_level: serial schedule, level-parallel",
    "bp_serial_stream_kernel": "host_serial.h decode_serial_streamed / decode_serial_pass: serial schedule on (6,3)-shaped matrices with >= 32 bits a level: <METHOD, MATH, 6, 3, RING 1 (default) / 2>",
    "bp_serial_lane_kernel": "host_serial.h serial_lane_launch: what a streamed pass leaves (<= 2048 rows), batches of <= 256 rows: a workgroup per syndrome",
    "bp_flood_lane_kernel": "host_stream.h decode_stream_repacked, on request only (FLOOD_LANES): the rows a first pass of the flooding schedule leaves, a workgroup per syndrome",
    "bp_spread_compact_kernel": "host_stream.h: the list of parked tiles without the final ones, every 8 rounds of a second pass with few expected stragglers",
    "bp_softinfo_kernel": "host_serial.h soft_info_device", "bp_softinfo_level_kernel": "host_serial.h soft_info_device (level-parallel)",
    "bp_serial_relative_kernel": "host_serial.h decode_serial_relative: codes beyond LDS (or LDPC_HIP_REL_LDS=0)",
    "bp_relative_lds_kernel": "host_serial.h decode_serial_relative_lds: <METHOD, MATH, DRT in 4/8/16, GS, DCT>: GS = 64 lanes per syndrome and the level-by-level sweep (DCT = 2/4/8 lanes per bit >= the heaviest column) when the order is a permutation of the bits; else bit by bit, product-sum with GS = 16 where four syndromes per wavefront fit (DCT 8 unused).  The 1 - 2 spilled VGPRs of the min-sum forms (~30 of the product-sum ones) sit around the call of the out-of-line sort, once per iteration",
    "osd0_reg_kernel": "host_osd.h: OSD-0, m <= 64/128/256", "osdw_reg_kernel": "host_osd.h: OSD_E / OSD_CS, m <= 256 and n <= 511",
    "osd_big_kernel": "host_osd.h: <HIGHER, MAT_LDS> workgroup per syndrome", "osd0_kernel": "host_osd.h", "osdw_kernel": "host_osd.h",
}

txt = open(sys.argv[1]).read()
rows = []
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split(" ")[0]
    g = lambda k: int(re.search(k + r": (\d+)", b).group(1))
    rows.append((name, g("VGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
dem = subprocess.run(["c++filt"] + [r[0] for r in rows], capture_output=True, text=True).stdout.strip().split("\n")
groups = {}
for d, r in zip(dem, rows):
    d = d.replace("void ", "")
    base = d.split("<")[0].split("(")[0]
    groups.setdefault(base, []).append((d.split("(")[0], r))
print(f"# {len(rows)} kernel instantiations in libldpc_hip.so (hipcc -Rpass-analysis=kernel-resource-usage, gfx950)")
print("# columns: VGPRs, VGPRs spilled, SGPRs spilled, scratch bytes/lane, occupancy waves/SIMD, static LDS bytes")
for base in sorted(groups, key=lambda k: (-len(groups[k]), k)):
    items = groups[base]
    print(f"\n## {base}  ({len(items)})")
    if base in RULES:
        print(f"   selected by: {RULES[base]}")
    for d, r in items:
        flag = "   <-- VGPR spills" if r[2] else ""
        print(f"   {d:58s} {r[1]:4d} {r[2]:5d} {r[3]:5d} {r[4]:6d} {r[5]:3d} {r[6]:7d}{flag}")
