#!/usr/bin/env python3
"""profiles/<tag>_instruction_mix_multconst.json from the passes of tools/pmc_mix_multconst.sh: per kernel the counter
sums of its 65 536-element launch, VALU lane-instructions per element, wave cycles per VALU instruction, issue / wait
fractions.    python tools/summarize_mix_multconst.py gpurun_out/r04_mix_mc r04"""
import csv
import glob
import json
import os
import re
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N = 65536
out = {"command": "tools/pmc_mix_multconst.sh (rocprofv3 --pmc <8 counters> --kernel-trace, one group per pass); 65536 elements, 1024-bit scalars, 1024-bit key",
       "kernels": {}}
for match, label in (("k_g1_mul_quad<", "level 1, lane groups"), ("k_g1_mul<", "level 1, one element per lane"),
                     ("k_gt_pow_quad_each<", "level 2, lane groups"), ("k_gt_pow<", "level 2, one element per lane")):
    ctr, meta, dur = {}, {}, None
    for p in (1, 2):
        files = glob.glob(os.path.join(src, "p%d" % p, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        rows = [r for r in csv.DictReader(open(files[0])) if re.search(r"bgn::" + re.escape(match), r["Kernel_Name"])]
        if not rows:
            continue
        span = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        best = max(rows, key=span)
        for r in rows:
            if r["Dispatch_Id"] == best["Dispatch_Id"]:
                ctr[r["Counter_Name"]] = float(r["Counter_Value"])
        meta = {"kernel": best["Kernel_Name"].split("(")[0], "grid_threads": int(best["Grid_Size"]), "workgroup": int(best["Workgroup_Size"]),
                "lds_bytes_per_workgroup": int(best["LDS_Block_Size"]), "vgpr": int(best["VGPR_Count"]), "agpr": int(best["Accum_VGPR_Count"]),
                "scratch_bytes_per_lane": int(best["Scratch_Size"])}
        if p == 1:
            dur = span(best) / 1e9
    if not ctr:
        continue
    d = dict(meta, what=label, launch_s_under_counters=dur, counters=ctr)
    if "SQ_INSTS_VALU" in ctr:
        d["valu_lane_instructions_per_element"] = ctr["SQ_INSTS_VALU"] * 64 / N
        if dur:
            d["valu_wave_instructions_per_s"] = ctr["SQ_INSTS_VALU"] / dur
    if "SQ_WAVE_CYCLES" in ctr and ctr["SQ_WAVE_CYCLES"]:
        wc = ctr["SQ_WAVE_CYCLES"]
        d["fraction_of_wave_cycles"] = {k: ctr[k] / wc for k in ("SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_LDS") if k in ctr}
    out["kernels"][match.rstrip("<")] = d
path = os.path.join(ROOT, "profiles", "%s_instruction_mix_multconst.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: {x: v[x] for x in v if x not in ("counters",)} for k, v in out["kernels"].items()}, indent=1))
