#!/usr/bin/env python3
"""profiles/<tag>_instruction_mix.json from the passes of tools/pmc_instruction_mix.sh: per kernel the counter sums of
its 2^20-pairing launch (the longest launch of that kernel in the pass), instructions per pairing, wave cycles per
VALU instruction, issue / wait fractions, LDS bank-conflict share.
    python tools/summarize_mix.py gpurun_out/r03_mix r03"""
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = 1 << 20
out = {"command": "tools/pmc_instruction_mix.sh (rocprofv3 --pmc <8 counters> --kernel-trace, one group per pass)", "kernels": {}}
for stem, match, label in (("lane", "k_pairing<", "one pairing per lane"), ("quad", "k_pairing_quad<", "lane-group kernel, launch 1 (Miller loop)")):
    ctr, dur, meta, table = {}, {}, {}, {}
    for p in (1, 2, 3):
        files = glob.glob(os.path.join(src, "%s_p%d" % (stem, p), "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        rows = [r for r in csv.DictReader(open(files[0])) if match in r["Kernel_Name"] and ", 2>" not in r["Kernel_Name"]]
        if not rows:
            continue
        span = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        longest = max(span(r) for r in rows)
        # one dispatch = one Dispatch_Id; take the dispatch with the longest duration (the lane kernel's 2^20 launch; the
        # lane-group kernel runs a 2^20 batch in pieces of 196 608 pairings: one full piece)
        best = max(rows, key=span)["Dispatch_Id"]
        for r in rows:
            if r["Dispatch_Id"] == best:
                ctr[r["Counter_Name"]] = float(r["Counter_Value"])
                meta = {"kernel": r["Kernel_Name"].split("(")[0], "grid_threads": int(r["Grid_Size"]), "workgroup": int(r["Workgroup_Size"]),
                        "lds_bytes_per_workgroup": int(r["LDS_Block_Size"]), "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]),
                        "scratch_bytes_per_lane": int(r["Scratch_Size"])}
        dur["p%d" % p] = longest / 1e9
        if stem == "quad":
            # the table launches of the width-w loop belong to the Miller loop: their instructions, per pairing of their own
            # grids, are added to the count below
            trows = [r for r in csv.DictReader(open(files[0])) if "k_pairing_quad_wtab<" in r["Kernel_Name"]]
            for name in {r["Counter_Name"] for r in trows}:
                num = sum(float(r["Counter_Value"]) for r in trows if r["Counter_Name"] == name)
                den = sum(int(r["Grid_Size"]) // 16 for r in trows if r["Counter_Name"] == name and ", 1>" in r["Kernel_Name"])
                if den:
                    table[name] = num / den
    if not ctr:
        continue
    d = {}
    t = dur.get("p1") or list(dur.values())[0]
    PAIRS = (1 << 20) if stem == "lane" else meta["grid_threads"] // 16      # pairings of the dispatch the counters are of
    if "SQ_INSTS_VALU" in ctr:
        d["valu_instructions_per_pairing"] = ctr["SQ_INSTS_VALU"] * 64 / PAIRS / (64 if stem == "lane" else 16) * (1 if stem == "lane" else 1)
        # wave instructions: a lane-kernel wave carries 64 pairings, a lane-group wave 4
        per_wave_pairings = 64 if stem == "lane" else 4
        d["valu_wave_instructions_per_pairing"] = ctr["SQ_INSTS_VALU"] / PAIRS
        d["valu_lane_instructions_per_pairing"] = ctr["SQ_INSTS_VALU"] * 64 / PAIRS
        d.pop("valu_instructions_per_pairing")
        d["pairings_per_wave"] = per_wave_pairings
        d["pairings_in_the_dispatch"] = PAIRS
        if table.get("SQ_INSTS_VALU"):
            d["valu_lane_instructions_per_pairing_table_launches"] = table["SQ_INSTS_VALU"] * 64
            d["valu_lane_instructions_per_pairing_with_table_launches"] = d["valu_lane_instructions_per_pairing"] + table["SQ_INSTS_VALU"] * 64
        d["chip_valu_Ginstr_per_s"] = ctr["SQ_INSTS_VALU"] / t / 1e9
        for k in ("SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if k in ctr:
                d[k.lower().replace("sq_insts_", "") + "_per_valu"] = ctr[k] / ctr["SQ_INSTS_VALU"]
    if "SQ_WAVE_CYCLES" in ctr:
        wc = ctr["SQ_WAVE_CYCLES"]
        d["frac_wave_cycles"] = {k: ctr[k] / wc for k in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
                                                          "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if k in ctr}
        if "SQ_INSTS_VALU" in ctr:
            d["wave_cycles_per_valu_instruction"] = wc / ctr["SQ_INSTS_VALU"]
    if "SQ_LDS_BANK_CONFLICT" in ctr and "SQ_LDS_IDX_ACTIVE" in ctr and ctr["SQ_LDS_IDX_ACTIVE"]:
        d["lds_bank_conflict_share_of_lds_active"] = ctr["SQ_LDS_BANK_CONFLICT"] / ctr["SQ_LDS_IDX_ACTIVE"]
    out["kernels"][label] = dict(meta, counters=ctr, duration_s=dur, derived=d)
path = os.path.join(ROOT, "profiles", "%s_instruction_mix.json" % tag)
json.dump(out, open(path, "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out["kernels"].items()}, indent=1))
