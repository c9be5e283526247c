#!/usr/bin/env python3
"""Round 6: the measurement table of DESIGN.md section 6 and the status table of README.md, generated from the tracked
files under profiles/ (r06_bench_line.json, r06_pmc_summary.json, r06_bench_kernel_stats.csv, r06_l2_add_pmc.txt) so
that the prose cannot drift from the numbers.   python tools/render_r06.py [bench_line.json]"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def last_json_line(path):
    txt = open(path).read()
    try:
        return json.loads(txt)
    except ValueError:
        return json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])


line = last_json_line(sys.argv[1] if len(sys.argv) > 1 else os.path.join(P, "r06_bench_line.json"))
pmc_path = os.path.join(P, "r06_pmc_summary.json")
pmc = json.load(open(pmc_path)) if os.path.exists(pmc_path) else {}
ex = line["extra"]


def e3(v):
    m, e = ("%.2e" % v).split("e")
    return "%s × 10^%d" % (m, int(e))


def frac(entry, key="frac"):
    return entry.get("roofline_valu", {}).get(key)


def cpu(entry):
    cb = entry.get("cpu_baseline") or {}
    if not cb:
        return "—"
    return "%s %s on %d cores (%s on one thread), bytes equal: %s" % (e3(cb["value"]), cb["unit"], cb["cores"],
                                                                       e3(cb.get("single_thread_per_s", 0)) if cb.get("single_thread_per_s") else "—",
                                                                       cb.get("matches_gpu_bit_exact"))


rows = []
r = line["roofline"]
stats_avg = None
sp = os.path.join(P, "r06_bench_kernel_stats.csv")
if os.path.exists(sp):
    for row in csv.DictReader(open(sp)):
        if re.match(r"void bgn::k_pairing<\d+, 0>", row["Name"]):
            stats_avg = (float(row["AverageNs"]) / 1e6, int(row["Calls"]))
traffic = r.get("traffic")
rows.append(("EMult (headline), 2^20, `%s`" % r["kernel"],
             "**%s pairings/s** (%.0f ms per step; kernel %.1f ms by HIP events%s).  HBM: %.3f GB/s of algorithmic bytes = %.2e of "
             "8 TB/s%s.  VALU: %.1f M multiply-adds per pairing, %.3f of the four-wave issue ceiling, **%.3f of the one-wave ceiling** "
             "this 512-register kernel runs at.  CPU (C oracle, %d cores): %.0f pairings/s, sample bit-exact: %s" %
             (e3(line["value"]), line["ms_per_step"], r["kernel_ms"],
              ", %.1f ms average of %d launches in `profiles/r06_bench_kernel_stats.csv`" % stats_avg if stats_avg else "",
              r["achieved"], r["frac"],
              "; PMC traffic %.3g B per launch = %.0f × algorithmic" % (traffic, traffic / (r["algorithmic_bytes_per_pairing"] * line["config"]["batch_per_gpu"])) if traffic else "",
              line["roofline_valu"]["mads_per_pairing"] / 1e6, line["roofline_valu"]["frac"], line["roofline_valu"]["frac_at_1_wave_per_simd"],
              line["cpu_baseline"]["cores"], line["cpu_baseline"]["value"], line["cpu_baseline"].get("matches_gpu_bit_exact"))))
d = line["decrypt"]
rows.append(("Decrypt (second headline), T = 2^40, level 1, 2^20, 1/16 negative, 1/4096 out of range",
             "**%s decrypts/s** (2^16: %s); lift `%s` %.0f ms + walks %.0f ms; %.3f / %.3f of the four- / one-wave ceilings; plaintexts and "
             "statuses exact: %s.  CPU: %s" %
             (e3(d["value"]), e3(ex["decrypt"]["value"]), d["roofline"]["kernel"], d["roofline"]["kernel_ms"], d["roofline"]["walk_kernels_ms"],
              frac(d), frac(d, "frac_at_1_wave_per_simd"), d["plaintexts_and_statuses_exact"], cpu(d))))
l2 = ex["eadd_l2"]
t2 = l2["roofline"].get("traffic")
rows.append(("EAdd level 2, 2^20, one launch of `k_gt_mul_wire`",
             "**%s adds/s**; kernel %.3f ms → %.0f GB/s of algorithmic bytes = **%.3f of HBM peak**%s; %d multiply-adds per add = %.3f of the "
             "two-wave issue ceiling (%.3f of the four-wave one).  CPU: %s" %
             (e3(l2["value"]), l2["roofline"]["kernel_ms"], l2["roofline"]["achieved"], l2["roofline"]["frac"],
              "; PMC traffic %.3g B per launch = %.2f × algorithmic" % (t2, t2 / l2["roofline"]["algorithmic_bytes_per_launch"]) if t2 else "",
              l2["roofline_valu"]["mads_per_unit"], l2["roofline_valu"]["frac_at_2_waves_per_simd"], l2["roofline_valu"]["frac"], cpu(l2))))
l1 = ex["eadd_l1"]
t1 = l1["roofline"].get("traffic")
rows.append(("EAdd level 1, 2^20, one launch of `%s`" % l1["kernel"],
             "%s adds/s (call %.3f ms) = %.0f GB/s of algorithmic bytes (%.3f of HBM peak)%s; %.3f / %.3f of the four- / one-wave "
             "ceilings at %.1f product-equivalents per addition.  CPU: %s" %
             (e3(l1["value"]), l1["roofline"]["call_ms"], l1["roofline"]["achieved"], l1["roofline"]["frac"],
              "; PMC traffic %.3g B per call = %.2f × algorithmic" % (t1, t1 / l1["roofline"]["algorithmic_bytes_per_call"]) if t1 else "",
              frac(l1), frac(l1, "frac_at_1_wave_per_simd"), l1["products_per_unit"], cpu(l1))))
if "neg_l1" in ex:
    ng = ex["neg_l1"]
    rows.append(("Neg level 1, 2^20, one launch of `%s`" % ng["kernel"],
                 "**%s negs/s**; kernel %.3f ms → %.0f GB/s of algorithmic bytes = **%.3f of HBM peak** (no field product: the "
                 "one HBM-bound operation of the path).  CPU: %s" %
                 (e3(ng["value"]), ng["roofline"]["kernel_ms"], ng["roofline"]["achieved"], ng["roofline"]["frac"], cpu(ng))))
enc = ex["encrypt"]
rows.append(("Encrypt, 2^20", "%s encrypts/s; %.3f / %.3f of the two ceilings.  CPU: %s" %
             (e3(enc["value"]), frac(enc), frac(enc, "frac_at_1_wave_per_simd"), cpu(enc))))
for key in ("multconst_l1_40b", "multconst_l1_1024b", "multconst_l2_40b", "multconst_l2_1024b"):
    if key in ex:
        m = ex[key]
        rows.append(("MultConst level %d, %d-bit scalars, 2^16 (`%s`)" % (m["level"], m["scalar_bits"], m["kernel"]),
                     "%s /s (call %.2f ms); %.3f / %.3f of the two ceilings at %.0f product-equivalents.  CPU: %s" %
                     (e3(m["value"]), m["call_ms"], frac(m), frac(m, "frac_at_1_wave_per_simd"), m["products_per_unit"], cpu(m))))
mp = ex["multpoly"]
rows.append(("MultPoly 16 × 16, 2^14 polynomials + one AddPoly (configs[4]), on a context that holds decryption tables",
             "%s coefficient pairs/s (%.0f ms per step); %.3f / %.3f of the two ceilings; the context holds %.0f GB after the call.  CPU: %s" %
             (e3(mp["value"]), mp["ms_per_step"], frac(mp), frac(mp, "frac_at_1_wave_per_simd"),
              mp.get("context_memory_bytes_after_the_call", 0) / 1e9, cpu(mp))))
d2 = ex["decrypt_l2"]
rows.append(("Decrypt level 2, 2^16", "%s /s; %.3f / %.3f of the two ceilings" % (e3(d2["value"]), frac(d2), frac(d2, "frac_at_1_wave_per_simd"))))
c0 = ex["config0_512bit_128"]
rows.append(("configs[0]: 512-bit, 128 ciphertexts, host buffers",
             "EMult %s ops/s (C oracle on one thread: %.0f), EAdd %s ops/s; one Mult %.2f ms" %
             (e3(c0["emult"]["value"]), c0["emult"].get("cpu_single_thread_ops_per_s", 0), e3(c0["eadd"]["value"]), c0["emult_count1_latency_ms"])))
if "mult_mid_batch" in ex:
    mb = ex["mult_mid_batch"]["sizes"]
    rows.append(("Mult at mid-size batches (whole calls, default dispatch)",
                 "; ".join("%s pairs: %.1f ms (`%s`)" % (k, v["ms"], v["kernel"]) for k, v in mb.items())))
if "fetch_calibration" in pmc:
    fc = pmc["fetch_calibration"]
    rows.append(("FETCH_SIZE calibration (`%s`)" % fc["kernel"], "known read volume ÷ counter = %.3f" % fc["factor"]))


def wrap(cells):
    import textwrap
    out = []
    for a, b in cells:
        out.extend(textwrap.wrap("* **%s** — %s" % (a, b), width=118, subsequent_indent="  ", break_long_words=False,
                                 break_on_hyphens=False))
    return "\n".join(out)


def replace_block(path, begin, end, body):
    s = open(path).read()
    a, b = s.index("<!-- %s -->" % begin), s.index("<!-- %s -->" % end)
    s = s[:a + len("<!-- %s -->" % begin)] + "\n" + body + "\n" + s[b:]
    open(path, "w").write(s)


table = wrap(rows)
replace_block(os.path.join(ROOT, "DESIGN.md"), "r06-table-begin", "r06-table-end", table)
print(table)
