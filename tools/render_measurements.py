#!/usr/bin/env python3
"""Rewrites, from the files of a round's collection under profiles/ (tools/collect_profiles.sh +
tools/summarize_profiles.py), the text that quotes them:
  * the measurement table of DESIGN.md section 6 (between the <!-- rNN-table-begin/end --> markers),
  * the rows of profiles/README.md for that round (between <!-- rNN-rows-begin/end -->).
    python tools/render_measurements.py r03"""
import csv
import json
import os
import re
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))
d = json.load(open(P("bench_line.json")))
pmc = json.load(open(P("pmc_summary.json")))
ex = d["extra"]
kern = d["roofline"]["kernel"]                                    # e.g. k_pairing<36, 0>
lift_kern = d["decrypt"]["roofline"]["kernel"]
kp = [r for r in csv.DictReader(open(P("bench_kernel_stats.csv"))) if r["Name"].startswith("void bgn::" + kern)][0]
c0 = ex["config0_512bit_128"]
cb = d["cpu_baseline"]
rv = lambda e: e["roofline_valu"]


def sweep(path):
    rows = {}
    for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):
        rows[(r["key"], int(r["count"]), r["kernel"])] = float(r["ms"])
    return rows


mid = sweep(P("mid_batch.csv")) if os.path.exists(P("mid_batch.csv")) else {}
sat = sweep(P("quad_saturated.csv")) if os.path.exists(P("quad_saturated.csv")) else {}
lift = pmc.get("decrypt_lift_k_pairing_1")

rows = [
    ("EMult (headline), 2²⁰",
     f"**{d['value']:.3g} pairings/s** ({d['ms_per_step']:.0f} ms per step; `{kern}` {d['roofline']['kernel_ms']:.1f} ms by HIP events in "
     f"`bench.py`, {float(kp['AverageNs']) / 1e6:.1f} ms average of {kp['Calls']} launches in `profiles/{tag}_bench_kernel_stats.csv`).  HBM: "
     f"{d['roofline']['achieved']:.3f} GB/s of algorithmic bytes = {d['roofline']['frac']:.2e} of 8 TB/s; PMC traffic "
     f"{pmc['hbm_bytes_per_launch']:.3g} B per launch = {pmc['traffic_over_algorithmic']:.0f} × algorithmic ≈ "
     f"{pmc['hbm_bytes_per_launch'] / d['roofline']['kernel_ms'] / 1e6:.0f} GB/s (`profiles/{tag}_pmc_summary.json`"
     + (f": FETCH_SIZE × {pmc['fetch_calibration']['factor']:.2f} + WRITE_SIZE, the factor calibrated on `k_encode`'s known read volume; the uncorrected sum "
        f"earlier rounds quoted is {pmc['hbm_bytes_per_launch_raw'] / pmc['algorithmic_bytes_per_launch']:.0f} ×" if pmc.get("fetch_calibration") else "") + "; "
     f"{pmc['FETCH_SIZE']['scratch_bytes_per_lane']} B/lane of spill scratch).  VALU: {rv(d)['mads_per_pairing'] / 1e6:.1f} M MADs per pairing → "
     f"{rv(d)['frac']:.3f} of the 4-waves/SIMD issue peak, **{rv(d)['frac_at_1_wave_per_simd']:.3f} of the 1-wave/SIMD ceiling** this "
     f"512-register kernel can reach"),
    ("Decrypt (second headline), T = 2⁴⁰, level 1, 1/16 negative, 1/4096 out of range",
     f"**{d['decrypt']['value']:.3g} decrypts/s** at 2²⁰, {ex['decrypt']['value']:.3g} at 2¹⁶; dominant kernel `{lift_kern}` (the lift) "
     f"{d['decrypt']['roofline']['kernel_ms']:.0f} ms per 2²⁰, the two walks with the full-width verification of hits "
     f"{d['decrypt']['roofline']['walk_kernels_ms']:.0f} ms; {rv(d['decrypt'])['mads_per_unit'] / 1e6:.1f} M MADs per decrypt "
     f"({d['decrypt']['products_per_unit']:.0f} products, {d['decrypt']['squarings_per_unit']:.0f} of them squarings) = "
     f"{rv(d['decrypt'])['frac_at_1_wave_per_simd']:.2f} of the 1-wave/SIMD issue ceiling, {rv(d['decrypt'])['frac']:.2f} of the chip's; HBM: "
     f"{d['decrypt']['roofline']['achieved']:.2f} GB/s of algorithmic bytes = {d['decrypt']['roofline']['frac']:.1e} of 8 TB/s"
     + (f", PMC traffic of the lift {lift['hbm_bytes_per_launch']:.3g} B per 2²⁰ = {lift['hbm_bytes_per_launch'] / (d['decrypt']['algorithmic_bytes_per_unit'] * 2**20):.1f} × algorithmic"
        if lift else "")),
    ("Decrypt, level 2, 2¹⁶", f"{ex['decrypt_l2']['value']:.3g} /s ({rv(ex['decrypt_l2'])['frac_at_1_wave_per_simd']:.2f} / {rv(ex['decrypt_l2'])['frac']:.2f} of the two issue ceilings)"),
    ("Encrypt, 2²⁰", f"{ex['encrypt']['value']:.3g} /s ({rv(ex['encrypt'])['frac_at_1_wave_per_simd']:.2f} / {rv(ex['encrypt'])['frac']:.2f} of the two issue ceilings at "
                     f"{ex['encrypt']['products_per_unit']:.0f} product-equivalents per encryption)"),
    ("EAdd level 1, 2²⁰, three launches",
     f"{ex['eadd_l1']['value']:.3g} /s = {ex['eadd_l1']['hbm']['achieved_GBps']:.0f} GB/s of wire traffic ({ex['eadd_l1']['hbm']['frac']:.3f} of HBM "
     f"peak), {rv(ex['eadd_l1'])['frac_at_1_wave_per_simd']:.2f} / {rv(ex['eadd_l1'])['frac']:.2f} of the two issue ceilings at "
     f"{ex['eadd_l1']['products_per_unit']:.1f} product-equivalents per addition"),
    ("MultPoly 16×16, %d polynomials%s" % (ex['multpoly'].get('polys', 4096), " + one AddPoly (configs[4] at its stated size)" if ex['multpoly'].get('polys', 0) >= 1 << 14 else ""),
     f"{ex['multpoly']['value']:.3g} coefficient pairs/s ({rv(ex['multpoly'])['frac_at_1_wave_per_simd']:.2f} / "
     f"{rv(ex['multpoly'])['frac']:.2f} of the two issue ceilings)" + (f", {ex['multpoly']['ms_per_step']:.0f} ms per step" if 'ms_per_step' in ex['multpoly'] else "")),
    ("configs[0]: 512-bit, 128 ciphertexts, host buffers",
     f"EMult **{c0['emult']['value']:.3g} ops/s** ({c0['emult']['wall_ms_for_128']:.2f} ms for the 128; C oracle on one host thread "
     f"{c0['emult'].get('cpu_single_thread_ops_per_s', 0):.0f}); EAdd {c0['eadd']['value']:.3g} ops/s; one Mult: {c0['emult_count1_latency_ms']:.2f} ms"),
    ("CPU baseline (C oracle on the GPU box's host cores, same inputs, outputs equal)",
     f"{cb['single_thread_pairings_per_s']:.0f} pairings/s on one thread; {cb['value']:.0f} /s with {cb['cores']} threads — the box's cgroup "
     f"grants {cb['host_cpu'].get('cgroup_cpus', '?')} CPUs (`cpu.max` {cb['host_cpu'].get('cgroup_cpu_max', '?')}) of its {cb['host_cpu']['nproc']}"),
]
if mid:
    def ms(key, n, k):
        return mid.get((key, n, k))
    rows.insert(1, ("EMult, mid-size batches at 1024 bits (`profiles/%s_mid_batch.csv`): lane-group / cooperative / lane kernel, ms" % tag,
                    "; ".join("%d pairs: %s" % (n, " / ".join("%.1f" % v if v is not None else "—" for v in (ms("k1024", n, "quad"), ms("k1024", n, "coop"), ms("k1024", n, "lane"))))
                              for n in (1024, 4096, 8192, 16384, 32768) if ms("k1024", n, "quad") is not None)))
if sat:
    q, l = sat.get(("k1024", 1 << 20, "quad")), sat.get(("k1024", 1 << 20, "lane"))
    if q and l:
        rows.insert(2, ("the lane-group kernel saturated (2²⁰ pairs) beside the lane kernel",
                        f"{(1 << 20) / q * 1e3:.3g} against {(1 << 20) / l * 1e3:.3g} pairings/s ({q:.0f} against {l:.0f} ms, `profiles/{tag}_quad_saturated.csv`): "
                        f"{q / l:.2f} × the time — it is the mid-batch kernel"))
mb = ex.get("mult_mid_batch")
if mb:
    rows.insert(2, ("the same sizes inside the bench line (`extra.mult_mid_batch`: whole `bgn_mult_batch_dev` calls, wire bytes to wire bytes, default dispatch)",
                    "; ".join("%s pairs: %.1f ms (`%s`)" % (n, v["ms"], v["kernel"]) for n, v in mb["sizes"].items())))
# ---- round 4 additions: the files exist from r04 on ----
def csv_rows(name):
    path = P(name)
    return list(csv.DictReader(l for l in open(path) if not l.startswith("#"))) if os.path.exists(path) else []


cc = csv_rows("concurrent_callers.csv")
if cc:
    def rate(op, threads, combine):
        r = [x for x in cc if x["op"] == op and int(x["threads"]) == threads and int(x["combine"]) == combine]
        return float(r[0]["calls_per_s"]) if r else None
    def lat(op, combine):
        r = [x for x in cc if x["op"] == op and int(x["threads"]) == 1 and int(x["combine"]) == combine]
        return float(r[0]["ms_per_call_per_thread"]) if r else None
    tmax = max(int(x["threads"]) for x in cc)
    parts = []
    for op, label in (("mult", "Mult"), ("decrypt_l1", "Decrypt"), ("add_l1", "Add"), ("multconst_l1_k40", "MultConst (40-bit k)")):
        if rate(op, tmax, 1):
            parts.append("%s **%.3g /s** (%.3g /s with the combiner off; one caller alone %.2f ms per call, %.2f off)" %
                         (label, rate(op, tmax, 1), rate(op, tmax, 0) or 0, lat(op, 1) or 0, lat(op, 0) or 0))
    rows.append(("the reference's call shape: %d native threads, each calling a single-element entry point on ONE context "
                 "(`profiles/%s_concurrent_callers.csv`, `tools/concurrent_callers.cpp`)" % (tmax, tag), "; ".join(parts)))
mc = csv_rows("multconst_mid_batch.csv")
if mc:
    def mcms(level, bits, count, kernel):
        r = [x for x in mc if x["key"] == "k1024" and int(x["level"]) == level and int(x["scalar_bits"]) == bits and
             int(x["count"]) == count and x["kernel"] == kernel]
        return float(r[0]["ms"]) if r else None
    cells = []
    for level in (1, 2):
        for bits in (1024, 40):
            seg = []
            for n in (1, 4096, 16384, 66000):
                q = mcms(level, bits, n, "quad") or mcms(level, bits, n, "default")
                l = mcms(level, bits, n, "lane")
                if q and l:
                    seg.append("%d: %.1f / %.1f" % (n, q, l))
            if seg:
                cells.append("level %d, %d-bit scalars — %s" % (level, bits, ", ".join(seg)))
    rows.append(("MultConst with per-element scalars by batch size, ms: lane groups (cut into lane rounds + remainder above 65536) / one "
                 "element per lane (`profiles/%s_multconst_mid_batch.csv`)" % tag, "; ".join(cells)))
# ---- round 5 additions ----
legs = []
for key, label, unit in (("encrypt", "Encrypt", "encrypts"), ("eadd_l1", "EAdd", "adds"), ("multpoly", "MultPoly", "polynomial products")):
    b = ex.get(key, {}).get("cpu_baseline")
    if b:
        legs.append("%s %.3g %s/s on %d cores (%.3g on one thread; bytes equal: %s)" % (label, b["value"], unit, b["cores"], b["single_thread_per_s"], b["matches_gpu_bit_exact"]))
if legs:
    rows.append(("CPU baselines of the secondaries (C oracle on the first items of the GPU's own batch, `cores` = min(affinity, cgroup quota))", "; ".join(legs)))
cachec = pmc.get("headline_cache_counters")
if cachec:
    rows.append(("where `%s`'s memory requests are served, per launch of 2²⁰ pairings (`profiles/%s_pmc_summary.json` `headline_cache_counters`)" % (kern, tag),
                 "L1 → L2 read requests %.3g, write requests %.3g; L2 hits %.3g / misses %.3g = **%.1f %% hit rate**; memory-side read requests %.3g, write requests %.3g" %
                 (cachec.get("TCP_TCC_READ_REQ_sum", 0), cachec.get("TCP_TCC_WRITE_REQ_sum", 0), cachec.get("TCC_HIT_sum", 0), cachec.get("TCC_MISS_sum", 0),
                  100 * cachec.get("l2_hit_rate", 0), cachec.get("TCC_EA0_RDREQ_sum", 0), cachec.get("TCC_EA0_WRREQ_sum", 0))))
l16 = pmc.get("decrypt_lift_2^16")
if l16:
    rows.append(("Decrypt at 2¹⁶ (configs[3]'s own batch), HBM-side traffic of the lift",
                 "%.3g B per launch of 65 536 lifts = %.1f × the algorithmic bytes (%.1f ms by the counter pass's timestamps, %.1f ms in the same run's bench line)" %
                 (l16["hbm_bytes_per_launch"], l16["hbm_bytes_per_launch"] / (d["decrypt"]["algorithmic_bytes_per_unit"] * 65536), l16["FETCH_SIZE"]["avg_ms"], l16["FETCH_SIZE"]["bench_line_kernel_ms"])))
mixp = P("instruction_mix.json")
if os.path.exists(mixp):
    mix = json.load(open(mixp)).get("one pairing per lane")
    if mix:
        rows.append(("instruction mix of `%s` (`profiles/%s_instruction_mix.json`)" % (kern, tag),
                     "%.1f M VALU lane-instructions per pairing (%.1f M of them the model's multiply-adds: %.0f %%) at %.0f G wave-instructions/s; wave cycles: VALU issuing %.0f %%, `SQ_WAIT_ANY` %.0f %%, `SQ_WAIT_INST_ANY` %.0f %%" %
                     (mix["valu_lane_instructions_per_pairing"] / 1e6, rv(d)["mads_per_pairing"] / 1e6, 100 * rv(d)["mads_per_pairing"] / mix["valu_lane_instructions_per_pairing"],
                      mix["chip_valu_Ginstr_per_s"], 100 * mix["frac_wave_cycles"]["SQ_ACTIVE_INST_VALU"], 100 * mix["frac_wave_cycles"]["SQ_WAIT_ANY"], 100 * mix["frac_wave_cycles"]["SQ_WAIT_INST_ANY"])))
dcb = d.get("decrypt", {}).get("cpu_baseline")
if dcb:
    rows.append(("CPU baseline of Decrypt (C oracle, same ciphertexts, plaintexts and statuses equal)",
                 f"{dcb['value']:.2f} decrypts/s with {dcb['cores']} threads on a bounded sample ({dcb['sample'].split(';')[0].split(',')[0]}); "
                 f"table build {dcb['table_setup_s']:.1f} s"))
er = ex.get("eadd_l1", {}).get("roofline")
if er and er.get("traffic"):
    rows.append(("EAdd level 1, HBM-side traffic of the call's four launches",
                 f"{er['traffic']:.3g} B per 2²⁰ additions = {er['traffic'] / er['algorithmic_bytes_per_call']:.2f} × the algorithmic bytes "
                 f"(`profiles/{tag}_pmc_summary.json` `eadd_l1`; FETCH_SIZE × {pmc.get('fetch_calibration', {}).get('factor', 1):.2f}, calibrated on "
                 f"`k_encode`'s known read volume)"))
dvt = csv_rows("decrypt_vs_table.csv")
if dvt:
    big = {int(r["table_log2"]): r for r in dvt if int(r["batch"]) == 1 << 20}
    rows.append(("Decrypt at 2²⁰ against the size of the baby-step table (`profiles/%s_decrypt_vs_table.csv`)" % tag,
                 "; ".join("2^%d entries (%.1f GB, set up in %.2f s): %.3g /s" % (k, float(r["table_GB"]), float(r["setup_s"]), float(r["decrypts_per_s"]))
                           for k, r in sorted(big.items(), reverse=True) if k in (31, 30, 29, 28, 26, 24))))
evw = csv_rows("encrypt_vs_window.csv")
if evw:
    rows.append(("Encrypt at 2²⁰ against the window width of Q's table (`profiles/%s_encrypt_vs_window.csv`)" % tag,
                 "; ".join(("2^%s entries a window, %s windows (%.1f GB): %.3g /s" % (r["q_table_index_bits"], r["q_windows"], float(r["q_table_GB"]), float(r["encrypts_per_s"])))
                           if "q_table_index_bits" in r else
                           ("%s bits (%.1f GB): %.3g /s" % (r["q_window_bits"], float(r["q_table_GB"]), float(r["encrypts_per_s"]))) for r in evw)))
table = "| | value |\n|---|---|\n" + "".join(f"| {a} | {b} |\n" for a, b in rows)


def replace_block(path, begin, end, text):
    s = open(path).read()
    s2, n = re.subn(r"(<!-- %s -->\n).*?(<!-- %s -->)" % (begin, end), lambda m: m.group(1) + text + m.group(2), s, flags=re.S)
    assert n == 1, "markers %s not found in %s" % (begin, path)
    open(path, "w").write(s2)


replace_block(os.path.join(ROOT, "DESIGN.md"), "%s-table-begin" % tag, "%s-table-end" % tag, table)

cmd_bench = "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra`; extras: `--steps 1 --warmup 0 --no-cpu-baseline`"
cmd_pmc = "`rocprofv3 --pmc FETCH_SIZE --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra` (and `WRITE_SIZE`), separate passes, no tracing"
readme_rows = [
    (f"`{tag}_bench_line.json`, `{tag}_bench_line_profiled.json`, `{tag}_bench_kernel_stats.csv`, `{tag}_extra_kernel_stats.csv`",
     f"round {int(tag[1:])} (`tools/collect_profiles{'_r05' if tag == 'r05' else ''}.sh`): `{kern}` avg {float(kp['AverageNs']) / 1e6:.1f} ms per 2^20 pairings (rocprofv3, {kp['Calls']} launches) / "
     f"{d['roofline']['kernel_ms']:.1f} ms (HIP events in bench.py): {d['value']:.3g} pairings/s; Decrypt {d['decrypt']['value']:.3g} /s (lift `{lift_kern}` "
     f"{d['decrypt']['roofline']['kernel_ms']:.0f} ms per 2^20); every secondary priced in multiply-adds against both issue ceilings", cmd_bench),
    (f"`{tag}_pmc_summary.json`, `{tag}_pmc_fetch_size_k_pairing.csv`, `{tag}_pmc_write_size_k_pairing.csv`",
     f"HBM-side traffic of `{kern}`: {pmc['hbm_bytes_per_launch']:.3g} B per launch = {pmc['traffic_over_algorithmic']:.0f} x the "
     f"{pmc['algorithmic_bytes_per_launch']:.3g} algorithmic bytes; scratch {pmc['FETCH_SIZE']['scratch_bytes_per_lane']} B per lane", cmd_pmc),
]
text = "".join("| %s | %s | %s |\n" % r for r in readme_rows)
replace_block(os.path.join(ROOT, "profiles", "README.md"), "%s-rows-begin" % tag, "%s-rows-end" % tag, text)
print(table)
print(text)
