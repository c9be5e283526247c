#!/usr/bin/env python3
"""Rewrites the measurement table of DESIGN.md section 6 (between the r02-table markers) from
profiles/r02_bench_line.json, profiles/r02_bench_kernel_stats.csv and profiles/r02_pmc_summary.json."""
import csv
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r02_bench_line.json")))
pmc = json.load(open(os.path.join(ROOT, "profiles", "r02_pmc_summary.json")))
ex = d["extra"]
kp = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r02_bench_kernel_stats.csv")))
      if r["Name"].startswith("void bgn::k_pairing<38, 0>")][0]
c0 = ex["config0_512bit_128"]
cb = d["cpu_baseline"]
rows = [
    ("EMult (headline), 2²⁰",
     f"**{d['value']:.3g} pairings/s** ({d['ms_per_step']:.0f} ms per step; `k_pairing<38, 0>` {d['roofline']['kernel_ms']:.1f} ms by HIP events in "
     f"`bench.py`, {float(kp['AverageNs']) / 1e6:.1f} ms average of {kp['Calls']} launches in `profiles/r02_bench_kernel_stats.csv`).  HBM: "
     f"{d['roofline']['achieved']:.3f} GB/s of algorithmic bytes = {d['roofline']['frac']:.2e} of 8 TB/s; PMC traffic "
     f"{pmc['hbm_bytes_per_launch']:.3g} B per launch = {pmc['traffic_over_algorithmic']:.0f} × algorithmic ≈ "
     f"{pmc['hbm_bytes_per_launch'] / d['roofline']['kernel_ms'] / 1e6:.0f} GB/s (`profiles/r02_pmc_summary.json`: per-step re-reads of the "
     f"operand coordinates, the per-pairing window table and {pmc['FETCH_SIZE']['scratch_bytes_per_lane']} B/lane of spill scratch "
     f"cycling through the 4 MB L2s).  VALU: {d['roofline_valu']['mads_per_pairing'] / 1e6:.1f} M MADs per pairing → "
     f"{d['roofline_valu']['frac']:.3f} of the 4-waves/SIMD issue peak, **{d['roofline_valu']['frac_at_1_wave_per_simd']:.3f} of the "
     f"1-wave/SIMD ceiling** this 512-register kernel can reach"),
    ("Decrypt (second headline), T = 2⁴⁰, level 1, 1/16 negative, 1/4096 out of range",
     f"**{d['decrypt']['value']:.3g} decrypts/s** at 2²⁰, {ex['decrypt']['value']:.3g} at 2¹⁶; dominant kernel `k_pairing<38, 1>` (the lift) "
     f"{d['decrypt']['roofline']['kernel_ms']:.0f} ms per 2²⁰, the two walks with the full-width verification of hits "
     f"{d['decrypt']['roofline']['walk_kernels_ms']:.0f} ms; {ex['decrypt_2^20']['products_per_unit']:.0f} products per decrypt = "
     f"{ex['decrypt_2^20']['frac_of_product_ceiling']:.2f} of the product ceiling (8.15 × 10⁹ /s, `profiles/ubench_fp_rates_r01.txt`; the count "
     f"prices squarings as products, hence a fraction near or above 1)"
     + (f"; HBM: {d['decrypt']['roofline']['achieved']:.2f} GB/s of algorithmic bytes = {d['decrypt']['roofline']['frac']:.1e} of 8 TB/s, PMC traffic of "
        f"the lift {pmc['decrypt_lift_k_pairing_38_1']['hbm_bytes_per_launch']:.3g} B per 2²⁰ = "
        f"{pmc['decrypt_lift_k_pairing_38_1']['hbm_bytes_per_launch'] / (274 * 2**20):.1f} × algorithmic (its SoA result and 636 B/lane of scratch)"
        if 'decrypt_lift_k_pairing_38_1' in pmc else "")),
    ("Decrypt, level 2, 2¹⁶", f"{ex['decrypt_l2']['value']:.3g} /s"),
    ("Encrypt, 2²⁰", f"{ex['encrypt']['value']:.3g} /s (22-bit windows for Q since round 2; {ex['encrypt']['frac_of_product_ceiling']:.2f} of the "
                     f"product ceiling at the 20-bit count)"),
    ("EAdd level 1, 2%s, three launches" % {19: "¹⁹", 20: "²⁰"}.get(ex['eadd_l1']['batch'].bit_length() - 1, "^?"),
     f"{ex['eadd_l1']['value']:.3g} /s = {ex['eadd_l1']['hbm']['achieved_GBps']:.0f} GB/s of wire traffic ({ex['eadd_l1']['hbm']['frac']:.3f} of HBM "
     f"peak), {ex['eadd_l1']['frac_of_product_ceiling']:.2f} of the product ceiling at {ex['eadd_l1']['products_per_unit']:.1f} product-equivalents "
     f"per addition (7 + one shared inversion per run of {max(1, ex['eadd_l1']['batch'] // 65536)}); by batch size: `profiles/r02_eadd_sweep.csv`"),
    ("MultPoly 16×16, 4096 polynomials", f"{ex['multpoly']['value']:.3g} coefficient pairs/s"),
    ("configs[0]: 512-bit, 128 ciphertexts, host buffers",
     f"EMult **{c0['emult']['value']:.3g} ops/s** ({c0['emult']['wall_ms_for_128']:.2f} ms for the 128; 4.6 × 10³ in round 1; C oracle on one "
     f"host thread {c0['emult'].get('cpu_single_thread_ops_per_s', 0):.0f}); EAdd {c0['eadd']['value']:.3g} ops/s; one Mult: "
     f"{c0['emult_count1_latency_ms']:.2f} ms"),
    ("CPU baseline (C oracle on the GPU box's host cores, same inputs, outputs equal)",
     f"{cb['single_thread_pairings_per_s']:.0f} pairings/s on one thread; {cb['value']:.0f} /s with {cb['cores']} threads — the box's cgroup "
     f"grants {cb['host_cpu'].get('cgroup_cpus', '?')} CPUs (`cpu.max` {cb['host_cpu'].get('cgroup_cpu_max', '?')}) of its {cb['host_cpu']['nproc']}"),
]
table = "| | value |\n|---|---|\n" + "".join(f"| {a} | {b} |\n" for a, b in rows)
path = os.path.join(ROOT, "DESIGN.md")
s = open(path).read()
s2, n = re.subn(r"(<!-- r02-table-begin -->\n).*?(<!-- r02-table-end -->)", lambda m: m.group(1) + table + m.group(2), s, flags=re.S)
assert n == 1, "markers not found"
open(path, "w").write(s2)
print(table)
