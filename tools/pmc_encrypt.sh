#!/bin/bash
# Where Encrypt's chain kernel spends its cycles: SQ counters of k_g1_fixed_chain (and k_g1_add) over the Encrypt calls of
# bench.py's extras, one pass.   tools/pmc_encrypt.sh OUTDIR
set -o pipefail
OUT=${1:-gpurun_out/r05_pmc_encrypt}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
  -d "$OUT/p1" -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/p1.json" 2> "$OUT/p1.err" || exit 1
f="$OUT/p1/p_counter_collection.csv"
head -1 "$f" > "$f.tmp"; grep -E "k_g1_fixed_chain<|k_g1_add<|k_gt_pow<|k_bsgs_search<|k_fixedpair_build_batch<" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(lambda: collections.defaultdict(float))
dur = collections.defaultdict(float); n = collections.defaultdict(int)
seen = set()
for r in rows:
    k = r["Kernel_Name"].split("(")[0].replace("void bgn::", "")
    # the largest launches of each kernel only (grid = 65536 lanes)
    if int(r["Grid_Size"]) != 65536: continue
    by[k][r["Counter_Name"]] += float(r["Counter_Value"])
    key = (k, r["Dispatch_Id"])
    if key not in seen:
        seen.add(key); n[k] += 1; dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("kernel,launches,avg_ms,valu_wave_instr_per_launch,cycles_per_valu_instr,frac_valu_issue,frac_wait_any,frac_wait_inst,vmem_rd_per_valu")
for k, c in by.items():
    wc = c["SQ_WAVE_CYCLES"] or 1
    print("%s,%d,%.3f,%.4g,%.3f,%.3f,%.3f,%.3f,%.5f" % (k, n[k], dur[k] / n[k], c["SQ_INSTS_VALU"] / n[k], wc / (c["SQ_INSTS_VALU"] or 1), c["SQ_ACTIVE_INST_VALU"] / wc,
          c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_INSTS_VMEM_RD"] / (c["SQ_INSTS_VALU"] or 1)))
PY
