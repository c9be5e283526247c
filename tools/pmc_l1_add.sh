#!/bin/bash
# Runs on the GPU box: HBM traffic (FETCH_SIZE, WRITE_SIZE: one pass each) and the SQ instruction / wait counters of
# the fused level-1 Add at 2^20, kernel trace only, the program itself after `--`.
#   tools/pmc_l2_add.sh OUTDIR
set -o pipefail
OUT=${1:-gpurun_out/r06_l1_pmc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
declare -A G
G[fetch]="FETCH_SIZE"
G[write]="WRITE_SIZE"
G[insts]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
G[waits]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
G[lds]="SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
for p in fetch write insts waits lds; do
  rocprofv3 --pmc ${G[$p]} --kernel-trace --output-format csv -d "$OUT/$p" -o p -- python3 tools/l1_add_one.py 20 4 \
    > "$OUT/$p.out" 2> "$OUT/$p.err" || { echo "pass $p failed"; tail -3 "$OUT/$p.err"; [ $p = lds ] || exit 1; }
  f=$(find "$OUT/$p" -name "p_counter_collection.csv" | head -1)
  [ -f "$f" ] && { head -1 "$f" > "$OUT/${p}_k_g1_add_wire.csv"; grep "k_g1_add_wire" "$f" >> "$OUT/${p}_k_g1_add_wire.csv"; }
  rm -rf "$OUT/$p"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o p -- python3 tools/l1_add_one.py 20 8 > "$OUT/stats.out" 2> "$OUT/stats.err" || exit 1
cp $(find "$OUT/stats" -name "p_kernel_stats.csv" | head -1) "$OUT/kernel_stats.csv"; rm -rf "$OUT/stats"
python3 - "$OUT" <<'PY'
import csv, sys, collections
out = sys.argv[1]
for p in ("fetch", "write", "insts", "waits", "lds"):
    try:
        rows = list(csv.DictReader(open("%s/%s_k_g1_add_wire.csv" % (out, p))))
    except OSError:
        continue
    acc = collections.defaultdict(list)
    for r in rows:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-24s launches %d  mean %.6g" % (k, len(v), sum(v) / len(v)))
PY
