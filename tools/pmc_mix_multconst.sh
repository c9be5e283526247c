#!/bin/bash
# Runs on the GPU box: SQ / GRBM counters of the lane-group MultConst kernels (quad/quad_g1.hpp) at 65 536 elements and
# 1024-bit scalars (level 1: k_g1_mul_quad, level 2: k_gt_pow_quad_each), one counter group per pass, kernel trace
# only, the program itself after `--`.
#   tools/pmc_mix_multconst.sh OUTDIR ; python tools/summarize_mix_multconst.py OUTDIR r04
set -o pipefail
OUT=${1:-gpurun_out/r04_mix_mc}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
export MC_COUNTS=65536 MC_KBYTES=128
for p in 1 2; do
  eval "G=\$G$p"
  echo "== pass $p"
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/p$p" -o p -- python3 tools/multconst_mid_batch.py k1024 \
    > "$OUT/p$p.csv" 2> "$OUT/p$p.err" || exit 1
done
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
for f in "$OUT"/p*/p_counter_collection.csv; do
  [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep -E "k_g1_mul_quad<|k_gt_pow_quad_each<|k_g1_mul<|k_gt_pow<" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }
done
du -sh "$OUT"
