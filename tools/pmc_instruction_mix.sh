#!/bin/bash
# Runs on the GPU box: instruction mix, clock and wait fractions of the two saturated pairing kernels from the SQ /
# GRBM counters — the one-pairing-per-lane kernel (bench.py's headline step) and the lane-group kernel at 2^20 pairs
# (tools/quad_sweep.py) — one counter group per pass (8 SQ slots), kernel trace only, the program itself after `--`.
#   tools/pmc_instruction_mix.sh OUTDIR ; python tools/summarize_mix.py OUTDIR r03
set -o pipefail
OUT=${1:-gpurun_out/r03_mix}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
G3="SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
export QUAD_SWEEP_COUNTS=1048576 QUAD_SWEEP_KERNELS=quad
for p in 1 2 3; do
  eval "G=\$G$p"
  echo "== lane kernel, pass $p"
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/lane_p$p" -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra \
    > "$OUT/lane_p$p.json" 2> "$OUT/lane_p$p.err" || { [ $p = 3 ] && echo "pass 3 failed (a counter of that name may not exist on gfx950)" || exit 1; }
  echo "== lane-group kernel, pass $p"
  rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/quad_p$p" -o p -- python3 tools/quad_sweep.py k1024 \
    > "$OUT/quad_p$p.csv" 2> "$OUT/quad_p$p.err" || { [ $p = 3 ] && echo "pass 3 failed (a counter of that name may not exist on gfx950)" || exit 1; }
done
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
for f in "$OUT"/*_p*/p_counter_collection.csv; do
  [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep -E "k_pairing<[0-9]+, 0>|k_pairing_quad<|k_pairing_quad_wtab<" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }
done
du -sh "$OUT"
