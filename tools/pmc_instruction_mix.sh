#!/bin/bash
# Runs on the GPU box: the instruction mix and the clock of k_pairing<38,0> from the SQ / GRBM counters, one
# counter group per pass (8 SQ slots), kernel trace only.  tools/pmc_instruction_mix.sh OUTDIR
set -o pipefail
OUT=${1:-gpurun_out/r02_mix}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
B="python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra"
rocprofv3 -L > "$OUT/counters_avail.txt" 2>&1 || true
echo "== pass 1: instruction counts"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$OUT/p1" -o p1 -- $B > "$OUT/p1.json" 2> "$OUT/p1.err" || exit 1
echo "== pass 2: cycles"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$OUT/p2" -o p2 -- $B > "$OUT/p2.json" 2> "$OUT/p2.err" || exit 1
echo "== pass 3: cycles by unit"
rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d "$OUT/p3" -o p3 -- $B > "$OUT/p3.json" 2> "$OUT/p3.err" || echo "pass 3 failed (a counter of that name may not exist on gfx950)"
find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
for f in "$OUT"/p*/p*_counter_collection.csv; do
  [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep "k_pairing<38, 0>" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }
done
du -sh "$OUT"
