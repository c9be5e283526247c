#!/bin/bash
# Round 6, runs on the GPU box (gpurun): the bench line, its kernel statistics and the counter passes profiles/README.md
# lists for this round.   tools/collect_profiles_r06.sh OUTDIR [STAGE]     STAGE: bench | pmc | mix | all (default)
# rocprofv3 is given the program itself (python3 bench.py ...), never a launcher; counter passes are separate from each
# other and carry no trace other than the kernel trace.
set -o pipefail
OUT=${1:-gpurun_out/r06_profiles}
STAGE=${2:-all}
mkdir -p "$OUT"
trim() {
  find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
  for f in "$OUT"/pmc_fetch_extra/fetch_counter_collection.csv "$OUT"/pmc_write_extra/write_counter_collection.csv; do
    [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep -E "k_pairing<[0-9]+, 1>|k_g1_add<|k_decode<[0-9]+, true>|k_encode<|k_gt_mul_wire<|k_g1_add_wire<" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }
  done
  for f in "$OUT"/*_p*/p_counter_collection.csv "$OUT"/cache_*/c_counter_collection.csv; do
    [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep -E "k_pairing<[0-9]+, 0>" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }
  done
  return 0
}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$STAGE" = all ] || [ "$STAGE" = bench ]; then
  echo "== bench (default line)"; python3 bench.py --steps 3 --warmup 1 > "$OUT/bench_line.json" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
  echo "== kernel stats, headline only"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -o bench -- python3 bench.py --no-live-traffic --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/bench_line_profiled.json" 2> "$OUT/prof_bench.err" || exit 1
  echo "== kernel stats, extras"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_extra" -o extra -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/extra_line_profiled.json" 2> "$OUT/prof_extra.err" || exit 1
fi
if [ "$STAGE" = all ] || [ "$STAGE" = pmc ]; then
  echo "== PMC FETCH_SIZE / WRITE_SIZE, headline"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || exit 1
  echo "== PMC FETCH_SIZE / WRITE_SIZE with the extras (Decrypt's lift at 2^16 and 2^20, the launches of an EAdd on level 1, the one of an EAdd on level 2)"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_extra" -o fetch -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch_extra.json" 2> "$OUT/pmc_fetch_extra.err" || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_extra" -o write -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write_extra.json" 2> "$OUT/pmc_write_extra.err" || exit 1
  echo "== where the headline kernel's memory requests are served (verdict r04 item 8): L1 -> L2 requests, L2 hits and misses"
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/cache_1" -o c -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/cache_1.json" 2> "$OUT/cache_1.err" || echo "cache pass 1 failed (counter names)"
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum --output-format csv -d "$OUT/cache_2" -o c -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/cache_2.json" 2> "$OUT/cache_2.err" || echo "cache pass 2 failed (counter names)"
fi
if [ "$STAGE" = all ] || [ "$STAGE" = mix ]; then
  G1="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
  G2="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"
  G3="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_MFMA_I8 GRBM_GUI_ACTIVE"
  for p in 1 2 3; do
    eval "G=\$G$p"
    echo "== lane kernel, SQ pass $p"
    rocprofv3 --pmc $G --kernel-trace --output-format csv -d "$OUT/lane_p$p" -o p -- python3 bench.py --no-live-traffic --steps 1 --warmup 0 --no-cpu-baseline --no-extra \
      > "$OUT/lane_p$p.json" 2> "$OUT/lane_p$p.err" || { [ $p = 3 ] && echo "pass 3 failed (a counter of that name may not exist on gfx950)" || exit 1; }
  done
fi
trim
du -sh "$OUT"
