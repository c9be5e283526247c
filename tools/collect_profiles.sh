#!/bin/bash
# Runs on the GPU box (gpurun): the measurements profiles/README.md lists for this round.
#   tools/collect_profiles.sh OUTDIR
# rocprofv3 is given the program itself (python3 bench.py ...), never a launcher; counter passes are separate
# from each other and carry no trace other than the kernel trace.
set -o pipefail
OUT=${1:-gpurun_out/r04_profiles}
mkdir -p "$OUT"
trim() {
  # keep the merge small: the raw traces are not needed
  find "$OUT" -name "*kernel_trace.csv" -size +2M -delete
  # the counter files of the runs with extras list every dispatch: keep Decrypt's lift and the kernels of an EAdd call
  for f in "$OUT"/pmc_fetch_extra/fetch_counter_collection.csv "$OUT"/pmc_write_extra/write_counter_collection.csv; do
    [ -f "$f" ] && { head -1 "$f" > "$f.tmp"; grep -E "k_pairing<[0-9]+, 1>|k_g1_add<|k_decode<[0-9]+, true>|k_encode<" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; }
  done
  return 0
}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# STAGE=bench: the bench line, its kernel statistics and counter passes; STAGE=sweeps: the sweeps; default: both
# (a gpurun call is limited to 20 minutes: the two stages are run as two calls)
if [ "${STAGE:-all}" = "pmc_extra" ]; then
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_extra" -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch_extra.json" 2> "$OUT/pmc_fetch_extra.err" || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_extra" -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write_extra.json" 2> "$OUT/pmc_write_extra.err" || exit 1
trim; du -sh "$OUT"; exit 0
fi
if [ "${STAGE:-all}" != "sweeps" ]; then
echo "== bench (default line)"; python3 bench.py --steps 3 --warmup 1 > "$OUT/bench_line.json" 2> "$OUT/bench.err" || exit 1
echo "== kernel stats, headline only"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_bench" -o bench -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/bench_line_profiled.json" 2> "$OUT/prof_bench.err" || exit 1
echo "== kernel stats, extras"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_extra" -o extra -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/extra_line_profiled.json" 2> "$OUT/prof_extra.err" || exit 1
echo "== PMC FETCH_SIZE"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/pmc_fetch.json" 2> "$OUT/pmc_fetch.err" || exit 1
echo "== PMC WRITE_SIZE"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > "$OUT/pmc_write.json" 2> "$OUT/pmc_write.err" || exit 1
echo "== PMC FETCH_SIZE / WRITE_SIZE with the extras (Decrypt's lift kernel)"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_extra" -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_fetch_extra.json" 2> "$OUT/pmc_fetch_extra.err" || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_extra" -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline > "$OUT/pmc_write_extra.json" 2> "$OUT/pmc_write_extra.err" || exit 1
trim
[ "${STAGE:-all}" = "bench" ] && { du -sh "$OUT"; exit 0; }
fi
echo "== small-batch sweep"; python3 tools/small_batch_sweep.py > "$OUT/small_batch.csv" 2> "$OUT/sweep.err" || exit 1
echo "== mid-size batches: the three pairing kernels"; python3 tools/quad_sweep.py k512 k1024 > "$OUT/mid_batch.csv" 2> "$OUT/mid.err" || exit 1
echo "== the lane-group kernel saturated, beside the lane kernel"
QUAD_SWEEP_COUNTS=1048576 QUAD_SWEEP_KERNELS=quad,lane python3 tools/quad_sweep.py k1024 > "$OUT/quad_saturated.csv" 2>> "$OUT/mid.err" || exit 1
echo "== single-call latencies"; python3 tools/single_op_latency.py > "$OUT/single_op_latency.csv" 2> "$OUT/single.err" || exit 1
echo "== concurrent single-element callers (the reference's call shape), combiner on / off"
CC_SECONDS=3 python3 tools/concurrent_callers.py k1024 > "$OUT/concurrent_callers.csv" 2> "$OUT/concurrent.err" || exit 1
echo "== MultConst by batch size: lane groups against one element per lane"
python3 tools/multconst_mid_batch.py k1024 k512 > "$OUT/multconst_mid_batch.csv" 2> "$OUT/multconst.err" || exit 1
echo "== EAdd by batch size"; python3 tools/eadd_sweep.py > "$OUT/eadd_sweep.csv" 2> "$OUT/eadd.err" || exit 1
echo "== bgn_ctx_calibrate beside the committed crossovers"; python3 tools/calibrate_report.py > "$OUT/calibrate.csv" 2> "$OUT/calibrate.err" || exit 1
echo "== what the default table sizes buy"
python3 tools/decrypt_vs_table.py > "$OUT/decrypt_vs_table.csv" 2> "$OUT/dvt.err" || exit 1
python3 tools/decrypt_vs_table.py encrypt > "$OUT/encrypt_vs_window.csv" 2>> "$OUT/dvt.err" || exit 1
find "$OUT" -name "*.csv" | head -60
trim
du -sh "$OUT"
