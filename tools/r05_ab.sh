#!/bin/bash
# Round 5: same-box A/B of library builds (BGN_AMD_LIB) — the GPU suite on the new build, then the bench line
# (headline + secondaries, no CPU leg) of every library named.   tools/r05_ab.sh OUTDIR lib1.so [lib2.so ...]
set -o pipefail
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# (the micro-benchmark is built from its source here: no binary of it is tracked)
if [ -f tools/ubench/lane_sop.hip ]; then
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Ibgn_amd/csrc tools/ubench/lane_sop.hip -o tools/ubench/lane_sop 2> "$OUT/lane_sop_build.err" \
    && { tools/ubench/lane_sop > "$OUT/lane_sop.txt" 2>&1 || exit 1; cat "$OUT/lane_sop.txt"; }
fi
if [ -z "$AB_SKIP_TESTS" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -q > "$OUT/gpu_suite.log" 2>&1; rc=$?; tail -3 "$OUT/gpu_suite.log"
  [ $rc -eq 0 ] || [ $rc -eq 1 ] || exit $rc          # (failed tests are reported; a killed run is not followed by another GPU step)
fi
for lib in "$@"; do
  name=$(basename "$lib" .so)
  echo "== $name"
  BGN_AMD_LIB="$lib" timeout -k 10 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline ${AB_BENCH_ARGS} > "$OUT/${name}_line.json" 2> "$OUT/${name}.err" || { tail -5 "$OUT/${name}.err"; exit 1; }
  python3 - "$OUT/${name}_line.json" <<'PY'
import json, sys
l = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][-1])
print("EMult %.4g /s  %.1f ms  kernel %s %.1f ms" % (l["value"], l["ms_per_step"], l["roofline"]["kernel"], l["roofline"]["kernel_ms"]))
ex = l.get("extra", {})
for k in ("encrypt", "eadd_l1", "multpoly", "decrypt", "decrypt_2^20", "decrypt_l2"):
    if k in ex:
        print("  %-12s %.4g %s" % (k, ex[k]["value"], ex[k]["unit"]))
if "mult_mid_batch" in ex:
    print("  mid-batch", {k: round(v["ms"], 1) for k, v in ex["mult_mid_batch"]["sizes"].items()})
PY
done
