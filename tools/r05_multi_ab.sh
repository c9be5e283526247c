#!/bin/bash
# Round 5: MultPoly (configs[4], bench.py --workload multpoly) per library and per value of option poly_multi, and the
# Decrypt legs of the default line per library.   tools/r05_multi_ab.sh OUTDIR lib1.so [lib2.so ...]
set -o pipefail
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  name=$(basename "$lib" .so)
  for m in 0 1; do
    BGN_AMD_LIB="$lib" BGN_POLY_MULTI=$m timeout -k 10 300 python3 bench.py --workload multpoly --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/${name}_multi$m.json" 2> "$OUT/${name}_multi$m.err" || { tail -3 "$OUT/${name}_multi$m.err"; exit 1; }
    python3 -c "
import json
l=json.loads([x for x in open('$OUT/${name}_multi$m.json') if x.startswith('{')][-1]); print('$name poly_multi=$m %.4g %s %.1f ms' % (l['value'], l['unit'], l['ms_per_step']))"
  done
  BGN_AMD_LIB="$lib" timeout -k 10 400 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > "$OUT/${name}_line.json" 2> "$OUT/${name}.err" || exit 1
  python3 -c "
import json
l=json.loads([x for x in open('$OUT/${name}_line.json') if x.startswith('{')][-1]); ex=l['extra']
print('$name', 'EMult %.4g' % l['value'], {k: float('%.4g' % ex[k]['value']) for k in ('decrypt','decrypt_2^20','decrypt_l2','multpoly','encrypt')})"
done
