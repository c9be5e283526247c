#!/bin/bash
# Same-box A/B of builds of the library (BGN_AMD_LIB): headline step time and the HBM-side traffic of the
# dominant kernel (separate rocprofv3 --pmc passes, no tracing), for DESIGN.md's traffic table.
#   tools/ab_traffic.sh OUTDIR lib1.so [lib2.so ...]
set -o pipefail
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lib in "$@"; do
  name=$(basename "$lib" .so)
  echo "== $name"
  BGN_AMD_LIB="$lib" python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra > "$OUT/${name}_line.json" 2> "$OUT/${name}.err" || exit 1
  BGN_AMD_LIB="$lib" rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${name}_fetch" -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/${name}_fetch.err" || exit 1
  BGN_AMD_LIB="$lib" rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${name}_write" -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra > /dev/null 2> "$OUT/${name}_write.err" || exit 1
done
python3 - "$OUT" "$@" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
print("build,ms_per_step,pairings_per_s,fetch_KB,write_KB,bytes_per_pairing,traffic_over_algorithmic,scratch_B_per_lane")
for lib in sys.argv[2:]:
    name = os.path.basename(lib)[:-3]
    line = json.loads([l for l in open(os.path.join(out, name + "_line.json")) if l.startswith("{")][-1])
    vals = {}
    scratch = None
    for ctr, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
        path = glob.glob(os.path.join(out, "%s_%s" % (name, sub), "**", "*counter_collection.csv"), recursive=True)[0]
        rows = [r for r in csv.DictReader(open(path)) if "k_pairing<" in r["Kernel_Name"] and ", 0>" in r["Kernel_Name"]
                and r["Counter_Name"] == ctr]
        vals[ctr] = sum(float(r["Counter_Value"]) for r in rows) / len(rows)
        scratch = rows[0]["Scratch_Size"]
    n = line["config"]["batch_per_gpu"]
    tot = (vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024
    print("%s,%.1f,%.0f,%.0f,%.0f,%.0f,%.1f,%s" % (name, line["ms_per_step"], line["value"], vals["FETCH_SIZE"], vals["WRITE_SIZE"],
                                                 tot / n, tot / n / line["roofline"]["algorithmic_bytes_per_pairing"], scratch))
PY
