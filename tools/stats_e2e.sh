#!/bin/bash
# rocprofv3 kernel stats of the default end-to-end bench into gpurun_out/$1 (copy what is kept to profiles/)
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-stats}; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_e2e -o s --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host --verify 0 > $OUT/bench_under_rocprof.json 2>/dev/null)
f=$(find $OUT/stats_e2e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_e2e.csv && python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 12: print("   %-44s calls %4s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
cut -c1-200 $OUT/bench_under_rocprof.json
