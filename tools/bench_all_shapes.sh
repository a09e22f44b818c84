#!/bin/bash
# Every BASELINE.json shape on one GPU: bench line + rocprofv3 kernel stats each, into gpurun_out/$1 (copy what is kept to profiles/).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-shapes}
mkdir -p $OUT
run() {   # name, bench args...
  local name=$1; shift
  timeout 900 python bench.py --steps 10 --warmup 3 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  echo "== $name: exit $?"; cut -c1-420 $OUT/bench_$name.json; echo; grep -E "host_to_host|cpu_baseline|INVALID|rror" $OUT/bench_$name.err | cut -c1-600
  (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_$name -o s --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --verify 0 "$@" > $OUT/bench_${name}_under_rocprof.json 2> $OUT/stats_$name.err)
  f=$(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1)
  find $OUT/stats_$name -name "*kernel_trace.csv" -delete
  [ -n "$f" ] && cp "$f" $OUT/kernel_stats_$name.csv && python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 7: print("   %-44s calls %4s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
}
run e2e_englishcode32000
run e2e_englishcode100256 --config englishcode-100256-clean --no-host-to-host
run e2e_code4096_nocapcode --config code-4096-balanced-nocapcode --no-host-to-host
run score_candidates65536 --workload score
