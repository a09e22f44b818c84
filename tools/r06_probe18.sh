#!/bin/bash
# round 6, GPU call 18: the tiled gather of the decode path - parity tests that decode, then the decode bench with kernel stats
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=gpurun_out/r06_probe18; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -p no:cacheprovider -m gpu -k "decode or Decode or capcode" > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -q failed $OUT/pytest.log && grep -v "^  File" $OUT/pytest.log | tail -40 | cut -c1-300
timeout 600 python bench.py --workload decode --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_decode.json 2> $OUT/bench_decode.err; cut -c1-400 $OUT/bench_decode.json; grep -iE "error|mismatch|INVALID" $OUT/bench_decode.err | head
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/stats_decode -o s --output-format csv -- python $ROOT/bench.py --workload decode --steps 5 --warmup 2 --no-cpu-baseline --verify 0 > $ROOT/$OUT/bench_decode_under_rocprof.json 2> $ROOT/$OUT/stats_decode.err)
f=$(find $OUT/stats_decode -name "*kernel_stats.csv" | head -1); find $OUT/stats_decode -name "*kernel_trace.csv" -delete
[ -n "$f" ] && cp "$f" $OUT/kernel_stats_decode.csv && python3 - "$f" <<'PY'
import csv, sys
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < 10: print("   %-44s calls %4s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][-44:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
