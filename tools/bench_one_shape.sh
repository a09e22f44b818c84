#!/bin/bash
# tools/bench_one_shape.sh <name> <bench args...>: one shape of tools/bench_all_shapes.sh (bench line + rocprofv3 kernel stats) into gpurun_out/one_shape
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/one_shape; mkdir -p $OUT
name=$1; shift
timeout 900 python bench.py --steps 10 --warmup 3 "$@" > $OUT/bench_$name.json 2> $OUT/bench_$name.err
echo "== $name: exit $?"; cut -c1-420 $OUT/bench_$name.json; echo; grep -E "cpu_baseline|INVALID|rror" $OUT/bench_$name.err | cut -c1-400
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_$name -o s --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --verify 0 "$@" > $OUT/bench_${name}_under_rocprof.json 2> $OUT/stats_$name.err)
f=$(find $OUT/stats_$name -name "*kernel_stats.csv" | head -1)
find $OUT/stats_$name -name "*kernel_trace.csv" -delete
[ -n "$f" ] && cp "$f" $OUT/kernel_stats_$name.csv && head -8 "$f" | cut -c1-50,120-330
