#!/bin/bash
# end-of-round check on the GPU box: the driver's three steps (pytest -m gpu, smoke, bench) + kernel stats of the default bench
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log; grep -q failed $OUT/pytest_gpu.log && grep -v "^  File" $OUT/pytest_gpu.log | tail -60 | cut -c1-400
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json; grep -E "host_to_host" $OUT/bench_default.err | cut -c1-500
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/stats_e2e -o s --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --verify 0 > $ROOT/$OUT/bench_under_rocprof.json 2>/dev/null)
f=$(find $ROOT/$OUT/stats_e2e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_e2e.csv && head -9 $f | cut -c1-40,150-330
find $OUT -name "*kernel_trace.csv" -size +4M -delete
