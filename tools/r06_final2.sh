#!/bin/bash
# round 6, last call: the driver's three steps on the final tree (pytest -m gpu, smoke, default bench) + kernel stats of the default step
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=gpurun_out/r06_final2; mkdir -p $OUT
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -q failed $OUT/pytest_gpu.log && grep -v "^  File" $OUT/pytest_gpu.log | tail -60 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-260 $OUT/bench_default.json; grep -E "INVALID|rror" $OUT/bench_default.err | head -5
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/stats_e2e -o s --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --verify 0 > $ROOT/$OUT/bench_under_rocprof.json 2>/dev/null)
f=$(find $ROOT/$OUT/stats_e2e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_e2e.csv && head -7 $f | cut -c1-40,150-260
find $OUT -name "*kernel_trace.csv" -delete
timeout 120 python tools/gpu_fuzz.py 40 860901 2>&1 | tail -1 | tee $OUT/gpu_fuzz.txt
