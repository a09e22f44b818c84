#!/bin/bash
# round 6, GPU call 12: the whole -m gpu suite on this tree; the decode workload (bench line + kernel stats)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe12; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 900 python bench.py --workload decode --steps 10 --warmup 3 > $OUT/bench_decode.json 2> $OUT/bench_decode.err; tail -3 $OUT/bench_decode.err | cut -c1-300; cut -c1-1500 $OUT/bench_decode.json
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/$OUT/stats_decode -o s --output-format csv -- python $R/bench.py --workload decode --steps 5 --warmup 2 --no-cpu-baseline --verify 0 > $R/$OUT/bench_decode_under_rocprof.json 2> $R/$OUT/stats_decode.err)
f=$(find $OUT/stats_decode -name "*kernel_stats.csv" | head -1)
find $OUT/stats_decode -name "*kernel_trace.csv" -delete
[ -n "$f" ] && cp "$f" $OUT/kernel_stats_decode.csv && head -12 "$f" | cut -c1-60,100-260
