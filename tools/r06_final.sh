#!/bin/bash
# round 6, lines of record from one tree: the driver's three steps (pytest -m gpu, smoke, default bench), kernel stats of the same step, the other
# shapes with their kernel stats, the decode and scoring lines, the rank protocol of the 8-GPU scoring target, device fuzz.  Output: gpurun_out/$1
cd "$(dirname "$0")/.."
ROOT=$PWD; NAME=${1:-r06_final}; OUT=gpurun_out/$NAME; mkdir -p $OUT
git rev-parse HEAD > $OUT/HEAD 2>/dev/null
timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1; grep -E "passed|failed" $OUT/pytest_gpu.log | tail -2; grep -q failed $OUT/pytest_gpu.log && grep -v "^  File" $OUT/pytest_gpu.log | tail -60 | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $OUT/smoke.txt
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-260 $OUT/bench_default.json; grep -E "INVALID|rror" $OUT/bench_default.err | head -5
bash tools/bench_all_shapes.sh $NAME/shapes > $OUT/shapes.log 2>&1; grep -E "^==|calls" $OUT/shapes.log | cut -c1-200
timeout 600 python bench.py --workload decode --steps 10 --warmup 3 > $OUT/bench_decode.json 2> $OUT/bench_decode.err; cut -c1-260 $OUT/bench_decode.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/stats_decode -o s --output-format csv -- python $ROOT/bench.py --workload decode --steps 5 --warmup 2 --no-cpu-baseline --verify 0 > $ROOT/$OUT/bench_decode_under_rocprof.json 2> $ROOT/$OUT/stats_decode.err)
f=$(find $OUT/stats_decode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/kernel_stats_decode.csv
timeout 600 python tools/score_rank_protocol.py > $OUT/score_rank_protocol.json 2> $OUT/score_rank_protocol.err; grep -E "projected_scaling_8|t_rank_ms_single|median" $OUT/score_rank_protocol.json | head -5
timeout 120 python tools/gpu_fuzz.py 40 860601 2>&1 | tail -1 | tee $OUT/gpu_fuzz.txt
find $OUT -name "*kernel_trace.csv" -delete
