#!/bin/bash
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=gpurun_out/r02h; mkdir -p $OUT
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default.json; grep -E "host_to_host" $OUT/bench_default.err | cut -c1-500
python bench.py --steps 10 --warmup 3 --workload score > $OUT/bench_score.json 2> $OUT/bench_score.err; cut -c1-300 $OUT/bench_score.json
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/stats_score -o s --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload score --verify 0 > /dev/null 2>&1)
f=$(find $OUT/stats_score -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_score.csv && head -6 $f | cut -c1-50,180-300
[ -x tools/candidate_throughput ] && timeout 600 tools/candidate_throughput 1024 48 8 > $OUT/candidate_throughput.txt 2>&1; cat $OUT/candidate_throughput.txt
find $OUT -name "*kernel_trace.csv" -size +4M -delete
