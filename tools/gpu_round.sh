#!/bin/bash
# One GPU call of a development round: the -m gpu parity suite, then the end-to-end bench, the scoring bench and the per-kernel times.  Everything lands in gpurun_out/$1.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-round}
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --timeout=400 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench_e2e_1g.json 2> $OUT/bench_e2e_1g.err
echo "bench exit $?" >> $OUT/bench_e2e_1g.err
tail -15 $OUT/pytest_gpu.log
cat $OUT/bench_e2e_1g.json
grep -E "variant|exit|normalize|INVALID|Error|error" $OUT/bench_e2e_1g.err | tail -20
timeout 600 python bench.py --workload score --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_score_1g.json 2> $OUT/bench_score_1g.err
cut -c1-400 $OUT/bench_score_1g.json; echo
# per-kernel times of the same step (rocprofv3 kernel trace)
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$OUT/stats_e2e -o e2e --output-format csv -- python $OLDPWD/bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 0 > $OLDPWD/$OUT/bench_under_rocprof.json 2> $OLDPWD/$OUT/stats_e2e.err)
f=$(find $OUT/stats_e2e -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cut -d, -f1-4 "$f" | cut -c1-60,200- | head -14
