#!/bin/bash
# One GPU call of a development round: the -m gpu parity suite, then the end-to-end bench with the kernel variants behind the
# debug bits timed in the same process.  Everything lands in gpurun_out/$1.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/${1:-round}
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --timeout=400 -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 2 --also-flags ${2:-128,256,384} > $OUT/bench_e2e_1g.json 2> $OUT/bench_e2e_1g.err
echo "bench exit $?" >> $OUT/bench_e2e_1g.err
tail -15 $OUT/pytest_gpu.log
cat $OUT/bench_e2e_1g.json
grep -E "variant|exit|normalize|INVALID|Error|error" $OUT/bench_e2e_1g.err | tail -20
