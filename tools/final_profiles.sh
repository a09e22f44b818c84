#!/bin/bash
# Regenerates the measurements quoted in DESIGN.md into gpurun_out/$1 (copy what is to be kept into profiles/ under the round's name).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-final}
mkdir -p $OUT
bash tools/bench_all_shapes.sh ${1:-final} > $OUT/shapes.log 2>&1
tail -40 $OUT/shapes.log | cut -c1-300
python bench.py --steps 10 --warmup 3 --hot-path-only > $OUT/bench_hot_1g.json 2> $OUT/bench_hot_1g.err
python tools/pmc_profile.py --mbytes 256 --groups 0,1,2,3 --out $OUT/pmc > $OUT/pmc_256m.json 2> $OUT/pmc_256m.err
python tools/pmc_profile.py --mbytes 1024 --groups 4,5 --kernel k_match_branch --e2e --out $OUT/pmc_traffic > $OUT/traffic_k1_1g.json 2> $OUT/traffic.err
cat $OUT/traffic_k1_1g.json
[ -x tools/candidate_throughput ] && timeout 600 tools/candidate_throughput 1024 48 8 > $OUT/candidate_throughput.txt 2>&1; cat $OUT/candidate_throughput.txt
