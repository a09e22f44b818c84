#!/bin/bash
# Regenerates the measurements quoted in DESIGN.md into gpurun_out/$1 (copy what is to be kept into profiles/).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/${1:-final}
mkdir -p $OUT
python bench.py --steps 10 --warmup 3 > $OUT/bench_e2e_1g.json 2> $OUT/bench_e2e_1g.err
python bench.py --steps 10 --warmup 3 --hot-path-only > $OUT/bench_hot_1g.json 2> $OUT/bench_hot_1g.err
python bench.py --workload score --steps 10 --warmup 3 > $OUT/bench_score_1g.json 2> $OUT/bench_score_1g.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats_e2e -o e2e --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --verify 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats_e2e.err)
python tools/pmc_profile.py --mbytes 256 --groups 0,1,3 --out $OUT/pmc > $OUT/pmc_256m.json 2> $OUT/pmc_256m.err
python tools/pmc_profile.py --mbytes 1024 --groups 4,5 --kernel k_match_branch --out $OUT/pmc_traffic > $OUT/traffic_k1_1g.json 2> $OUT/traffic.err
for f in bench_e2e_1g bench_hot_1g bench_score_1g; do cut -c1-260 $OUT/$f.json; echo; done
cat $OUT/traffic_k1_1g.json
