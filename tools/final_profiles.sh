#!/bin/bash
# Regenerates the measurements quoted in DESIGN.md into gpurun_out/$1 (copy what is to be kept into profiles/).
set -u
OUT=/root/repo/gpurun_out/${1:-final}
mkdir -p $OUT
cd /root/repo
python bench.py --steps 10 --warmup 3 > $OUT/bench_e2e_1g.json 2> $OUT/bench_e2e_1g.err
python bench.py --steps 10 --warmup 3 --hot-path-only > $OUT/bench_hot_1g.json 2> $OUT/bench_hot_1g.err
python bench.py --workload score --steps 10 --warmup 3 > $OUT/bench_score_1g.json 2> $OUT/bench_score_1g.err
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $OUT/stats_e2e -o e2e --output-format csv -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats_e2e.err)
python tools/pmc_profile.py --mbytes 1024 --groups 0,1,2,3 --out $OUT/pmc > $OUT/pmc_1g.json 2> $OUT/pmc_1g.err
python tools/pmc_profile.py --mbytes 1024 --groups 4,5 --kernel k_match_branch --out $OUT/pmc_traffic > $OUT/traffic_k1_1g.json 2> $OUT/traffic.err
tail -c 600 $OUT/bench_e2e_1g.json; echo; cat $OUT/traffic_k1_1g.json
