#!/bin/bash
# (history: ran at the commit before the split pipeline and the profiling switches were removed from the default build; results in
# profiles/r02a_*)
# round 2, GPU call 1: where K1's time goes (phases off) and the per-kernel / counter breakdown of the split pipeline (debug bit 11)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
OUT=$ROOT/gpurun_out/r02a
mkdir -p $OUT
bash tools/k1_ablation.sh > $OUT/ablation.txt 2>&1
cat $OUT/ablation.txt
for d in 0 2048; do
  (cd /tmp && export TMPDIR=/tmp && TM_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_d$d -o s --output-format csv -- python $ROOT/bench.py --mbytes 256 --steps 5 --warmup 2 --hot-path-only --no-cpu-baseline --verify 0 > $OUT/bench_d$d.json 2> $OUT/stats_d$d.err)
  f=$(find $OUT/stats_d$d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $OUT/kernel_stats_d$d.csv && cut -d, -f1-4 "$f" | cut -c1-50,200- | head -8
done
TM_DBG=2048 python tools/pmc_profile.py --mbytes 256 --groups 0,1,3 --out $OUT/pmc_split > $OUT/pmc_split_256m.json 2> $OUT/pmc_split.err
TM_DBG=2048 python tools/pmc_profile.py --mbytes 256 --groups 4,5 --kernel k_match --out $OUT/pmc_split_traffic > $OUT/traffic_split_256m.json 2> $OUT/traffic_split.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r02a/pmc_split_256m.json"))
for k, c in j.items():
    if "k_match" in k:
        w = c.get("SQ_WAVES", 1)
        print(k[:40], {n: round(v / w, 1) for n, v in c.items() if n.startswith("SQ_INSTS")}, {n: int(v) for n, v in c.items() if n.startswith("TC")}, {n: int(v) for n, v in c.items() if "CYCLES" in n or "WAIT" in n})
print(open("gpurun_out/r02a/traffic_split_256m.json").read())
PY
