#!/bin/bash
# round 6, GPU call 19: what bounds k_dec_capcode - the kernel without its stores, without its scans, without both (variants/dec_*: wrong output, timing only)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe19; mkdir -p $OUT
cp tokenmonster_amd/libtokenmonster_hip.so /tmp/product.so
for v in product dec_nostore dec_noscan dec_both; do
  [ $v = product ] && cp /tmp/product.so tokenmonster_amd/libtokenmonster_hip.so || cp variants/$v/libtokenmonster_hip.so tokenmonster_amd/libtokenmonster_hip.so
  timeout 300 python bench.py --workload decode --steps 5 --warmup 2 --no-cpu-baseline --verify 0 > $OUT/bench_$v.json 2> $OUT/bench_$v.err
  python - $OUT/bench_$v.json $v <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); print(sys.argv[2], "ms/step", d["ms_per_step"], json.dumps(d["config"].get("stage_ms") or d.get("stage_ms") or {k:v for k,v in d.items() if "stage" in k}))
PY
done 2>&1 | tee $OUT/ablation.txt
cp /tmp/product.so tokenmonster_amd/libtokenmonster_hip.so
