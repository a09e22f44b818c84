#!/bin/bash
# round 6, GPU call 21: tm_vocab_load_sample - parity test, the bench line with its tables-by-use leg on two shapes, traffic of the match kernel under tuned tables
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe21; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_host_api.py -x -q -p no:cacheprovider -k "laid_out" 2>&1 | tail -2
for cfg in englishcode-32000-consistent englishcode-100256-clean; do
  timeout 900 python bench.py --steps 8 --warmup 3 --config $cfg --no-cpu-baseline --no-host-to-host --verify 0 > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  python - $OUT/bench_$cfg.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step", d["ms_per_step"], "K1", r["kernel_ms"].get("match_branch"), "frac", r["frac"], "traffic", r.get("traffic"), "| by use:", json.dumps(d.get("tables_by_use")))
PY
done
for t in 0 16; do
  TM_K1_TUNE_MIB=$t python tools/pmc_profile.py --fast --mbytes 1024 --groups 3,4,5 --kernel k_match_branch --extra="--config englishcode-100256-clean" --out $OUT/pmc_t$t > $OUT/traffic_100256_tune$t.json 2> $OUT/traffic_t$t.err
  echo "tune $t:"; cat $OUT/traffic_100256_tune$t.json
done
