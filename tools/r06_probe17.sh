#!/bin/bash
# round 6, GPU call 17: tables laid out by use (tm_vocab_tune) on every shape, beside the untuned lines
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe17; mkdir -p $OUT
for cfg in englishcode-32000-consistent englishcode-100256-clean; do
  for tune in 0 64; do
    timeout 600 python bench.py --steps 8 --warmup 3 --config $cfg --tune-mib $tune --no-cpu-baseline --no-host-to-host --verify 0 > $OUT/bench_${cfg}_tune$tune.json 2> $OUT/bench_${cfg}_tune$tune.err
    python - $OUT/bench_${cfg}_tune$tune.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step", d["ms_per_step"], "K1 frac", r["frac"], "traffic", r.get("traffic"), "l2", d.get("roofline_l2",{}).get("requests_per_segment"))
PY
  done
done
for tune in 0 64; do
  timeout 600 python bench.py --workload score --steps 8 --warmup 3 --tune-mib $tune --no-cpu-baseline --verify 0 > $OUT/bench_score_tune$tune.json 2> $OUT/bench_score_tune$tune.err
  python - $OUT/bench_score_tune$tune.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step", d["ms_per_step"], "K1 frac", r["frac"], "traffic", r.get("traffic"))
PY
done
