#!/bin/bash
# development aid: K1 time (256 MiB, hot path, ids verified against the oracle) + its dynamic instruction / cache counters per wavefront
cd "$(dirname "$0")/.."
python bench.py --mbytes 256 --steps 8 --warmup 2 --hot-path-only --no-cpu-baseline 2>&1 | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('ms', j['ms_per_step'], j['roofline']['kernel_ms'], 'verified', j['config']['verified_docs_vs_oracle'])
    elif 'INVALID' in l or 'rror' in l: print(l.strip())
"
python tools/pmc_profile.py --kernel k_match_branch --groups 0,3 --out gpurun_out/pmc_quick > gpurun_out/pmc_quick.json 2> gpurun_out/pmc_quick.err
python - <<'PY'
import json
k = list(json.load(open("gpurun_out/pmc_quick.json")).values())[0]
w = k["SQ_WAVES"]
print({c.replace("SQ_INSTS_", ""): round(v / w, 1) for c, v in k.items() if c.startswith("SQ_INSTS")}, {c.replace("_sum", ""): round(v / w, 1) for c, v in k.items() if c.startswith("TCP")})
PY
