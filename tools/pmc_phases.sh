#!/bin/bash
# development aid: dynamic instruction counts and time of k_match_branch with phases switched off (TM_DBG bits; needs a -DTM_DEVEL build:
#   TM_EXTRA_FLAGS=-DTM_DEVEL python tokenmonster_amd/build.py --force   — results are wrong by design, rebuild without it afterwards)
cd "$(dirname "$0")/.."
for d in "$@"; do
  TM_DBG=$d python tools/pmc_profile.py --kernel k_match_branch --groups 0,3 --out gpurun_out/pmc_d$d > gpurun_out/pmc_d$d.json 2> gpurun_out/pmc_d$d.err
  t=$(TM_DBG=$d python bench.py --mbytes 256 --steps 5 --warmup 2 --hot-path-only --no-cpu-baseline --verify 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['roofline']['kernel_ms']['match_branch'])")
  python - "$d" "$t" <<'PY'
import json, sys
d, t = sys.argv[1], sys.argv[2]
try:
    j = json.load(open("gpurun_out/pmc_d%s.json" % d))
    k = list(j.values())[0]
    w = k["SQ_WAVES"]
    print("DBG=%s K1 %s ms/256MiB" % (d, t), {c.replace("SQ_INSTS_", ""): round(v / w, 1) for c, v in k.items() if c.startswith("SQ_INSTS")}, {c: round(v / w, 1) for c, v in k.items() if c.startswith("TCP")})
except Exception as e:
    print("DBG=%s failed: %s" % (d, e)); print(open("gpurun_out/pmc_d%s.err" % d).read()[-800:])
PY
done
