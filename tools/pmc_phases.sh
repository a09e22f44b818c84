#!/bin/bash
# development aid: dynamic instruction counts of k_match_branch with phases switched off (TM_DBG bits)
for d in "$@"; do
  TM_DBG=$d python tools/pmc_profile.py --kernel k_match_branch --groups 0 --out gpurun_out/pmc_d$d > gpurun_out/pmc_d$d.json 2> gpurun_out/pmc_d$d.err
  python - "$d" <<'PY'
import json, sys
d = sys.argv[1]
try:
    j = json.load(open("gpurun_out/pmc_d%s.json" % d))
    k = list(j.values())[0]
    w = k["SQ_WAVES"]
    print("DBG=%s" % d, {c.replace("SQ_INSTS_", ""): round(v / w, 1) for c, v in k.items()})
except Exception as e:
    print("DBG=%s failed: %s" % (d, e)); print(open("gpurun_out/pmc_d%s.err" % d).read()[-800:])
PY
done
