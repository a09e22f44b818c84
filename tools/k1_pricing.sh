#!/bin/bash
# tools/k1_pricing.sh [MiB=256] — ON THE GPU BOX: what the candidate changes to k_match_branch could be worth, priced on the device BEFORE they are built.
# The -DTM_DEVEL library (variants/devel, built by tools/variant_ab.sh build) with phases switched off or loads added through TM_DBG bits —
# RESULTS ARE WRONG BY DESIGN for every line but the first, only the time of K1 and its counters matter:
#   0        the kernel as it is (the -DTM_DEVEL build: its dbg tests cost ~1 %)
#   0x10000  every row gather of step B goes to the same 1 KiB (64 rows): the time the 256 row requests per segment cost = the most any
#            scheme that removes row gathers can win at unchanged instruction count
#   0x20000  one more 16-byte load per A1 round from the SAME 32 bytes as the entry gathered: what the second half of a 32-byte slot
#            (double-array entry + its node's link entry side by side) costs before it saves anything
#   0x40000  one more 16-byte load per A1 round from ANOTHER line (4 KiB away): the price of a separate gather, for comparison
#   4        no double-array probes (SET gathers only)
#   8        no forward-delete probes (A3 + T(p,1))
# and the L1 / L2 request counters of the first two.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
MB=${1:-256}
OUT=$ROOT/gpurun_out/k1_pricing; mkdir -p "$OUT"
LIB=$ROOT/variants/devel/libtokenmonster_hip.so
[ -f "$LIB" ] || { echo "variants/devel not built (bash tools/variant_ab.sh build)"; exit 1; }
for cfg in englishcode-32000-consistent englishcode-100256-clean; do
  for dbg in 0 65536 131072 262144 4 8; do
    echo -n "TM_DBG=$dbg "
    TM_DBG=$dbg timeout 300 python tools/k1_time.py --one "$LIB" --mbytes "$MB" --config $cfg --reps 4 2>&1 | tail -1 | sed -e 's/ids md5.*//'
  done
done | tee "$OUT/k1_pricing_${MB}m.txt"
for dbg in 0 65536; do
  ( cd /tmp && export TMPDIR=/tmp && TM_DBG=$dbg timeout 300 python "$ROOT/tools/pmc_profile.py" --fast --lib "$LIB" --mbytes "$MB" --groups 0,3 --kernel k_match_branch --out "$OUT/pmc_$dbg" > "$OUT/pmc_$dbg.json" 2> "$OUT/pmc_$dbg.err" )
  python - "$OUT/pmc_$dbg.json" $dbg <<'PY'
import json, sys
try:
    k = list(json.load(open(sys.argv[1])).values())[0]
    w = k["SQ_WAVES"]
    print("TM_DBG=%s per wavefront: %s" % (sys.argv[2], {c: round(v / w, 1) for c, v in k.items() if c != "SQ_WAVES"}))
except Exception as ex:
    print("TM_DBG=%s: no counters (%s)" % (sys.argv[2], ex))
PY
done | tee -a "$OUT/k1_pricing_${MB}m.txt"
