#!/bin/bash
# tools/variant_ab.sh — A/B of build-time kernel variants (experiments that are off in the product build until they have been timed).
#   bash tools/variant_ab.sh build          HERE, before the GPU call: variants/<name>/libtokenmonster_hip.so for every variant
#                                           (in-tree and git-ignored, so it travels to the GPU box like the product library)
#   bash tools/variant_ab.sh run [MiB=256]  ON THE GPU BOX: K1 time and ids md5 of the product library and of every variant (torch-free: seconds)
#   bash tools/variant_ab.sh check <name>   ON THE GPU BOX: the parity tests with ONE variant library in place, and its instruction counters
# A variant is adopted (its macro flipped in the source) only if its ids are bit-exact and its K1 time is lower on the device.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
declare -A VARIANTS=( [devel]="-DTM_DEVEL" [nohist]="-DTM_SCORE_NOHIST" )
case "${1:-}" in
build)
  for name in "${!VARIANTS[@]}"; do
    tmp=$(mktemp -d)
    cp -r tokenmonster_amd include "$tmp/"; mkdir -p "$tmp/tools"; cp -r tools/devel "$tmp/tools/"
    rm -rf "$tmp/tokenmonster_amd/csrc/build" "$tmp"/tokenmonster_amd/*.so
    ( cd "$tmp" && TM_EXTRA_FLAGS="${VARIANTS[$name]}" python tokenmonster_amd/build.py > "$tmp/build.log" 2>&1 ) || { tail -20 "$tmp/build.log"; exit 1; }
    mkdir -p "variants/$name"
    cp "$tmp/tokenmonster_amd/libtokenmonster_hip.so" "variants/$name/"
    echo "${VARIANTS[$name]}" > "variants/$name/FLAGS"
    rm -rf "$tmp"
    echo "built variants/$name (${VARIANTS[$name]})"
  done ;;
run)
  # seconds, not minutes: K1 of the product library and of every variant library through tools/k1_time.py (ctypes, no torch); the ids of a
  # variant must have the md5 of the product library's
  MB=${2:-256}
  OUT=$ROOT/gpurun_out/variant_ab; mkdir -p "$OUT"
  libs=()
  for name in "${!VARIANTS[@]}"; do [ -f "variants/$name/libtokenmonster_hip.so" ] && libs+=(--lib "variants/$name/libtokenmonster_hip.so"); done
  python tools/k1_time.py --mbytes "$MB" "${libs[@]}" --lib tokenmonster_amd/libtokenmonster_hip.so 2>&1 | tee "$OUT/k1_time_${MB}m.txt" ;;
ablate)
  # where K1's time goes: the TM_DEVEL build with phases switched off through TM_DBG (RESULTS ARE WRONG BY DESIGN, only the time counts):
  # 0 everything, 1 no walks at all, 4 no hash probes, 8 no forward-delete probes, 16 no exit maps, 24 neither
  MB=${2:-256}
  OUT=$ROOT/gpurun_out/variant_ab; mkdir -p "$OUT"
  [ -f variants/devel/libtokenmonster_hip.so ] || { echo "devel: not built (run: bash tools/variant_ab.sh build)"; exit 1; }
  for dbg in 0 1 4 8 16 24; do
    echo -n "TM_DBG=$dbg  "; TM_DBG=$dbg python tools/k1_time.py --one "$ROOT/variants/devel/libtokenmonster_hip.so" --mbytes "$MB" --reps 4 2>&1 | tail -1 | cut -c40-
  done | tee "$OUT/k1_ablation_${MB}m.txt" ;;
check)
  # the parity tests of the tokenizer with ONE variant library in place of the product's (a copy of the tree), and its K1 counters
  name=${2:?variant name}; MB=${3:-256}
  OUT=$ROOT/gpurun_out/variant_ab; mkdir -p "$OUT"
  [ -f "$ROOT/variants/$name/libtokenmonster_hip.so" ] || { echo "$name: not built (run: bash tools/variant_ab.sh build)"; exit 1; }
  work=/tmp/ab_$name; rm -rf "$work"; mkdir -p "$work"
  tar -C "$ROOT" --exclude=.git --exclude=gpurun_out --exclude=variants -cf - . | tar -C "$work" -xf -
  cp "$ROOT/variants/$name/libtokenmonster_hip.so" "$work/tokenmonster_amd/libtokenmonster_hip.so"
  ( cd "$work" && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -p no:cacheprovider -k "not full_size" > "$OUT/pytest_$name.log" 2>&1; echo "$name: pytest exit $? $(tail -1 "$OUT/pytest_$name.log")" )
  ( cd /tmp && export TMPDIR=/tmp && timeout 600 python "$ROOT/tools/pmc_profile.py" --fast --lib "$ROOT/variants/$name/libtokenmonster_hip.so" --mbytes "$MB" --groups 0 --kernel k_match_branch --out "$OUT/pmc_$name" > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" )
  python - "$OUT/pmc_$name.json" "$name" <<'PY'
import json, sys
try:
    k = list(json.load(open(sys.argv[1])).values())[0]
    w = k["SQ_WAVES"]
    print("%s: per wavefront %s" % (sys.argv[2], {c.replace("SQ_INSTS_", ""): round(v / w, 1) for c, v in k.items() if c.startswith("SQ_INSTS")}))
except Exception as ex:
    print("%s: no counters (%s)" % (sys.argv[2], ex))
PY
  ;;
*) echo "usage: $0 build | run [MiB] | check <variant> [MiB] | ablate [MiB]"; exit 2 ;;
esac
