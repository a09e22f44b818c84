#!/bin/bash
# tools/variant_ab.sh — A/B of build-time kernel variants (experiments that are off in the product build until they have been timed).
#   bash tools/variant_ab.sh build          HERE, before the GPU call: variants/<name>/libtokenmonster_hip.so for every variant
#                                           (in-tree and git-ignored, so it travels to the GPU box like the product library)
#   bash tools/variant_ab.sh run [MiB=256]  ON THE GPU BOX: for the default build and every variant, a copy of the tree with that
#                                           library in place runs the parity tests of the tokenizer and times K1 on the bench corpus
# A variant is adopted (its macro flipped in the source) only if its ids are bit-exact and its K1 time is lower on the device.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
declare -A VARIANTS=( [halo]="-DTM_K1_HALO_SHARE=1" [skip]="-DTM_SKIP_EDGES=1" [halo_skip]="-DTM_K1_HALO_SHARE=1 -DTM_SKIP_EDGES=1" [dense]="-DTM_HASH_QUARTERS=5" [sparse]="-DTM_HASH_QUARTERS=20" [halo_dense]="-DTM_K1_HALO_SHARE=1 -DTM_HASH_QUARTERS=5" [hot]="-DTM_HASH_HOT_FIRST=1" [dense_hot]="-DTM_HASH_QUARTERS=5 -DTM_HASH_HOT_FIRST=1" )
case "${1:-}" in
build)
  for name in "${!VARIANTS[@]}"; do
    tmp=$(mktemp -d)
    cp -r tokenmonster_amd include "$tmp/"
    rm -rf "$tmp/tokenmonster_amd/csrc/build" "$tmp"/tokenmonster_amd/*.so
    ( cd "$tmp" && TM_EXTRA_FLAGS="${VARIANTS[$name]}" python tokenmonster_amd/build.py > "$tmp/build.log" 2>&1 ) || { tail -20 "$tmp/build.log"; exit 1; }
    mkdir -p "variants/$name"
    cp "$tmp/tokenmonster_amd/libtokenmonster_hip.so" "variants/$name/"
    echo "${VARIANTS[$name]}" > "variants/$name/FLAGS"
    rm -rf "$tmp"
    echo "built variants/$name (${VARIANTS[$name]})"
  done ;;
run)
  MB=${2:-256}
  OUT=$ROOT/gpurun_out/variant_ab; mkdir -p "$OUT"
  for name in default "${!VARIANTS[@]}"; do
    work=/tmp/ab_$name; rm -rf "$work"; mkdir -p "$work"
    tar -C "$ROOT" --exclude=.git --exclude=gpurun_out --exclude=variants -cf - . | tar -C "$work" -xf -
    if [ "$name" != default ]; then
      [ -f "$ROOT/variants/$name/libtokenmonster_hip.so" ] || { echo "$name: not built (run: bash tools/variant_ab.sh build)"; continue; }
      cp "$ROOT/variants/$name/libtokenmonster_hip.so" "$work/tokenmonster_amd/libtokenmonster_hip.so"
    fi
    ( cd "$work" && timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -q -m gpu -x -k "not full_size" > "$OUT/pytest_$name.log" 2>&1; echo "$name: pytest exit $? $(tail -1 "$OUT/pytest_$name.log")"
      timeout 600 python bench.py --mbytes "$MB" --steps 8 --warmup 2 --hot-path-only --no-cpu-baseline --no-host-to-host > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err"
      # dynamic instruction counts of K1 per wavefront (what the host model predicts): one rocprofv3 --pmc pass
      ( cd /tmp && export TMPDIR=/tmp && timeout 600 python "$work/tools/pmc_profile.py" --fast --mbytes "$MB" --groups 0 --kernel k_match_branch --out "$OUT/pmc_$name" > "$OUT/pmc_$name.json" 2> "$OUT/pmc_$name.err" )
      python - "$OUT/bench_$name.json" "$name" "$OUT/pmc_$name.json" <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1]))
    print("%s: %.3f ms/step, kernels %s, verified %s" % (sys.argv[2], j["ms_per_step"], j["roofline"]["kernel_ms"], j["config"]["verified_docs_vs_oracle"]))
except Exception as ex:
    print("%s: no bench line (%s)" % (sys.argv[2], ex))
try:
    k = list(json.load(open(sys.argv[3])).values())[0]
    w = k["SQ_WAVES"]
    print("%s: per wavefront %s" % (sys.argv[2], {c.replace("SQ_INSTS_", ""): round(v / w, 1) for c, v in k.items() if c.startswith("SQ_INSTS")}))
except Exception as ex:
    print("%s: no counters (%s)" % (sys.argv[2], ex))
PY
    )
  done ;;
*) echo "usage: $0 build | run [MiB]"; exit 2 ;;
esac
