#!/bin/bash
# Keeps the multi-GPU benches ready for the day an N-GPU node is there (SCALE has never run on hardware): runs bench.py for N = 1 and
# N = every visible device, both workloads, BOTH drivers - one torch.distributed rank per GPU (what the round-end SCALE run launches) and
# --in-process (ONE process through the library's own multi-device driver, RCCL inside tm_score_multi: the path a Go host uses) - on a
# small corpus (--mbytes, default 64), then checks: every line names the n_gpus asked for, RCCL saw N ranks (config.rccl_ranks), and at
# N = 1 the two drivers agree within 3 %.  On a one-GPU box `TM_VIRTUAL_DEVICES=4 tools/scale_dryrun.sh` walks the in-process code paths
# with 4 members on device 0 (a code-path check, not a measurement).  Output: gpurun_out/scale_dryrun/*.json + a summary table.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/scale_dryrun; mkdir -p $OUT
MB=${MBYTES:-64}
NDEV=$(python -c "from tokenmonster_amd import _native as N; print(N.lib.tm_device_count())")
NS="1"; [ "$NDEV" -gt 1 ] && NS="1 $NDEV"
[ -n "${TM_VIRTUAL_DEVICES:-}" ] && NSV="1 ${TM_VIRTUAL_DEVICES}" || NSV="$NS"
COMMON="--steps 5 --warmup 2 --mbytes $MB --no-cpu-baseline --no-host-to-host --no-measure-traffic"
for W in tokenize score; do
  for N in $NS; do
    timeout 900 python bench.py --gpus $N --workload $W $COMMON > $OUT/ranks_${W}_$N.json 2> $OUT/ranks_${W}_$N.err || echo "ranks $W N=$N failed (see $OUT/ranks_${W}_$N.err)"
  done
  for N in $NSV; do
    timeout 900 python bench.py --in-process --gpus $N --workload $W $COMMON > $OUT/inproc_${W}_$N.json 2> $OUT/inproc_${W}_$N.err || echo "in-process $W N=$N failed (see $OUT/inproc_${W}_$N.err)"
  done
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = sys.argv[1]
rows, ok = {}, True
for p in sorted(glob.glob(os.path.join(out, "*.json"))):
    name = os.path.basename(p)[:-5]
    try:
        j = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception as ex:
        print("%-24s NO JSON LINE (%s)" % (name, ex)); ok = False; continue
    drv, w, n = name.split("_")
    rows[(drv, w, int(n))] = j
    c = j["config"]
    print("%-24s n_gpus %d  %8.2f GB/s  %8.3f ms/step  rccl_ranks %s  verified %s" % (
        name, j["n_gpus"], j["value"], j["ms_per_step"], c.get("rccl_ranks"), c.get("verified_docs_vs_reference", c.get("verified_bytes_vs_oracle"))))
    if j["n_gpus"] != int(n):
        print("   !! n_gpus differs from the request"); ok = False
    if w == "score" and int(n) > 1 and not c.get("virtual_devices") and c.get("rccl_ranks") != int(n):
        print("   !! RCCL did not see %s ranks" % n); ok = False
for w in ("tokenize", "score"):
    a, b = rows.get(("ranks", w, 1)), rows.get(("inproc", w, 1))
    if a and b:
        d = abs(a["value"] - b["value"]) / a["value"]
        print("N = 1 %-8s ranks %.2f vs in-process %.2f GB/s: %.1f %%%s" % (w, a["value"], b["value"], 100 * d, "" if d <= 0.03 or w == "score" else "   !! more than 3 % apart"))
        # (score: the in-process step ends with the histogram on the HOST, the rank step with it in HBM - a fixed ~0.3 ms at any size)
print("scale dry run:", "OK" if ok else "PROBLEMS")
PY
