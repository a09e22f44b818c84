#!/bin/bash
# round 6, GPU call 11: what the scoring walk costs without its histogram; the bench's host-to-host with eight queues from the start
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe11; mkdir -p $OUT
python tools/k1_time.py --score --config candidates-65536 --mbytes 1024 --lib variants/nohist/libtokenmonster_hip.so 2>&1 | grep -v Warn | tee $OUT/score_nohist.txt
TM_TEST_HOOKS=1 TM_DBG=0 python - <<'PY' 2>&1 | tee -a $OUT/score_nohist.txt
import subprocess, sys, os
# hook 15: the word-staging walk (k_score_tiles) on the same rows, for the time of record beside it
env = dict(os.environ, TM_K1_HOOKS="32768")
PY
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --verify 0 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r06_probe11/bench_default.json"))
print(d["value"], d["ms_per_step"], d["value_host_to_host"], d["host_to_host"]["pinned"]["ms_each"], d["host_to_host"]["pageable"]["value"])
PY
