#!/bin/bash
# development aid: time of k_match_branch with phases switched off / occupancy reduced (TM_DBG bits; results are wrong, only the
# time is of interest).  256 MiB, hot path only.
cd "$(dirname "$0")/.."
for d in 0 512 4 8 16 1; do
  TM_DBG=$d timeout 200 python bench.py --mbytes 256 --steps 5 --warmup 2 --hot-path-only --no-cpu-baseline --verify 0 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read()); print('TM_DBG=$d', j['ms_per_step'], j['roofline']['kernel_ms'])"
done
