#!/bin/bash
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        j=json.loads(l); print('e2e ms', j['ms_per_step'], 'k1', j['roofline']['kernel_ms'], 'verified', j['config']['verified_docs_vs_oracle'])
"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-to-host --workload score 2>/dev/null | cut -c1-250
python tools/pmc_profile.py --mbytes 1024 --groups 4,5 --kernel k_match_branch --e2e --out gpurun_out/pmc_traffic2 2>/dev/null
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r0stats -o s --output-format csv -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --verify 0 > /dev/null 2>&1)
head -8 $(find gpurun_out/r0stats -name "*kernel_stats.csv" | head -1) | cut -c1-120
