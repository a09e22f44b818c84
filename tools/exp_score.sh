#!/bin/bash
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "score or hist" 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --workload score 2>&1 | grep -E "^\{|INVALID|rror" | cut -c1-260
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/scstats3 -o s --output-format csv -- python /root/repo/bench.py --steps 5 --warmup 2 --no-cpu-baseline --workload score --verify 0 > /dev/null 2>&1)
head -4 $(find gpurun_out/scstats3 -name "*kernel_stats.csv" | head -1) | cut -c1-60,200-300
