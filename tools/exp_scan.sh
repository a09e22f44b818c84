#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E "host_to_host|^\{" | cut -c1-560
for cfg in "4 32" "6 16" "8 16"; do set -- $cfg; python tools/h2h_trace.py --lanes $1 --chunk-mib $2 --passes 5 2>&1 | grep -E "^pass [34]|rror"; done
