#!/bin/bash
# round 6, GPU call 2: does the number of hardware queues the runtime spreads the streams over (default 4) bound the lanes?
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe2; mkdir -p $OUT
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q timeout 300 python tools/h2h_sweep.py 4:32 6:32 8:32 8:16 12:16 3:64 2>&1 | grep -v Warning | tee -a $OUT/hwq_sweep.txt
done
