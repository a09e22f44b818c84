#!/bin/bash
# round 6, GPU call 8: the ring with its thin kernels fused (21 -> 10 launches per chunk)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe8; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_host_api.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_host_api.log 2>&1; tail -3 $OUT/pytest_host_api.log
S="python tools/h2h_sweep.py"
( $S 4:32 4:48 4:24 4:16
  TM_RING_STREAMS=3 $S 4:32 4:48
  TM_RING_FIRST_KIB=1024 $S 4:32 4:48 ) 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
TM_TRACE=1 python tools/h2h_lane_trace.py 8 2> $OUT/ring_trace.txt | tail -2
