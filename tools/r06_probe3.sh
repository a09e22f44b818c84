#!/bin/bash
# round 6, GPU call 3: the ring on the device - parity first, then its knobs
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_host_api.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_host_api.log 2>&1; tail -5 $OUT/pytest_host_api.log
S="python tools/h2h_sweep.py"
( TM_RING=0 $S 4:32
  $S 4:32 4:16 4:64 2:32
  TM_RING_STREAMS=1 $S 4:32
  TM_RING_STREAMS=3 $S 4:32
  TM_RING_SLOTS=3 $S 4:32
  TM_RING_SLOTS=6 $S 4:32
  TM_RING_SLACK=110 $S 4:32
  TM_RING_SLACK=150 $S 4:32
  GPU_MAX_HW_QUEUES=8 $S 4:32
  GPU_MAX_HW_QUEUES=8 TM_RING_STREAMS=3 TM_RING_SLOTS=6 $S 4:32
  GPU_MAX_HW_QUEUES=16 TM_RING=0 $S 4:32 ) 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
TM_TRACE=1 python tools/h2h_lane_trace.py 8 2> $OUT/ring_trace.txt | tail -3
