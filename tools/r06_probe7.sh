#!/bin/bash
# round 6, GPU call 7: the ring's ramp and chunk size
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe7; mkdir -p $OUT
S="python tools/h2h_sweep.py"
( $S 4:32 4:40 4:48 4:56 4:64
  TM_RING_RAMP=200 $S 4:32 4:48
  TM_RING_RAMP=130 $S 4:32 4:48
  TM_RING_FIRST_KIB=1024 $S 4:32 4:48
  TM_RING_FIRST_KIB=4096 $S 4:32 4:48
  TM_RING_FIRST_KIB=4096 TM_RING_RAMP=130 $S 4:48
  TM_RING_SLOTS=6 $S 4:32 4:48 ) 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
TM_TRACE=1 python tools/h2h_lane_trace.py 8 2> $OUT/ring_trace.txt | tail -2
