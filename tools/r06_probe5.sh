#!/bin/bash
# round 6, GPU call 5: the match kernels of consecutive chunks gated one behind the other
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe5; mkdir -p $OUT
S="python tools/h2h_sweep.py"
( $S 4:32 4:48
  TM_RING_GATE=0 $S 4:32
  TM_RING_STREAMS=3 $S 4:32 4:48
  TM_RING_STREAMS=3 TM_RING_SLOTS=6 $S 4:32
  TM_RING_STREAMS=4 TM_RING_SLOTS=6 $S 4:32 4:16
  TM_RING_STREAMS=3 TM_RING_SLOTS=5 $S 4:24 ) 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
TM_RING_STREAMS=3 TM_TRACE=1 python tools/h2h_lane_trace.py 8 2> $OUT/ring_trace.txt | tail -3
