#!/bin/bash
# round 6, GPU call 15: the ring's first chunk and ramp once more, at the 48 MiB default
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe15; mkdir -p $OUT
S="python tools/h2h_sweep.py"
( $S 4:0
  TM_RING_FIRST_KIB=4096 $S 4:0
  TM_RING_FIRST_KIB=8192 $S 4:0
  TM_RING_FIRST_KIB=16384 $S 4:0
  TM_RING_FIRST_KIB=8192 TM_RING_RAMP=200 $S 4:0
  TM_RING_FIRST_KIB=4096 TM_RING_RAMP=200 $S 4:0
  TM_RING_FIRST_KIB=6144 TM_RING_RAMP=175 $S 4:0
  TM_RING_FIRST_KIB=8192 TM_RING_RAMP=130 $S 4:0 ) 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
