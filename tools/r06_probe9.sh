#!/bin/bash
# round 6, GPU call 9: k_seg_fill + the verdict folded into the serializer; the width of the one-workgroup scan under a saturating match kernel
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe9; mkdir -p $OUT
S="python tools/h2h_sweep.py"
( $S 4:32 4:48
  TM_SCAN1=256 $S 4:32 4:48
  TM_SCAN1=0 $S 4:32 4:48
  $S 4:32 4:48 ) 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
