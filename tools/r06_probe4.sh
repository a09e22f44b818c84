#!/bin/bash
# round 6, GPU call 4: the ring with eight hardware queues (the library's default now) and the halving tail
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe4; mkdir -p $OUT
S="python tools/h2h_sweep.py"
( $S 4:32 4:24 4:48 4:64
  GPU_MAX_HW_QUEUES=4 $S 4:32
  GPU_MAX_HW_QUEUES=6 $S 4:32
  GPU_MAX_HW_QUEUES=12 $S 4:32
  GPU_MAX_HW_QUEUES=16 $S 4:32
  TM_RING_STREAMS=3 $S 4:32
  TM_RING_STREAMS=3 TM_RING_SLOTS=6 $S 4:32
  TM_RING_STREAMS=4 TM_RING_SLOTS=6 $S 4:32
  TM_RING_SLOTS=3 $S 4:32
  TM_RING=0 $S 4:32 ) 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
TM_TRACE=1 python tools/h2h_lane_trace.py 8 2> $OUT/ring_trace.txt | tail -3
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$OUT/h2h_tl -o t -- python $R/tools/h2h_trace.py --lanes 4 --chunk-mib 32 --passes 10 > $R/$OUT/h2h_passes.txt 2> $R/$OUT/h2h_trace.err)
tail -3 $OUT/h2h_passes.txt
python tools/h2h_trace.py --analyze $OUT/h2h_tl --head 4 > $OUT/h2h_analysis.txt 2>&1
head -30 $OUT/h2h_analysis.txt
rm -rf $OUT/h2h_tl
