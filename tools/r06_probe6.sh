#!/bin/bash
# round 6, GPU call 6: the device's timeline in the ring's steady state, two and three compute streams
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe6; mkdir -p $OUT
R=$PWD
for ns in 2 3; do
(cd /tmp && export TMPDIR=/tmp && TM_RING_STREAMS=$ns timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$OUT/tl$ns -o t -- python $R/tools/h2h_trace.py --lanes 4 --chunk-mib 32 --passes 10 > $R/$OUT/passes$ns.txt 2> $R/$OUT/trace$ns.err)
tail -2 $OUT/passes$ns.txt
python tools/h2h_trace.py --analyze $OUT/tl$ns --window 10 13.5 > $OUT/analysis$ns.txt 2>&1
head -3 $OUT/tl$ns/*/*memory_copy_trace.csv | cut -c1-300
rm -rf $OUT/tl$ns
done
