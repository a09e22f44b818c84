#!/bin/bash
# round 6, GPU call 22: engine timeline of the host-to-host ring (uploads, kernels, downloads of neighbouring chunks side by side)
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe22; mkdir -p $OUT
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/trace -o t -- python $ROOT/tools/h2h_trace.py --lanes 4 --chunk-mib 0 --passes 8 > $OUT/passes.txt 2> $OUT/trace.err)
cat $OUT/passes.txt | tail -4
python tools/h2h_trace.py --analyze $OUT/trace --head 4 > $OUT/h2h_timeline.txt 2>&1; cat $OUT/h2h_timeline.txt | head -60
find $OUT/trace -name "*.csv" -size +2M -delete
