#!/bin/bash
# round 6, GPU call 1: the host link of the box, and the timeline of the host-to-host pipeline as it stands
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe1
mkdir -p $OUT
timeout 300 tools/build/pcie_probe > $OUT/pcie_probe.txt 2>&1
cat $OUT/pcie_probe.txt
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/$OUT/h2h_tl -o t -- python $R/tools/h2h_trace.py --lanes 4 --chunk-mib 32 --passes 10 > $R/$OUT/h2h_passes.txt 2> $R/$OUT/h2h_trace.err)
cat $OUT/h2h_passes.txt
python tools/h2h_trace.py --analyze $OUT/h2h_tl --head 5 > $OUT/h2h_analysis.txt 2>&1
head -60 $OUT/h2h_analysis.txt
rm -rf $OUT/h2h_tl
