#!/bin/bash
# round 6, GPU call 28: the filter pass of the byte-level normalizer flags, kernel by kernel (tools/norm_flags_time.py under rocprofv3 --kernel-trace --stats)
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe28; mkdir -p $OUT
python $ROOT/tools/norm_flags_time.py 256 > $OUT/plain.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -o p --output-format csv -- python $ROOT/tools/norm_flags_time.py 256 > $OUT/run.txt 2>&1
f=$(find $OUT/st -name '*kernel_stats.csv' | head -1)
head -25 "$f" > $OUT/kernel_stats_head.csv
cat $OUT/plain.txt; cat $OUT/kernel_stats_head.csv
