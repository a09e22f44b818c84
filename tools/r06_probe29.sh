#!/bin/bash
# round 6, GPU call 29: the second form of the filter pass (a wavefront per document, a dword per lane): parity tests, tools/norm_flags_time.py plain and under rocprofv3 --kernel-trace --stats
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe29; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lossy or normalizer or stays_on_the_device or slabs" > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k norm > $OUT/pytest_fuzz.txt 2>&1
tail -3 $OUT/pytest_fuzz.txt
python $ROOT/tools/norm_flags_time.py 256 > $OUT/plain.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -o p --output-format csv -- python $ROOT/tools/norm_flags_time.py 256 > $OUT/run.txt 2>&1
f=$(find $OUT/st -name '*kernel_stats.csv' | head -1)
head -12 "$f" | cut -c1-200 > $OUT/kernel_stats_head.csv
cat $OUT/plain.txt; cat $OUT/kernel_stats_head.csv
