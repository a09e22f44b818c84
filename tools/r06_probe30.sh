#!/bin/bash
# round 6, GPU call 30: the filter pass with spans of 4 KiB for long documents: parity, timing, kernel stats; host to host with the example flags (lanes' form) beside the ring and the lanes' form without flags
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe30; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lossy or normalizer or stays_on_the_device or slabs" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
python $ROOT/tools/norm_flags_time.py 256 > $OUT/plain.txt 2>&1
python $ROOT/tools/h2h_sweep.py 4:32 > $OUT/h2h.txt 2>&1
TM_RING=0 python $ROOT/tools/h2h_sweep.py 4:32 >> $OUT/h2h.txt 2>&1
TM_SWEEP_FLAG=186 python $ROOT/tools/h2h_sweep.py 4:32 >> $OUT/h2h.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -o p --output-format csv -- python $ROOT/tools/norm_flags_time.py 256 > $OUT/run.txt 2>&1
f=$(find $OUT/st -name '*kernel_stats.csv' | head -1)
python3 - "$f" > $OUT/kernel_stats_head.txt <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-60s calls %4s avg %10.1f us  %5s %%" % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs'])/1e3, r['Percentage']))
PY
cat $OUT/plain.txt $OUT/h2h.txt $OUT/kernel_stats_head.txt
