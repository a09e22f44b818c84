#!/bin/bash
# round 6, GPU call 31: the ring takes vocabularies with byte-level flags (filter pass enqueued, normalizer pass over a bound): ring tests, normalizer tests, host to host with and without flags, the default bench line
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe31; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_host_api.py tests/test_gpu_parity.py -x -q -m gpu -k "ring or lossy or normalizer or stays_on_the_device or slabs" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
python $ROOT/tools/h2h_sweep.py 4:32 > $OUT/h2h.txt 2>&1
TM_SWEEP_FLAG=186 python $ROOT/tools/h2h_sweep.py 4:32 >> $OUT/h2h.txt 2>&1
TM_SWEEP_FLAG=255 python $ROOT/tools/h2h_sweep.py 4:32 >> $OUT/h2h.txt 2>&1
TM_SWEEP_FLAG=186 TM_RING=0 python $ROOT/tools/h2h_sweep.py 4:32 >> $OUT/h2h.txt 2>&1
cat $OUT/h2h.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.log; tail -c 1500 $OUT/bench.json
