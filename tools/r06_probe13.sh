#!/bin/bash
# round 6, GPU call 13: Latin Extended Additional on the device - parity, goldens, and what the normalizer pass costs now
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe13; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python tools/k1_time.py --e2e --mbytes 512 --reps 4 2>&1 | grep -v Warn | cut -c1-420
