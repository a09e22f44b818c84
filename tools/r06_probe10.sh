#!/bin/bash
# round 6, GPU call 10: K4's position-staging walk for the larger vocabularies and for the scoring pass - parity, then the shapes
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe10; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_golden.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
bash tools/bench_all_shapes.sh r06_probe10/shapes 2>&1 | tail -60
