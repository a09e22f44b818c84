#!/bin/bash
# round 6, GPU call 16: the ring with K4 writing two-byte ids itself - host API tests, then the sweep
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe16; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_host_api.py -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1
grep -v "^  File" $OUT/pytest.log | tail -80 | cut -c1-300
python tools/h2h_sweep.py 4:0 2>&1 | grep -v Warning | tee $OUT/ring_sweep.txt
