#!/bin/bash
# round 6, GPU call 32: the voiced kana, the three-byte marks of canonical class > 0 and the three-byte digits on the device: the normalizer / golden / ring / fuzz tests, Japanese text through tools/decode_scripts.py-like timing, the default bench line
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe32; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_host_api.py tests/test_gpu_fuzz.py -x -q -m gpu -k "hindi or japanese or multilingual or emoji or european or vietnamese or normalizer or lossy or golden or ring or norm or slabs" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
python bench.py > $OUT/bench.json 2> $OUT/bench.log; cut -c1-400 $OUT/bench.json
