#!/bin/bash
# round 6, GPU call 33: characters NFD splits in three (Kannada, Sinhala) on the device: normalizer tests; tools/decode_scripts.py with Japanese / Hindi / Thai / Bengali + Tamil: normalize + tokenize + decode per GiB and the documents left to the host
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe33; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py -x -q -m gpu -k "hindi or japanese or multilingual or emoji or european or vietnamese or normalizer or lossy or golden or norm or slabs" > $OUT/pytest.txt 2>&1
tail -2 $OUT/pytest.txt
python tools/decode_scripts.py 64 > $OUT/scripts.txt 2>&1; cat $OUT/scripts.txt | cut -c1-250
python bench.py > $OUT/bench.json 2> $OUT/bench.log; python3 - <<'PY'
import json
j=json.loads(open('gpurun_out/r06_probe33/bench.json').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['config']['normalize_ms_per_step'], j['roofline']['kernel_ms'])
PY
