#!/bin/bash
# round 6, GPU call 14: tm_score_multi with the exit states chained on the device - the multi-device tests, then one rank's share of the 8-GPU pass
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe14; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_zz_dist_ranks.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
timeout 900 python tools/score_rank_protocol.py > $OUT/score_rank_protocol.json 2> $OUT/score_rank_protocol.err; tail -40 $OUT/score_rank_protocol.json
