#!/bin/bash
# round 6, GPU call 20: instruction and wait counters of the decode kernels
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe20; mkdir -p $OUT
python tools/pmc_profile.py --mbytes 256 --groups 0,1,2 --kernel k_dec --extra "--workload decode" --e2e --out $OUT/pmc > $OUT/pmc_decode.json 2> $OUT/pmc_decode.err
cat $OUT/pmc_decode.json | head -80; tail -5 $OUT/pmc_decode.err
