#!/bin/bash
# The first GPU call of round 3 (round 2 ended without GPU minutes: everything after commit c275605 has only run on the emulated device).
#   (1) what the driver checks, on the code as it is: pytest -m gpu, smoke, default bench, kernel stats        -> gpurun_out/r03_start
#   (2) A/B of the K1 variants built by `bash tools/variant_ab.sh build` (run that HERE first, then gpurun)    -> gpurun_out/variant_ab
cd "$(dirname "$0")/.."
bash tools/end_of_round_check.sh r03_start
bash tools/variant_ab.sh run 256      # (seconds; `check <name>` runs the parity tests with one variant in place)
