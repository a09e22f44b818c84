#!/bin/bash
# tools/final_gpu_short.sh — ON THE GPU BOX, when GPU minutes are short: the default bench line, kernel stats of the same step, the test files that are not picked by
# tools/end_of_round_check.sh's full run, 20 s of device fuzz.  Output in gpurun_out/final_r05d.
cd "$(dirname "$0")/.." 2>/dev/null || true
ROOT=$PWD; OUT=gpurun_out/final_r05d; mkdir -p $OUT
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json; grep -E "host_to_host" $OUT/bench_default.err | cut -c1-400
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/stats_e2e -o s --output-format csv -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-to-host --verify 0 > $ROOT/$OUT/bench_under_rocprof.json 2>/dev/null)
f=$(find $ROOT/$OUT/stats_e2e -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/kernel_stats_e2e.csv && head -7 $f | cut -c1-40,150-260
find $OUT -name "*kernel_trace.csv" -delete
timeout 200 python -m pytest tests/test_gpu_golden.py tests/test_gpu_host_api.py tests/test_gpu_cpp_port_shim.py tests/test_gpu_server_jobs.py -x -q -m gpu 2>&1 | tail -2 | tee $OUT/pytest_subset.txt
python tools/gpu_fuzz.py 20 860001 2>&1 | tail -1 | tee $OUT/gpu_fuzz.txt
