#!/bin/bash
# tools/emu/tsan_host.sh — builds the library's sources for the host (tools/emu) together with tools/emu/tsan_host.cpp under
# -fsanitize=thread and runs it: data races in the host layer (lanes, worker pool, mailbox, block cache) are reported by ThreadSanitizer.
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
CXX=${TM_EMU_CXX:-/opt/rocm/lib/llvm/bin/clang++}
OUT=${1:-/tmp/tm_tsan}; mkdir -p "$OUT"
FLAGS="-O1 -g -std=c++17 -fsanitize=thread -fno-omit-frame-pointer -I $ROOT/tools/emu -I $ROOT/include -I $ROOT/tokenmonster_amd/csrc -Wno-unused-result -Wno-unknown-pragmas -Wno-pass-failed"
pids=()
for f in tm_vocab.hip tm_kernels.hip tm_score.hip tm_norm.hip tm_decode.hip tm_host.hip tm_decoder.hip tm_formats.hip tm_multi.hip tm_build.cpp tm_normalize.cpp; do
  $CXX $FLAGS -x c++ -c "$ROOT/tokenmonster_amd/csrc/$f" -o "$OUT/$f.o" & pids+=($!)
done
$CXX $FLAGS -x c++ -c "$ROOT/tools/emu/emu_runtime.cpp" -o "$OUT/emu_runtime.o" & pids+=($!)
$CXX $FLAGS -x c++ -c "$ROOT/tokenmonster_amd/testsupport/tm_synth.cpp" -o "$OUT/tm_synth.o" & pids+=($!)
$CXX $FLAGS -x c++ -c "$ROOT/tools/emu/tsan_host.cpp" -o "$OUT/tsan_host.o" & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
$CXX -fsanitize=thread -o "$OUT/tsan_host" "$OUT"/*.o -licuuc -licui18n -lz -lpthread -ldl
TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1 history_size=4" setarch "$(uname -m)" -R "$OUT/tsan_host"
