#!/bin/bash
# tools/emu/isa_same.sh <git-rev>: is the gfx950 code of the kernel units at <git-rev> the same as in the working tree?  (A change that is
# meant to be source-only — macro wrapping for tools/emu, comments, host code — must leave every instruction where it was; only the
# __hip_cuid_* symbol, a hash of the source text, may differ.)
set -e
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
REV=${1:-HEAD}
OLD=$(mktemp -d) ; NEW=$(mktemp -d)
git -C "$ROOT" archive "$REV" tokenmonster_amd/csrc include | tar -x -C "$OLD"
rc=0
for f in tm_kernels tm_norm tm_decode; do
  ( cd "$OLD/tokenmonster_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ../../include -I . -x hip $f.hip --cuda-device-only -S -o "$OLD/$f.s" 2>/dev/null ) &
  ( cd "$ROOT/tokenmonster_amd/csrc" && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I ../../include -I . -x hip $f.hip --cuda-device-only -S -o "$NEW/$f.s" 2>/dev/null ) &
done
wait
for f in tm_kernels tm_norm tm_decode; do
  if diff <(grep -v __hip_cuid_ "$OLD/$f.s") <(grep -v __hip_cuid_ "$NEW/$f.s") > /dev/null; then echo "$f: identical"; else echo "$f: DIFFERS"; rc=1; fi
done
rm -rf "$OLD" "$NEW"
exit $rc
