#!/bin/bash
# tools/isa_stats.sh <file.hip> <kernel-name-substring>: static instruction mix of one gfx950 kernel (development aid)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p /tmp/isa
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -I "$ROOT/include" -I "$ROOT/tokenmonster_amd/csrc" -x hip "$ROOT/tokenmonster_amd/csrc/$1" --cuda-device-only -S -o /tmp/isa/out.s 2>/dev/null
sym=$(grep -o "^_Z[A-Za-z0-9_]*$2[A-Za-z0-9_]*:" /tmp/isa/out.s | head -1 | tr -d ':')
awk -v s="$sym" '$0 ~ "^"s":" {p=1} p {print} p && /^\.Lfunc_end/ {exit}' /tmp/isa/out.s > /tmp/isa/kernel.s
echo "kernel $sym: $(wc -l < /tmp/isa/kernel.s) lines -> /tmp/isa/kernel.s"
echo "VALU $(grep -c '^\s*v_' /tmp/isa/kernel.s)  SALU $(grep -c '^\s*s_' /tmp/isa/kernel.s)  readlane/writelane $(grep -c 'v_readlane\|v_writelane' /tmp/isa/kernel.s)  LDS $(grep -c '^\s*ds_' /tmp/isa/kernel.s)  VMEM $(grep -c '^\s*global_\|^\s*buffer_' /tmp/isa/kernel.s)"
grep -A16 "\.name: *$sym" /tmp/isa/out.s | grep -i "vgpr_count\|sgpr_count\|spill_count\|group_segment_fixed_size" | tr -s ' ' | tr '\n' ' '; echo
