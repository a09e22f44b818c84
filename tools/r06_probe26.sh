#!/bin/bash
# round 6, GPU call 26: per-phase cycles of k_match_branch (the -DTM_PHASE_TIMERS build, variants/phases): how long a wavefront waits at the workgroup's barrier
cd "$(dirname "$0")/.."
W=/tmp/phases_pkg; rm -rf $W; mkdir -p $W; cp -r tokenmonster_amd $W/; cp variants/phases/libtokenmonster_hip.so $W/tokenmonster_amd/libtokenmonster_hip.so
for cfg in englishcode-32000-consistent englishcode-100256-clean; do TM_PKG_ROOT=$W python tools/phase_profile.py $cfg 128 2>&1 | grep -v Warn; done | tee gpurun_out/r06_k1_phases.txt
