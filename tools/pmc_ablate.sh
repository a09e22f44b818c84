#!/bin/bash
# tools/pmc_ablate.sh — ON THE GPU BOX: instruction counters of k_match_branch phase by phase - the -DTM_DEVEL build (bash tools/variant_ab.sh build,
# variants/devel) with phases switched off through TM_DBG (results WRONG by design: 1 the walks of step A1, 4 hash probes, 8 forward-delete probes, 16 exit
# maps, 8388608 greedy step B), one rocprofv3 --pmc pass each (tools/pmc_profile.py --fast).  profiles/r05_issue_model.txt (2).
for dbg in 0 1 4 8 16 8388608; do
  echo "TM_DBG=$dbg"
  TM_DBG=$dbg python tools/pmc_profile.py --fast --mbytes 128 --kernel k_match_branch --groups 0 --lib variants/devel/libtokenmonster_hip.so --out gpurun_out/pmc_ab_$dbg 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=list(d.values())[0]; w=k['SQ_WAVES']
print('  per segment: SALU %.0f VALU %.0f LDS %.0f VMEM %.0f BRANCH %.0f  model cycles %.0f' % (k['SQ_INSTS_SALU']/w, k['SQ_INSTS_VALU']/w, k['SQ_INSTS_LDS']/w, (k['SQ_INSTS_VMEM_RD']+k['SQ_INSTS_VMEM_WR'])/w, k['SQ_INSTS_BRANCH']/w, k['SQ_INSTS_SALU']/w + k['SQ_INSTS_VALU']/w/2))"
done
