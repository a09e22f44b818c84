for dbg in 0 1 4 8 16 8388608; do
  echo "TM_DBG=$dbg"
  TM_DBG=$dbg python tools/pmc_profile.py --fast --mbytes 128 --kernel k_match_branch --groups 0 --lib variants/devel/libtokenmonster_hip.so --out gpurun_out/pmc_ab_$dbg 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); k=list(d.values())[0]; w=k['SQ_WAVES']
print('  per segment: SALU %.0f VALU %.0f LDS %.0f VMEM %.0f BRANCH %.0f  model cycles %.0f' % (k['SQ_INSTS_SALU']/w, k['SQ_INSTS_VALU']/w, k['SQ_INSTS_LDS']/w, (k['SQ_INSTS_VMEM_RD']+k['SQ_INSTS_VMEM_WR'])/w, k['SQ_INSTS_BRANCH']/w, k['SQ_INSTS_SALU']/w + k['SQ_INSTS_VALU']/w/2))"
done
