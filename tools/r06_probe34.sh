#!/bin/bash
# round 6, GPU call 34: counters of the filter pass and of the normalizer pass on the final tree (tools/norm_flags_time.py 256 under rocprofv3 --pmc, two passes)
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe34; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $OUT/g0 -o p --output-format csv -- python $ROOT/tools/norm_flags_time.py 256 > $OUT/run0.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $OUT/g1 -o p --output-format csv -- python $ROOT/tools/norm_flags_time.py 256 > $OUT/run1.txt 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
for g in ("g0","g1"):
    f=glob.glob(out+"/"+g+"/**/*counter_collection.csv", recursive=True)[0]
    rows=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0]
        if "k_pf_filter" not in k and "k_norm_emit2" not in k and "k_pf_long" not in k: continue
        rows.setdefault((k, r["Dispatch_Id"]), {})[r["Counter_Name"]]=float(r["Counter_Value"])
    last={}
    for (k,d),v in rows.items(): last[k]=v
    for k,v in last.items(): print(g, "%-34s"%k[-34:], {a:int(b) for a,b in v.items()})
PY
