#!/bin/bash
# round 6, GPU call 27: counters of k_dec_capcode per script (tools/decode_scripts.py): why text that mixes ASCII with a few characters beyond it decodes slower than all-Cyrillic text
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/r06_probe27; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES --kernel-trace -d $OUT/g0 -o p --output-format csv -- python $ROOT/tools/decode_scripts.py 64 > $OUT/run0.txt 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace -d $OUT/g1 -o p --output-format csv -- python $ROOT/tools/decode_scripts.py 64 > $OUT/run1.txt 2>&1
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
names=[l.split()[0] for l in open(out+"/run0.txt") if "MB encoded" in l]
for g in ("g0","g1"):
    f=glob.glob(out+"/"+g+"/**/*counter_collection.csv", recursive=True)[0]
    rows=collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        if "k_dec_capcode" not in r["Kernel_Name"]: continue
        rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]]=float(r["Counter_Value"])
    disp=list(rows.values())
    per=len(disp)//max(1,len(names))
    for i,n in enumerate(names):
        d=disp[i*per+per-1] if (i*per+per-1) < len(disp) else {}
        print(g, "%-18s"%n, {k:int(v) for k,v in d.items()})
PY
