#!/bin/bash
# round 6, GPU call 23: the ring's passes come in two lengths on some boxes (26.9 / 30.5 ms): the per-chunk trace of fast and slow passes side by side
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe23; mkdir -p $OUT
TM_TRACE=1 python tools/h2h_sweep.py 4:0 > $OUT/sweep.txt 2> $OUT/trace.txt
cat $OUT/sweep.txt | grep -v Warn
python - $OUT/trace.txt <<'PY'
import re,sys
passes=[]; cur=None
for line in open(sys.argv[1]):
    if line.startswith("[pipe]") and "laid out" in line:
        cur={"issue":[], "computed":[], "complete":[]}; passes.append(cur); continue
    if cur is None: continue
    m=re.match(r"\[ring\] issue chunk\s+(\d+) \(\s*([\d.]+) MiB.*slot (\d+) at\s+([\d.]+) ms,\s+([\d.]+) ms of launches", line)
    if m: cur["issue"].append((int(m.group(1)), float(m.group(2)), float(m.group(4)), float(m.group(5)))); continue
    m=re.match(r"\[ring\] chunk\s+(\d+) computed at\s+([\d.]+) ms", line)
    if m: cur["computed"].append((int(m.group(1)), float(m.group(2)))); continue
    m=re.match(r"\[ring\] chunk\s+(\d+) complete at\s+([\d.]+) ms", line)
    if m: cur["complete"].append((int(m.group(1)), float(m.group(2))))
print(len(passes), "passes traced")
ends=[p["complete"][-1][1] if p["complete"] else 0 for p in passes]
print("ends:", [round(e,2) for e in ends[-14:]])
last=passes[-14:]
fast=min(last, key=lambda p: p["complete"][-1][1]); slow=max(last, key=lambda p: p["complete"][-1][1])
print("chunk  MiB | fast: issued  computed  complete | slow: issued  computed  complete | d(computed)")
fc=dict(fast["computed"]); sc=dict(slow["computed"]); fp=dict(fast["complete"]); sp=dict(slow["complete"])
si={k:(a,b,c) for k,a,b,c in slow["issue"]}
for k,mib,at,ln in fast["issue"]:
    print("%4d %6.1f | %8.2f %8.2f %8.2f | %8.2f %8.2f %8.2f | %+6.2f" % (k, mib, at, fc.get(k,0), fp.get(k,0), si.get(k,(0,0,0))[1], sc.get(k,0), sp.get(k,0), sc.get(k,0)-fc.get(k,0)))
PY
