#!/bin/bash
# round 6, GPU call 25: on some boxes the ring's passes come in two lengths (26.9 / 30.5 ms): if this is such a box, the per-chunk trace of a fast and a slow pass side by side
cd "$(dirname "$0")/.."
OUT=gpurun_out/r06_probe25; mkdir -p $OUT
TM_TRACE=1 python tools/h2h_sweep.py 4:0 > $OUT/sweep.txt 2> $OUT/trace.txt
grep -v Warn $OUT/sweep.txt
python - $OUT/trace.txt <<'PY'
import re,sys
passes=[]; cur=None
for line in open(sys.argv[1]):
    if line.startswith("[pipe]") and "laid out" in line:
        cur={"issue":[], "computed":[], "complete":[]}; passes.append(cur); continue
    if cur is None: continue
    m=re.match(r"\[ring\] issue chunk\s+(\d+) \(\s*([\d.]+) MiB.*slot (\d+) at\s+([\d.]+) ms,\s+([\d.]+) ms of launches", line)
    if m: cur["issue"].append((int(m.group(1)), float(m.group(2)), float(m.group(4)), float(m.group(5)), int(m.group(3)))); continue
    m=re.match(r"\[ring\] chunk\s+(\d+) computed at\s+([\d.]+) ms", line)
    if m: cur["computed"].append((int(m.group(1)), float(m.group(2)))); continue
    m=re.match(r"\[ring\] chunk\s+(\d+) complete at\s+([\d.]+) ms", line)
    if m: cur["complete"].append((int(m.group(1)), float(m.group(2))))
last=[p for p in passes[-10:] if p["complete"]]
ends=[p["complete"][-1][1] for p in last]
print("ends of the last passes:", [round(e,2) for e in ends])
fast=min(last, key=lambda p: p["complete"][-1][1]); slow=max(last, key=lambda p: p["complete"][-1][1])
if slow["complete"][-1][1] - fast["complete"][-1][1] < 1.5:
    print("one length on this box"); sys.exit(0)
print("chunk  MiB | fast: issued (launch ms, slot) computed complete | slow: issued (launch ms, slot) computed complete | d(computed)")
fc=dict(fast["computed"]); sc=dict(slow["computed"]); fp=dict(fast["complete"]); sp=dict(slow["complete"])
si={k:(a,b,c,d) for k,a,b,c,d in slow["issue"]}
for k,mib,at,ln,slot in fast["issue"]:
    s=si.get(k,(0,0,0,0))
    print("%4d %6.1f | %7.2f (%.2f, %d) %7.2f %7.2f | %7.2f (%.2f, %d) %7.2f %7.2f | %+6.2f" % (k, mib, at, ln, slot, fc.get(k,0), fp.get(k,0), s[1], s[2], s[3], sc.get(k,0), sp.get(k,0), sc.get(k,0)-fc.get(k,0)))
PY
