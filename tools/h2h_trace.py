#!/usr/bin/env python3
"""Host-to-host pipeline under a timeline: run as
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- python tools/h2h_trace.py [--lanes L --chunk-mib C]
and feed DIR to `tools/h2h_trace.py --analyze DIR`: GPU-busy time (union of kernel intervals), copy-engine busy time per
direction, and the wall span of the last pipeline pass.  Development aid for tm_tokenize_pipeline (tm_host.hip)."""
import argparse
import csv
import glob
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def union(iv):
    iv.sort()
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        if cur_b is None or a > cur_b:
            if cur_b is not None:
                tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None:
        tot += cur_b - cur_a
    return tot


def analyze(d, head_ms=0.0, win=None):
    kf = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    mf = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
    ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]) for f in kf for r in csv.DictReader(open(f))]
    kq = {(int(r["Start_Timestamp"]), int(r["End_Timestamp"])): r.get("Queue_Id", r.get("Stream_Id", "?")) for f in kf for r in csv.DictReader(open(f))}
    ms = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"], int(r.get("Bytes", 0) or 0)) for f in mf for r in csv.DictReader(open(f))]
    if not ks:
        print("no kernel trace under", d)
        return
    # the last pass = everything after the last gap of > 20 ms without any kernel
    ks.sort()
    cut = ks[0][0]
    for (a0, b0, _), (a1, b1, _) in zip(ks, ks[1:]):
        if a1 - b0 > 20_000_000:
            cut = a1
    kl = [k for k in ks if k[0] >= cut]
    lo, hi = kl[0][0], max(k[1] for k in kl)
    ml = [m for m in ms if m[1] >= lo - 30_000_000 and m[0] <= hi + 30_000_000]
    if ml:
        lo = min(lo, min(m[0] for m in ml if m[0] >= lo - 30_000_000))
        hi = max(hi, max(m[1] for m in ml))
    print("last pass: span %.2f ms, %d kernels, %d copies" % ((hi - lo) / 1e6, len(kl), len(ml)))
    print("  GPU busy (union of kernels)  %.2f ms ; sum of kernel durations %.2f ms" % (union([(a, b) for a, b, _ in kl]) / 1e6, sum(b - a for a, b, _ in kl) / 1e6))
    for dirn in sorted(set(m[2] for m in ml)):
        sel = [m for m in ml if m[2] == dirn]
        big = [m for m in sel if m[3] >= 1 << 20]
        by = sum(m[3] for m in sel)
        print("  copies %-22s n %4d  bytes %8.1f MB  busy %.2f ms  (>=1 MiB: n %d, %.1f GB/s while active)" % (
            dirn, len(sel), by / 1e6, union([(m[0], m[1]) for m in sel]) / 1e6, len(big),
            (sum(m[3] for m in big) / max(1, union([(m[0], m[1]) for m in big]))) if big else 0.0))
    agg = {}
    for a, b, n in kl:
        t = agg.setdefault(n, [0, 0])
        t[0] += 1
        t[1] += b - a
    for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1])[:12]:
        print("    %-40s calls %4d  total %8.2f ms" % (n[-40:], c, t / 1e6))
    # idle gaps of the compute queue inside the span
    iv = sorted((a, b) for a, b, _ in kl)
    gaps, cur = [], iv[0][1]
    for a, b in iv[1:]:
        if a > cur:
            gaps.append(a - cur)
        cur = max(cur, b)
    gaps.sort(reverse=True)
    print("  idle gaps between kernels: total %.2f ms, largest %s ms" % (sum(gaps) / 1e6, [round(g / 1e6, 2) for g in gaps[:8]]))
    # the engines side by side, per millisecond of the pass: share of the millisecond in which an upload / a kernel / a download was running
    def busy_in(iv, a, b):
        return union([(max(x, a), min(y, b)) for x, y in iv if y > a and x < b]) / max(1, b - a)
    h2d = [(m[0], m[1]) for m in ml if "HOST_TO_DEVICE" in m[2].upper() or "H2D" in m[2].upper()]
    d2h = [(m[0], m[1]) for m in ml if "DEVICE_TO_HOST" in m[2].upper() or "D2H" in m[2].upper()]
    kiv = [(a, b) for a, b, _ in kl]
    print("  timeline (1 ms bins; tenths of the bin busy: upload | kernels | download):")
    line_u, line_k, line_d = "", "", ""
    nb = int((hi - lo) / 1e6) + 1
    for i in range(nb):
        a, b = lo + i * 1_000_000, min(hi, lo + (i + 1) * 1_000_000)
        if b <= a:
            break
        f = lambda x: "#" if x >= 0.95 else str(int(x * 10))
        line_u += f(busy_in(h2d, a, b)); line_k += f(busy_in(kiv, a, b)); line_d += f(busy_in(d2h, a, b))
    print("    upload   %s\n    kernels  %s\n    download %s" % (line_u, line_k, line_d))
    # how much of the span two engines run side by side
    def inter(x, y):
        ev = sorted([(a, 1, 0) for a, b in x] + [(b, -1, 0) for a, b in x] + [(a, 0, 1) for a, b in y] + [(b, 0, -1) for a, b in y])
        cx = cy = 0; last = None; tot = 0
        for t, dx, dy in ev:
            if last is not None and cx > 0 and cy > 0:
                tot += t - last
            cx += dx; cy += dy; last = t
        return tot
    span = hi - lo
    print("  side by side: upload & kernels %.2f ms, kernels & download %.2f ms, upload & download %.2f ms of a span of %.2f ms; kernels busy %.0f %%, upload %.0f %%, download %.0f %%" % (
        inter(h2d, kiv) / 1e6, inter(kiv, d2h) / 1e6, inter(h2d, d2h) / 1e6, span / 1e6, 100 * union(list(kiv)) / span, 100 * union(list(h2d)) / span, 100 * union(list(d2h)) / span))
    if win:
        # everything the device did between win[0] and win[1] ms of the pass, one line per kernel / copy, with its queue
        ev = [(a, b, "q%s %s" % (kq.get((a, b), "?"), n[-34:])) for a, b, n in kl] + [(m[0], m[1], "      %s %.2f MB" % (m[2][-14:], m[3] / 1e6)) for m in ml]
        for a, b, what in sorted(ev):
            if (b - lo) / 1e6 >= win[0] and (a - lo) / 1e6 <= win[1]:
                print("    %8.3f ms  +%7.3f  %s" % ((a - lo) / 1e6, (b - a) / 1e6, what))
    if head_ms > 0:
        # what the device did in the first head_ms of the pass: every kernel (queue, name) and copy, in order of their start
        ev = [(a, b, "q%s %s" % (kq.get((a, b), "?"), n[-34:])) for a, b, n in kl] + [(m[0], m[1], "%s %.2f MB" % (m[2], m[3] / 1e6)) for m in ml]
        for a, b, what in sorted(ev):
            if a - lo > head_ms * 1e6:
                break
            if b - a >= 15_000 or "MB" in what:           # (kernels of 15 us and more; all copies)
                print("    %8.3f ms  +%7.3f  %s" % ((a - lo) / 1e6, (b - a) / 1e6, what))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--analyze")
    ap.add_argument("--head", type=float, default=0.0, help="with --analyze: list what ran in the first HEAD ms of the last pass")
    ap.add_argument("--window", type=float, nargs=2, help="with --analyze: list what ran between A and B ms of the last pass")
    ap.add_argument("--lanes", type=int, default=3)
    ap.add_argument("--chunk-mib", type=int, default=64)
    ap.add_argument("--mbytes", type=int, default=1024)
    ap.add_argument("--passes", type=int, default=3)
    a = ap.parse_args()
    if a.analyze:
        return analyze(a.analyze, a.head, a.window)
    import numpy as np
    import tokenmonster_amd as tm
    from tokenmonster_amd import synth
    cfg = "englishcode-32000-consistent"
    kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[cfg]
    vocab = tm.Vocab(synth.config_vocab(cfg))
    raw, roffs = synth.synth_corpus(kind, a.mbytes << 20, seed=0x434F5250 + 2)
    pin_in = tm.PinnedBuffer(raw.size)
    pin_in.array[:] = raw
    pin_out = tm.PinnedBuffer(raw.size + 4096)
    for i in range(a.passes):
        if i == a.passes - 1:
            time.sleep(0.1)          # a gap the analysis can find
        t0 = time.perf_counter()
        blob, boff, _, enc, st = vocab.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=a.chunk_mib << 20, lanes=a.lanes, out=pin_out.array)
        dt = time.perf_counter() - t0
        print("pass %d: %.2f ms  %.2f GB/s  (%d ids, lanes %d, chunk %d MiB)" % (i, dt * 1e3, raw.size / dt / 1e9, int(boff[-1]) // enc, a.lanes, a.chunk_mib), flush=True)


if __name__ == "__main__":
    main()
