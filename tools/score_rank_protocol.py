#!/usr/bin/env python3
"""tools/score_rank_protocol.py [--mbytes 1024] [--ranks 8] — ON THE GPU BOX: the 8-GPU scoring target of BASELINE.json (>= 6 x on configs[4]: 1 GiB
dataset, 65 536-id candidate set, one RCCL all-reduce per pass) PROJECTED from what one GPU can measure.  No N-GPU node has been available to any
round; what CAN be measured is one rank's share of the pass through the real protocol, stage by stage:

  t_full   tm_score of the whole dataset as one strip on one GPU (the N = 1 pass)
  t_rank   the pass of ONE of `ranks` ranks: 1/ranks of the dataset + 128 bytes of halo, continues = 1:
             begin    tm_score_begin   match kernel over the range + exit states composed on the device + D2H of the 80 bytes (returns when the host has them)
             chain    the host's walk through the exits of the ranks before it (ranks - 1 table look-ups)
             finish   tm_score_finish  resolve + histogram walk, enqueued
             reduce   ncclAllReduce(sum, uint32) of the n_ids + 260 words - on ONE rank here (TM_RCCL=1 on a one-member handle: RCCL's own launch and
                      kernel, without the seven peers' links), through tm_score_multi, which also holds the stream sync and the read of the result
  projected_scaling = t_full / (t_rank + allreduce_extra) for allreduce_extra = 0 (what this box measures) and for 50 / 100 us (what a 263 KB all-reduce
                      over eight xGMI peers may add to the one-rank call: latency-bound, 7 ring steps of 33 KB)

torch-free (ctypes on the C ABI).  The histogram of the rank pass is checked against tm_score over the same range as a strip of the whole dataset."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbytes", type=int, default=1024)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--config", default="candidates-65536")
    a = ap.parse_args()
    os.environ["TM_RCCL"] = "1"              # a one-member handle runs the collective anyway
    import numpy as np
    import tokenmonster_amd as tm
    from tokenmonster_amd import _native as N, synth, multi
    N.check(N.lib.tm_set_device(0))
    kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[a.config]
    img = synth.config_vocab(a.config)
    v = tm.Vocab(img)
    raw, roffs = synth.synth_corpus(kind, a.mbytes << 20, seed=0x434F5250 + 5)
    text, _ = synth.normalize_batch(raw, roffs, capcode, norm_flag)
    del raw
    data = np.ascontiguousarray(text)
    n = int(data.size)
    n_ids = v.n_ids()
    words = n_ids + 4 + 256

    def clock(fn, reps):
        fn()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0], ts[-1]

    # ---- N = 1: the whole dataset as one strip
    ds = C.c_void_p()
    N.check(N.lib.tm_dataset_upload(N.ptr(data), n, C.byref(ds)))
    full = np.zeros(n_ids, dtype=np.uint32); tit = C.c_uint64(); ms8 = np.zeros(32, dtype=np.uint8)
    t_full = clock(lambda: N.check(N.lib.tm_score(v.handle, ds, None, None, 0, N.ptr(full), C.byref(tit), N.ptr(ms8))), a.reps)
    # the range of a middle rank, as a strip of the whole (the reference for the rank pass below is the same bytes walked from their true entry state:
    # a strip starts in state 0, so the comparison is made on the single-device protocol with the TRUE entry state, computed from the ranges before)
    per = (n // a.ranks) // 4 * 4
    r = a.ranks // 2
    lo, hi = per * r, per * (r + 1)
    N.lib.tm_dataset_free(ds)

    # ---- one rank's pass, stage by stage (single-device entry points)
    part = np.ascontiguousarray(data[lo:hi + 128])
    dr = C.c_void_p()
    N.check(N.lib.tm_dataset_upload(N.ptr(part), int(part.size), C.byref(dr)))
    exits = np.zeros(80, dtype=np.uint8)
    # the true entry state of rank r: chain the exit maps of the ranges before it (each computed on its own bytes)
    entry = 0
    t_chain_all = time.perf_counter()
    for k in range(r):
        pk = np.ascontiguousarray(data[per * k: per * (k + 1) + 128])
        dk = C.c_void_p()
        N.check(N.lib.tm_dataset_upload(N.ptr(pk), int(pk.size), C.byref(dk)))
        ek = np.zeros(80, dtype=np.uint8)
        N.check(N.lib.tm_score_begin(v.handle, dk, 0, per, 1, None, N.ptr(ek)))
        entry = int(ek[entry])
        N.lib.tm_dataset_free(dk)
    got = np.zeros(n_ids, dtype=np.uint32); tit_r = C.c_uint64(); ms_r = np.zeros(32, dtype=np.uint8)
    st = {"begin": [], "finish_enqueue": [], "read": []}
    for it in range(a.reps + 1):
        t0 = time.perf_counter()
        N.check(N.lib.tm_score_begin(v.handle, dr, 0, hi - lo, 1, None, N.ptr(exits)))
        t1 = time.perf_counter()
        N.check(N.lib.tm_score_finish(v.handle, dr, entry, None, None, 0))
        t2 = time.perf_counter()
        N.check(N.lib.tm_score_read(v.handle, dr, N.ptr(got), C.byref(tit_r), N.ptr(ms_r)))
        t3 = time.perf_counter()
        if it:
            st["begin"].append((t1 - t0) * 1e3); st["finish_enqueue"].append((t2 - t1) * 1e3); st["read"].append((t3 - t2) * 1e3)
    stages = {k: round(float(np.median(x)), 4) for k, x in st.items()}
    t0 = time.perf_counter()
    e = 0
    chain = [exits] * (a.ranks - 1)
    for k in range(a.ranks - 1):
        e = int(chain[k][e]) if int(chain[k][e]) < 80 else 0
    stages["chain_host"] = round((time.perf_counter() - t0) * 1e3, 4)
    N.lib.tm_dataset_free(dr)

    # ---- the same pass through the library's own driver on a one-member handle, RCCL all-reduce of `words` uint32 included (TM_RCCL=1)
    g = multi.Devices([0])
    vs = multi.VocabSet(g, img)
    dset = multi.DatasetSet(g, part[: hi - lo])          # (one member: its range is the whole upload, entry state 0 - the same work, not the same histogram)
    ranks, why = g.rccl_ranks()
    t_multi = clock(lambda: dset.score(vs), a.reps)
    os.environ["TM_RCCL"] = "0"
    g2 = multi.Devices([0]); vs2 = multi.VocabSet(g2, img); d2 = multi.DatasetSet(g2, part[: hi - lo])
    t_multi_nor = clock(lambda: d2.score(vs2), a.reps)
    d2.close(); vs2.close(); g2.close()
    dset.close(); vs.close(); g.close()

    t_rank = stages["begin"] + stages["chain_host"] + stages["finish_enqueue"] + stages["read"]
    out = {"config": a.config, "n_ids": n_ids, "dataset_bytes": n, "ranks": a.ranks, "range_bytes": hi - lo, "histogram_words": words,
           "t_full_ms": {"median": round(t_full[0], 3), "min": round(t_full[1], 3), "max": round(t_full[2], 3)},
           "rank_stages_ms": stages, "t_rank_ms_single_device_protocol": round(t_rank, 3),
           "t_rank_ms_tm_score_multi_one_member": {"with_rccl_allreduce_1rank": round(t_multi[0], 3), "without_collective": round(t_multi_nor[0], 3),
                                                   "rccl_ranks": ranks, "rccl_note": why},
           "allreduce_cost_1rank_ms": round(t_multi[0] - t_multi_nor[0], 3)}
    # round 6: tm_score_multi keeps the exit states on the device (all-gather + chain kernel on the member's stream) and reads the sum behind the
    # collective on the same stream - ONE wait per pass; the single-device entry points (what one process per GPU runs, tokenmonster_amd/dist.py)
    # still hand the 80 bytes to the host and take the entry state back.  Both are projected.
    out["projected_scaling"] = {"allreduce_extra_us_%d" % us: round(t_full[0] / (t_multi[0] + us / 1e3), 2) for us in (0, 50, 100, 250)}
    out["projected_scaling_rank_processes"] = {"allreduce_extra_us_%d" % us: round(t_full[0] / (max(t_rank, t_multi[0]) + us / 1e3), 2) for us in (0, 50, 100, 250)}
    out["projected_scaling_%d" % a.ranks] = out["projected_scaling"]["allreduce_extra_us_50"]
    out["note"] = ("projected_scaling: t_full / (tm_score_multi on a one-member handle with a real one-rank ncclAllGather + ncclAllReduce: the library's own driver, "
                   "what a Go host runs); projected_scaling_rank_processes: the slower of that and (begin + chain + finish + read on the single-device entry points); "
                   "what one GPU cannot measure is what seven xGMI peers add to the two collectives: priced at 0 / 50 / 100 / 250 us")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
