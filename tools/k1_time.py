#!/usr/bin/env python3
"""tools/k1_time.py [--lib path/to/libtokenmonster_hip.so ...] [--mbytes 128] [--config shape] [--e2e] [--score] — kernel times without torch:
the hot path (host-normalized text) of a vocabulary shape through tm_batch_run_timed (HIP events on the launch stream), optionally the
end-to-end step on raw text (host clock) or the scoring pass, for the product library and for every variant library given, in ONE process each (a fresh interpreter per library: the binding loads one library).  Prints the per-kernel
milliseconds and a checksum of the ids, which must be the same for every library.  Development aid for the A/B of build-time kernel
variants when GPU time is short (bench.py's torch import alone takes a minute on a fresh box)."""
import argparse
import ctypes as C
import hashlib
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(lib, mbytes, config="englishcode-32000-consistent", e2e=False, score=False, reps=6):
    if lib:
        real = C.CDLL

        class Patched(real):
            def __init__(self, name, *a, **k):
                if isinstance(name, str) and os.path.basename(name) == "libtokenmonster_hip.so" and "testsupport" not in name:
                    name = lib
                super().__init__(name, *a, **k)
        C.CDLL = Patched
    sys.path.insert(0, ROOT)
    import numpy as np
    import tokenmonster_amd as tm
    from tokenmonster_amd import _native as N, synth
    N.check(N.lib.tm_set_device(0))
    t0 = time.time()
    name = config
    kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[name]
    v = tm.Vocab(synth.config_vocab(name))
    raw, roffs = synth.synth_corpus(kind, mbytes << 20, seed=0x434F5250 + (5 if score else 2))
    text, offs = synth.normalize_batch(raw, roffs, capcode, norm_flag)
    nd = offs.size - 1
    label = os.path.relpath(lib, ROOT) if lib else "product library"
    tune_mib = float(os.environ.get("TM_K1_TUNE_MIB", "0") or 0)
    if tune_mib > 0:
        # tm_vocab_tune on OTHER text of the same kind (another seed), not on what is timed
        sraw, sroffs = synth.synth_corpus(kind, int(tune_mib * (1 << 20)), seed=0x434F5250 + 77)
        stext, _ = synth.normalize_batch(sraw, sroffs, capcode, norm_flag)
        tt = time.time()
        v.tune(stext)
        label += " tuned on %.3g MiB (%.2f s)" % (tune_mib, time.time() - tt)
    if score:
        # the trainvocab scoring pass over ONE strip (tm_score: match kernel, resolve, histogram walk), wall clock around the call
        ds = C.c_void_p()
        data = np.ascontiguousarray(text)
        N.check(N.lib.tm_dataset_upload(N.ptr(data), int(data.size), C.byref(ds)))
        got = np.zeros(v.n_ids(), dtype=np.uint32)
        tit = C.c_uint64()
        ms8 = np.zeros(32, dtype=np.uint8)
        N.check(N.lib.tm_score(v.handle, ds, None, None, 0, N.ptr(got), C.byref(tit), N.ptr(ms8)))
        t1 = time.perf_counter()
        for _ in range(reps):
            N.check(N.lib.tm_score(v.handle, ds, None, None, 0, N.ptr(got), C.byref(tit), N.ptr(ms8)))
        dt = (time.perf_counter() - t1) / reps * 1e3
        h = hashlib.md5(got.tobytes() + ms8.tobytes()).hexdigest()
        print("%-40s %s %d MiB scoring pass: %.3f ms (host clock, incl. the histogram read)  tokens_in_text %d  histogram md5 %s  (%.1f s)" % (
            label, name, mbytes, dt, tit.value, h[:12], time.time() - t0), flush=True)
        N.lib.tm_dataset_free(ds)
        return
    batch = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, int(text.size) + (1 << 20), nd, C.byref(batch)))
    ntok, nmiss = C.c_uint64(), C.c_uint64()
    step_ms = None
    if e2e:
        # the end-to-end step of bench.py: RAW text resident, normalize on the device + tokenize; host clock, tm_batch_totals synchronizes
        N.check(N.lib.tm_batch_upload_raw(batch, N.ptr(raw), N.ptr(roffs), nd))
        for _ in range(2):
            N.check(N.lib.tm_batch_normalize(batch, None))
            N.check(N.lib.tm_batch_run(batch, None))
        N.check(N.lib.tm_batch_totals(batch, C.byref(ntok), C.byref(nmiss)))
        t1 = time.perf_counter()
        for _ in range(reps):
            N.check(N.lib.tm_batch_normalize(batch, None))
            N.check(N.lib.tm_batch_run(batch, None))
        N.check(N.lib.tm_batch_totals(batch, C.byref(ntok), C.byref(nmiss)))
        step_ms = (time.perf_counter() - t1) / reps * 1e3
        # the normalizer alone (tm_batch_normalize synchronizes): median of 2 x reps calls
        tn = []
        for _ in range(2 * reps):
            t1 = time.perf_counter(); N.check(N.lib.tm_batch_normalize(batch, None)); tn.append((time.perf_counter() - t1) * 1e3)
        norm_ms = sorted(tn)[len(tn) // 2]
    else:
        N.check(N.lib.tm_batch_upload(batch, N.ptr(text), N.ptr(offs), nd))
    ms = (C.c_float * N.TM_NUM_KERNELS)()
    acc = np.zeros(N.TM_NUM_KERNELS)
    N.check(N.lib.tm_batch_run(batch, None))
    for _ in range(reps):
        N.check(N.lib.tm_batch_run_timed(batch, None, ms))
        acc += np.array(list(ms))
    acc /= reps
    N.check(N.lib.tm_batch_totals(batch, C.byref(ntok), C.byref(nmiss)))
    ids = np.empty(max(int(ntok.value), 1), dtype=np.uint32)
    toff = np.empty(nd + 1, dtype=np.uint64)
    N.check(N.lib.tm_batch_download(batch, N.ptr(ids), int(ntok.value), N.ptr(toff), None))
    h = hashlib.md5(ids[: int(ntok.value)].tobytes() + toff.tobytes()).hexdigest()
    names = [N.lib.tm_kernel_name(k).decode() for k in range(N.TM_NUM_KERNELS)]
    print("%-40s %s %d MiB: %s%s  tokens %d  ids md5 %s  (%.1f s)" % (label, name, mbytes, " ".join("%s %.3f" % (n, x) for n, x in zip(names, acc)),
          "" if step_ms is None else "  | end-to-end step %.3f ms = %.2f GB/s raw, normalize alone %.3f ms" % (step_ms, raw.size / step_ms / 1e6, norm_ms), int(ntok.value), h[:12], time.time() - t0), flush=True)
    N.lib.tm_batch_free(batch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--mbytes", type=int, default=128)
    ap.add_argument("--config", default="englishcode-32000-consistent", help="vocabulary shape (tokenmonster_amd/synth.py CONFIGS)")
    ap.add_argument("--e2e", action="store_true", help="also time the end-to-end step (device normalizer + tokenizer) on RAW resident text")
    ap.add_argument("--score", action="store_true", help="time the trainvocab scoring pass (tm_score, one strip) instead")
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--one", default=None)
    a = ap.parse_args()
    if a.one is not None:
        return one(a.one or None, a.mbytes, a.config, a.e2e, a.score, a.reps)
    for lib in [""] + a.lib:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one", os.path.abspath(lib) if lib else "", "--mbytes", str(a.mbytes), "--config", a.config,
                         "--reps", str(a.reps)] + (["--e2e"] if a.e2e else []) + (["--score"] if a.score else []))


if __name__ == "__main__":
    main()
