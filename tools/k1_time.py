#!/usr/bin/env python3
"""tools/k1_time.py [--lib path/to/libtokenmonster_hip.so ...] [--mbytes 128] — K1 time without torch: the hot path (host-normalized text)
of the englishcode-32000 shape through tm_batch_run_timed (HIP events on the launch stream) for the product library and for every
variant library given, in ONE process each (a fresh interpreter per library: the binding loads one library).  Prints the per-kernel
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


def one(lib, mbytes):
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
    name = "englishcode-32000-consistent"
    kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[name]
    v = tm.Vocab(synth.config_vocab(name))
    raw, roffs = synth.synth_corpus(kind, mbytes << 20, seed=0x434F5250 + 2)
    text, offs = synth.normalize_batch(raw, roffs, capcode, norm_flag)
    nd = offs.size - 1
    batch = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, int(text.size) + (1 << 20), nd, C.byref(batch)))
    N.check(N.lib.tm_batch_upload(batch, N.ptr(text), N.ptr(offs), nd))
    ms = (C.c_float * N.TM_NUM_KERNELS)()
    acc = np.zeros(N.TM_NUM_KERNELS)
    N.check(N.lib.tm_batch_run(batch, None))
    reps = 6
    for _ in range(reps):
        N.check(N.lib.tm_batch_run_timed(batch, None, ms))
        acc += np.array(list(ms))
    acc /= reps
    ntok, nmiss = C.c_uint64(), C.c_uint64()
    N.check(N.lib.tm_batch_totals(batch, C.byref(ntok), C.byref(nmiss)))
    ids = np.empty(max(int(ntok.value), 1), dtype=np.uint32)
    toff = np.empty(nd + 1, dtype=np.uint64)
    N.check(N.lib.tm_batch_download(batch, N.ptr(ids), int(ntok.value), N.ptr(toff), None))
    h = hashlib.md5(ids[: int(ntok.value)].tobytes() + toff.tobytes()).hexdigest()
    names = [N.lib.tm_kernel_name(k).decode() for k in range(N.TM_NUM_KERNELS)]
    print("%-40s %d MiB: %s  tokens %d  ids md5 %s  (%.1f s)" % (os.path.relpath(lib, ROOT) if lib else "product library", mbytes,
          " ".join("%s %.3f" % (n, x) for n, x in zip(names, acc)), int(ntok.value), h[:12], time.time() - t0), flush=True)
    N.lib.tm_batch_free(batch)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", action="append", default=[])
    ap.add_argument("--mbytes", type=int, default=128)
    ap.add_argument("--one", default=None)
    a = ap.parse_args()
    if a.one is not None:
        return one(a.one or None, a.mbytes)
    for lib in [""] + a.lib:
        subprocess.call([sys.executable, os.path.abspath(__file__), "--one", os.path.abspath(lib) if lib else "", "--mbytes", str(a.mbytes)])


if __name__ == "__main__":
    main()
