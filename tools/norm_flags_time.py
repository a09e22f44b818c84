#!/usr/bin/env python3
"""tools/norm_flags_time.py [MiB] — development aid: tm_batch_normalize on the bench corpus for a few normalization flag values (the englishcode-32000
vocabulary shape with its flag byte replaced): time per call, documents left to the host normalizer, and the bytes against the host normalizer on a sample."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import _native as N, synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS["englishcode-32000-consistent"]
img0 = synth.config_vocab("englishcode-32000-consistent")
raw, roffs = synth.synth_corpus(kind, mb << 20, seed=0x434F5250 + 2)
nd = roffs.size - 1
for flag, name in ((1, "nfd (the vocabulary's own)"), (2 | 8 | 16 | 32 | 128, "lowercase collapse trim quotemarks unixlines (training/README.md's example)"),
                   (1 | 4 | 64, "nfd accents leadingspace"), (255, "all eight")):
    img = bytes(img0[:2]) + bytes([flag]) + bytes(img0[3:])
    v = tm.Vocab(img)
    b = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, int(raw.size) + int(raw.size) // 4 + (1 << 20), nd, C.byref(b)))
    N.check(N.lib.tm_batch_upload_raw(b, N.ptr(raw), N.ptr(roffs), nd))
    for _ in range(2):
        N.check(N.lib.tm_batch_normalize(b, None))
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); N.check(N.lib.tm_batch_normalize(b, None)); ts.append((time.perf_counter() - t0) * 1e3)
    nb = int(N.lib.tm_batch_normalized_bytes(b))
    fb = int(N.lib.tm_batch_host_fallback_docs(b))
    text = np.empty(max(nb, 1), dtype=np.uint8); offs = np.zeros(nd + 1, dtype=np.uint64)
    N.check(N.lib.tm_batch_download_text(b, N.ptr(text), nb, N.ptr(offs)))
    k = min(nd, 3000)
    exp, eoff = synth.normalize_batch(raw[: int(roffs[k])], roffs[: k + 1], capcode, flag)
    ok = (offs[: k + 1] == eoff).all() and (text[: exp.size] == exp).all()
    print("flag %3d (%s): %.2f ms per %d MiB (median of 5) = %.2f ms per GiB, %d of %d documents to the host, first %d documents %s the host normalizer" % (
        flag, name, sorted(ts)[2], mb, sorted(ts)[2] * 1024 / mb, fb, nd, k, "equal" if ok else "DIFFER FROM"), flush=True)
    N.lib.tm_batch_free(b)
