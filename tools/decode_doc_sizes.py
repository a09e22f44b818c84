#!/usr/bin/env python3
"""tools/decode_doc_sizes.py [MiB] — development aid: the stages of the device-resident decode (tm_batch_decode_timed) on the same text cut into documents of
different sizes (every k-th document boundary of the synthetic corpus kept): what of k_dec_capcode's time is per document and what per byte."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import _native as N, synth
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = "englishcode-32000-consistent"
kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[cfg]
v = tm.Vocab(synth.config_vocab(cfg))
raw, roffs0 = synth.synth_corpus(kind, mb << 20, seed=0x434F5250 + 2)
def cut_at_spaces(size):
    """boundaries every ~size bytes, each moved back to the next space (so that no character is cut and documents begin on a word)"""
    pos = np.arange(size, raw.size - size, size, dtype=np.int64)
    sp = np.flatnonzero(raw == 32)
    pos = sp[np.minimum(np.searchsorted(sp, pos), sp.size - 1)]
    return np.unique(np.concatenate([[0], pos, [raw.size]])).astype(np.uint64)
for keep in (1, 4, 16, 64, 1024, -512, -4096, -32768):
    roffs = np.ascontiguousarray(np.concatenate([roffs0[:-1:keep], roffs0[-1:]])) if keep > 0 else np.ascontiguousarray(cut_at_spaces(-keep))
    nd = roffs.size - 1
    b = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, int(raw.size) + int(raw.size) // 4 + (1 << 20), nd, C.byref(b)))
    N.check(N.lib.tm_batch_upload_raw(b, N.ptr(raw), N.ptr(roffs), nd))
    N.check(N.lib.tm_batch_normalize(b, None)); N.check(N.lib.tm_batch_run(b, None))
    nbytes, hostd = C.c_uint64(), C.c_uint32()
    ms = (C.c_float * 3)()
    acc = np.zeros(3)
    for i in range(4):
        N.check(N.lib.tm_batch_decode_timed(b, 0, None, C.byref(nbytes), C.byref(hostd), ms))
        if i: acc += np.array(list(ms))
    acc /= 3
    print("every %4d-th boundary: %7d documents, mean %8.0f bytes, longest %8d: tile lengths + scan %.3f  gather %.3f  capcode %.3f ms per %d MiB (host docs %d)" % (
        keep, nd, raw.size / nd, int(np.diff(roffs.astype(np.int64)).max()), acc[0], acc[1], acc[2], mb, hostd.value), flush=True)
    N.lib.tm_batch_free(b)
