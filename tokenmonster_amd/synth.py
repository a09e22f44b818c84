"""Synthetic vocabularies / corpora of the BASELINE.json shapes (include/tm_testsupport.h, libtm_testsupport.so: test and
benchmark support, not the product) and thin wrappers of the vocabulary builder / host normalizer (include/tm_build.h)."""
import ctypes as C
import os

import numpy as np

from . import _native as N

ENGLISH, ENGLISHCODE, CODE = N.KIND_ENGLISH, N.KIND_ENGLISHCODE, N.KIND_CODE

# BASELINE.json configs as (kind, vocab size, capcode, normalization flag, mode, vocab seed)
CONFIGS = {
    "english-24000-consistent": (ENGLISH, 24000, 2, 1, 3, 0x544D0001),
    "englishcode-32000-consistent": (ENGLISHCODE, 32000, 2, 1, 3, 0x544D0002),
    "englishcode-100256-clean": (ENGLISHCODE, 100256, 2, 1, 1, 0x544D0003),
    "code-4096-balanced-nocapcode": (CODE, 4096, 0, 1, 2, 0x544D0004),
    "candidates-65536": (ENGLISHCODE, 65536, 2, 1, 5, 0x544D0005),
}
CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_vocab_cache")


def build_vocab(tokens, capcode=0, charset=1, norm_flag=0, level=5, with_unk=False, special=None):
    """tokens: iterable of bytes -> .vocab image (bytes).  go/tokenmonster.go:3423-3793 rules."""
    tokens = [bytes(t) for t in tokens]
    blob = np.frombuffer(b"".join(tokens), dtype=np.uint8) if tokens else np.zeros(0, np.uint8)
    off = np.zeros(len(tokens) + 1, dtype=np.uint32)
    np.cumsum([len(t) for t in tokens], out=off[1:])
    sp = None
    if special is not None:
        sp = np.ascontiguousarray(np.asarray(special, dtype=np.uint8))
    out = C.c_void_p()
    n = C.c_size_t()
    N.check(N.lib.tm_build_vocab(N.ptr(blob), N.ptr(off), len(tokens), N.ptr(sp), capcode, charset, norm_flag, level,
                                 1 if with_unk else 0, C.byref(out), C.byref(n)))
    return N.take(out, n.value)


def synth_vocab(kind, vocab_size, capcode=2, norm_flag=1, level=3, seed=1, with_unk=False):
    out = C.c_void_p()
    n = C.c_size_t()
    N.check(N.support_lib().tm_synth_vocab(kind, vocab_size, capcode, norm_flag, level, seed, 1 if with_unk else 0, C.byref(out),
                                 C.byref(n)))
    return N.take(out, n.value)


def config_vocab(name, cache=True):
    """.vocab image for one of the BASELINE.json shapes; cached on disk (deterministic in the seed)."""
    kind, size, capcode, norm_flag, level, seed = CONFIGS[name]
    path = os.path.join(CACHE_DIR, name + ".vocab")
    if cache and os.path.exists(path):
        with open(path, "rb") as f:
            return f.read()
    img = synth_vocab(kind, size, capcode, norm_flag, level, seed)
    if cache:
        os.makedirs(CACHE_DIR, exist_ok=True)
        with open(path + ".tmp%d" % os.getpid(), "wb") as f:
            f.write(img)
        os.replace(path + ".tmp%d" % os.getpid(), path)
    return img


def synth_corpus(kind, nbytes, seed=1, median_doc=2048, max_docs=None):
    """raw (un-normalized) synthetic documents -> (text u8[n], offsets u64[ndocs+1])"""
    if max_docs is None:
        max_docs = int(nbytes // 64 + 16)
    text = np.empty(int(nbytes) + 70000, dtype=np.uint8)
    offsets = np.empty(max_docs + 1, dtype=np.uint64)
    nd = C.c_uint32()
    nb = C.c_uint64()
    N.check(N.support_lib().tm_synth_corpus(kind, seed, int(nbytes), median_doc, N.ptr(text), N.ptr(offsets), max_docs, C.byref(nd),
                                  C.byref(nb)))
    return text[: nb.value], offsets[: nd.value + 1].copy()


def normalize(data, capcode, norm_flag):
    a = N.as_u8(data)
    out = C.c_void_p()
    n = C.c_size_t()
    N.check(N.lib.tm_normalize(N.ptr(a), a.size, capcode, norm_flag, C.byref(out), C.byref(n)))
    return N.take(out, n.value)


def normalize_batch(text, offsets, capcode, norm_flag, threads=0):
    text = N.as_u8(text)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
    nd = offsets.size - 1
    out_off = np.empty(nd + 1, dtype=np.uint64)
    out = C.c_void_p()
    N.check(N.lib.tm_normalize_batch(N.ptr(text), N.ptr(offsets), nd, capcode, norm_flag, threads, C.byref(out), N.ptr(out_off)))
    try:
        n = int(out_off[nd])
        res = np.frombuffer(C.string_at(out.value, n), dtype=np.uint8) if n else np.zeros(0, np.uint8)
    finally:
        N.lib.tm_free(out)
    return res, out_off
