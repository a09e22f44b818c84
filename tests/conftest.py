import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """compile the product library and the CPU checkers once per session"""
    import __graft_entry__ as g
    g.build()


def _gpu_available():
    try:
        from tokenmonster_amd import _native as N
        return N.lib.tm_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip: the product has no CPU fallback.
    pass


# ---------------------------------------------------------------------------------------------
# fixture vocabularies and texts shared by CPU and GPU tests
# ---------------------------------------------------------------------------------------------
def unit_vocab_image():
    """byte-for-byte the vocabulary tokenmonster-cpp/tests/unit.cpp:50-85 writes"""
    NONE = b"\xff\xff\xff"
    out = bytearray([0, 0, 0, 5, 0, 0, 0, 0])
    out += NONE + (4).to_bytes(3, "little") + (4).to_bytes(3, "little") + (4).to_bytes(3, "little") + NONE + bytes([2])
    for tok, i in ((b" ", 0), (b"a", 1), (b"b", 2), (b"ab", 3)):
        out += bytes([len(tok)]) + tok + bytes([0, 0]) + NONE + NONE + i.to_bytes(3, "little") + np.float32(1.0).tobytes()
    out += bytes(256) + (0).to_bytes(3, "little")
    return bytes(out)


def fuzz_vocab_tokens(rng, capcode, n_tokens, alphabet=None, singles=True):
    """random small vocabulary over a tiny alphabet: dense enough that alternatives, forward-delete
    branches, ties and missing bytes all occur within a few kilobytes of text"""
    letters = b"abcde"
    others = b".,1_\n" + (b"DCW" if capcode == 2 else b"")
    toks = set()
    if singles:
        for ch in letters + others + b" ":
            if rng.random() < 0.9:
                toks.add(bytes([ch]))
        if capcode == 2:
            toks.add(b"D")
    while len(toks) < n_tokens:
        kind = rng.random()
        L = int(rng.integers(2, 9))
        if kind < 0.45:   # " word" style
            w = bytes(rng.choice(list(letters), size=L - 1).tolist())
            t = b" " + w
            if rng.random() < 0.3:
                t += b" " + bytes(rng.choice(list(letters), size=int(rng.integers(1, 4))).tolist())
        elif kind < 0.75:  # subword
            t = bytes(rng.choice(list(letters), size=L).tolist())
        elif kind < 0.85 and capcode == 2:
            t = bytes([int(rng.choice(list(b"DCW")))]) + b" " + bytes(rng.choice(list(letters), size=max(1, L - 2)).tolist())
        else:
            t = bytes(rng.choice(list(letters + others + b" "), size=L).tolist())
        toks.add(t[:40])
    return sorted(toks)


def fuzz_text(rng, capcode, n):
    letters = b"abcde"
    out = bytearray()
    while len(out) < n:
        r = rng.random()
        if r < 0.55:
            out += b" " + bytes(rng.choice(list(letters), size=int(rng.integers(1, 9))).tolist())
        elif r < 0.65:
            out += bytes(rng.choice(list(letters), size=int(rng.integers(1, 12))).tolist())
        elif r < 0.75 and capcode == 2:
            out += bytes([int(rng.choice(list(b"DCW")))]) + b" " + bytes(rng.choice(list(letters), size=int(rng.integers(1, 6))).tolist())
        elif r < 0.9:
            out += bytes([int(rng.choice(list(b".,1_\n ")))])
        else:
            out += bytes([int(rng.choice(list(b"xyz\x00\xff")))])   # bytes the vocabulary may not have
    return bytes(out[:n])
