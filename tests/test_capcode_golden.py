"""Capcode level 2 and the NFD pre-step, pinned to the reference's OWN statement of capcode.

tests/golden/capcode_js.json.gz was produced by tests/golden/make_capcode_golden.js, which evaluates
/root/reference/javascript/tokenmonster.js:872-1065 (capcode_encode, CapcodeDecoder) where it lies, under node 12, on
5 261 strings (caps / digits / apostrophes / U+2019 / combining marks / accents / CJK / Cyrillic / Greek / emoji, plus every
hand-written string of the device-normalizer test).  Replayed here against
  * the product's host normalizer  tm_normalize            (tokenmonster_amd/csrc/tm_normalize.cpp)
  * the checker's capcode          oracle/capcode/capcode.hpp through the reference runtime's Vocab::normalize / ::decode
and, under -m gpu, against the device normalizer (tm_batch_normalize) and the device decode path (tm_decode_batch).
"""
import base64
import gzip
import json
import os

import numpy as np
import pytest

from oracle_bind import Reference, have_ref
from tokenmonster_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "capcode_js.json.gz")


def cases():
    g = json.load(gzip.open(GOLDEN))
    assert g["n"] >= 5000 and len(g["cases"]) == g["n"]
    return [tuple(base64.b64decode(c[k]) for k in ("in", "nfd", "enc", "dec")) for c in g["cases"]]


def byte_vocab(capcode=2, norm_flag=1):
    """every single byte is a token: tokenizing X and decoding it again exercises nothing but normalize / capcode"""
    return synth.build_vocab([bytes([c]) for c in range(256)], capcode=capcode, charset=1, norm_flag=norm_flag)


def byte_ids(img):
    """byte value -> token id of the one-byte record"""
    n_info = int.from_bytes(img[17:20], "little")
    pos, ids = 24, {}
    for _ in range(n_info):
        kl = img[pos]
        key = img[pos + 1: pos + 1 + kl]
        p = pos + 1 + kl
        if kl == 1:
            ids[key[0]] = int.from_bytes(img[p + 8:p + 11], "little")
        pos = p + 15
    return ids


def test_host_normalizer_equals_reference_js_capcode():
    bad = []
    for raw, nfd, enc, _ in cases():
        if synth.normalize(nfd, 2, 0) != enc:        # capcode alone, on what JS's NFD produced
            bad.append(("capcode", raw))
        if synth.normalize(raw, 2, 1) != enc:        # NFD (ICU) + capcode, the pre-step of Tokenize (go/tokenmonster.go:242-253)
            bad.append(("nfd+capcode", raw))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:5])


@pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built")
def test_checker_capcode_equals_reference_js_capcode():
    """oracle/capcode/capcode.hpp is what the reference runtime in oracle/_ref is linked against: pin it the same way"""
    img = byte_vocab()
    ref = Reference(img)
    ids = byte_ids(img)
    bad = []
    for raw, nfd, enc, dec in cases():
        if ref.normalize(raw) != enc:
            bad.append(("encode", raw))
        toks = np.array([ids[b] for b in enc], dtype=np.uint32)
        if ref.decode(toks) != dec:
            bad.append(("decode", raw))
    assert not bad, "%d mismatches, first: %r" % (len(bad), bad[:5])


def test_js_decoder_inverts_js_encoder_on_nfd_text():
    """property of the fixture itself (and the reason decode can be checked at scale by round trips)"""
    n_ok = sum(1 for _, nfd, _, dec in cases() if dec == nfd)
    assert n_ok == len(cases())
