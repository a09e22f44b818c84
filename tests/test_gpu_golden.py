"""-m gpu: every committed fixture through the HIP path (C ABI).

  * tests/golden/*.json(.gz)      ids / missing / count produced by the reference's own C++ runtime (make_golden.py,
                                  make_gpt2_golden.py) -> tm_tokenize_batch, tm_count_batch, tm_tokenize_batch_serialized
  * tests/golden/capcode_js.json.gz  the reference's own JavaScript capcode (make_capcode_golden.js) -> the device
                                  normalizer tm_batch_normalize and the device decode path tm_decode_batch
"""
import base64
import os

import numpy as np
import pytest

import tokenmonster_amd as tm
from conftest import golden_token_cases, load_golden
from test_capcode_golden import byte_ids, byte_vocab, cases as capcode_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", golden_token_cases(), ids=lambda p: os.path.basename(p).split(".")[0])
def test_golden_fixture_through_hip(path):
    g = load_golden(path)
    v = tm.Vocab(base64.b64decode(g["vocab_b64"]))
    docs = [base64.b64decode(d) for d in g["docs_b64"]]
    text, offs = tm.pack_documents(docs)
    ids, toff, missing = v.tokenize_packed(text, offs)
    counts, cmiss = v.count_packed(text, offs)
    for k in range(len(docs)):
        got = ids[int(toff[k]):int(toff[k + 1])].tolist()
        assert got == g["ids"][k], "doc %d" % k
        assert int(missing[k]) == g["missing"][k] and int(cmiss[k]) == g["missing"][k]
        assert int(counts[k]) == g["count"][k]
    # serialized, automatic width (go/tokenmonster.go:990-996): 2 bytes up to 65536 ids
    blob, boff, bmiss, enc = v.tokenize_serialized_packed(text, offs, 0)
    assert enc == (2 if v.n_ids() <= 65536 else 3)
    flat = np.array([x for doc in g["ids"] for x in doc], dtype=np.uint32)
    exp = np.zeros((flat.size, enc), dtype=np.uint8)
    for b in range(enc):
        exp[:, b] = (flat >> (8 * b)) & 0xFF
    assert blob.tobytes() == exp.tobytes() and int(boff[-1]) == flat.size * enc


def test_device_normalizer_equals_reference_js_capcode():
    cs = capcode_cases()
    v = tm.Vocab(byte_vocab(capcode=2, norm_flag=1))
    text, offs = tm.pack_documents([c[0] for c in cs])
    got, goff, nfb = v.normalize_packed_device(text, offs)
    bad = [k for k, c in enumerate(cs) if got[int(goff[k]):int(goff[k + 1])].tobytes() != c[2]]
    assert not bad, "%d mismatches, first: %r" % (len(bad), [cs[k][0] for k in bad[:5]])
    assert nfb < len(cs)      # the ASCII / general-punctuation documents stay on the device


def test_device_decode_equals_reference_js_capcode_decoder():
    cs = capcode_cases()
    img = byte_vocab(capcode=2, norm_flag=1)
    v = tm.Vocab(img)
    ids = byte_ids(img)
    toks = [np.array([ids[b] for b in c[2]], dtype=np.uint32) for c in cs]
    toff = np.zeros(len(cs) + 1, dtype=np.uint64)
    toff[1:] = np.cumsum([t.size for t in toks])
    flat = np.concatenate(toks) if toks else np.zeros(0, np.uint32)
    out, ooff = v.decode_packed(flat, toff, raw=False)
    bad = [k for k, c in enumerate(cs) if out[int(ooff[k]):int(ooff[k + 1])].tobytes() != c[3]]
    assert not bad, "%d mismatches, first: %r" % (len(bad), [cs[k][0] for k in bad[:5]])
    # and the whole round trip on the device path: raw -> normalize -> ids -> decode == NFD(raw)
    text, offs = tm.pack_documents([c[0] for c in cs])
    ntext, noff, _ = v.normalize_packed_device(text, offs)
    ids2, toff2, miss = v.tokenize_packed(ntext, noff)
    assert int(miss.sum()) == 0
    out2, ooff2 = v.decode_packed(ids2, toff2, raw=False)
    for k, c in enumerate(cs):
        assert out2[int(ooff2[k]):int(ooff2[k + 1])].tobytes() == c[1], c[0]
