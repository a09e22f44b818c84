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


def test_streaming_decoder_equals_reference_decoder():
    """tm_decoder_* against the reference's own Decoder (tokenmonster.cpp:1509-1721): ids fed in random small pieces (one at a time
    included), multi-byte characters split across tokens (every UTF-8 byte is a token of its own in these vocabularies), capcode
    state carried across calls, serialized form, flush; and against the JS CapcodeDecoder golden when everything is fed at once."""
    from oracle_bind import Reference, ReferenceDecoder, have_ref
    from tokenmonster_amd import synth
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    img = synth.synth_vocab(synth.ENGLISHCODE, 3000, capcode=2, norm_flag=1, level=3, seed=0x44454344)
    v, ref = tm.Vocab(img), Reference(img)
    # The reference holds back `incompleteUTF8Bytes` bytes, which is the number of bytes still MISSING from the last character, not the
    # number present (go/tokenmonster.go:183-185 == tokenmonster.cpp:105-107): with a 3- or 4-byte character cut after its first byte
    # that is more than the buffer may hold and Go panics (slice bounds) / the C++ port reads out of bounds.  tm_decoder keeps the
    # reference's arithmetic wherever it is defined and holds back the whole buffer where it is not, so the comparison with the
    # reference runs on text whose non-ASCII characters are two bytes long; longer ones are checked against the one-shot decode.
    docs = ["Hello WORLD it's a naïve café, ÜBER straße! Ça va? señor".encode(), b"plain ascii text, Mixed CASE Words and HTTPServer2Go", b""]
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 60_000, seed=3)
    docs += [x for x in (raw[int(roffs[d]):int(roffs[d + 1])].tobytes() for d in range(min(24, roffs.size - 1))) if max(x, default=0) < 0xE0]
    long_chars = ["Hello WORLD — it’s a naïve café, 中文 日本語 😀 done.".encode(), "’".encode() * 5 + b"A" + "😀😀".encode()]
    rng = np.random.default_rng(12)
    for doc in docs + long_chars:
        with_ref = doc not in long_chars
        ids = v.tokenize(doc)
        whole = v.decode(ids)
        for mode in ("ids", "ser2"):
            d_ours, d_ref = v.decoder(), ReferenceDecoder(ref)
            got = b""
            k = 0
            while k < ids.size:
                step = int(rng.choice([1, 1, 1, 2, 3, 7, 40]))
                part = ids[k:k + step]
                k += step
                ser = part.astype("<u2").tobytes()
                a = d_ours.decode(part) if mode == "ids" else d_ours.decode_serialized(ser, 2)
                if with_ref:
                    b = d_ref.decode(part) if mode == "ids" else d_ref.decode_serialized(ser, 2)
                    assert a == b, (doc[:40], k)
                got += a
            fa = d_ours.flush()
            if with_ref:
                assert fa == d_ref.flush()
            assert got + fa == whole            # and the pieces add up to the one-shot decode
    # the JS CapcodeDecoder golden through the streaming decoder, fed at once and byte by byte
    cs = capcode_cases()
    img2 = byte_vocab(capcode=2, norm_flag=1)
    v2 = tm.Vocab(img2)
    bid = byte_ids(img2)
    for c in cs[::7]:
        toks = np.array([bid[b] for b in c[2]], dtype=np.uint32)
        d1 = v2.decoder()
        assert d1.decode(toks) + d1.flush() == c[3]
        if max(c[2], default=0) < 0xE0:      # (a longer character cut in two is handed on in pieces, as the reference does: see above)
            d2 = v2.decoder()
            assert b"".join(d2.decode(toks[i:i + 1]) for i in range(toks.size)) + d2.flush() == c[3]


@pytest.mark.parametrize("key", ["gpt2", "gpt2-capcode2-nfd"])
def test_real_text_fixture_through_hip(key):
    """tests/golden/realtext.json.gz (make_realtext_golden.py): REAL prose and code — the reference tree's own READMEs and Go / JS / C++ /
    Python sources, one document per file and per 4 KiB slice — with the gpt2.json vocabulary as it stands (capcode 0) and as a capcode-2 /
    NFD vocabulary; ids / missing / count by the reference runtime.  Through tm_tokenize_batch and tm_count_batch on host-normalized text,
    through the RAW device pass (device normalizer + walk in one batch), through the chunked host-to-host pipeline, and back through
    tm_decode_batch."""
    from conftest import GOLDEN_DIR, realtext_fixture, realtext_ids
    from oracle_bind import Reference, have_ref
    g, docs = realtext_fixture()
    if docs is None:
        pytest.skip("the documents' text (tests/golden/_realtext_docs.bin.gz, git-ignored) is not here and /root/reference is absent")
    img = base64.b64decode(load_golden(os.path.join(GOLDEN_DIR, "gpt2_vocab.json.gz"))["vocab_b64"] if key == "gpt2" else g["vocab_b_b64"])
    fx = g["vocabs"][key]
    exp = [realtext_ids(g, key, k) for k in range(len(docs))]
    v = tm.Vocab(img)
    # (1) host-normalized text -> tm_tokenize_batch / tm_count_batch
    norm = [v.normalize(d) for d in docs]
    assert [len(x) for x in norm] == fx["normalized_bytes"]
    text, offs = tm.pack_documents(norm)
    ids, toff, missing = v.tokenize_packed(text, offs)
    counts, _ = v.count_packed(text, offs)
    for k in range(len(docs)):
        got = ids[int(toff[k]):int(toff[k + 1])]
        assert got.size == exp[k].size and (got == exp[k]).all(), g["names"][k]
        assert int(missing[k]) == fx["missing"][k] and int(counts[k]) == fx["count"][k], g["names"][k]
    # (2) RAW text -> device normalizer + walk in one batch (tm_batch_upload_raw / tm_batch_normalize / tm_batch_run)
    for k, got in enumerate(v.tokenize(docs)):
        assert got.size == exp[k].size and (got == exp[k]).all(), g["names"][k]
    # (3) RAW text -> the chunked pipeline, four bytes per id, chunks far smaller than the corpus
    rtext, roffs = tm.pack_documents(docs)
    blob, boff, pmiss, enc, st = v.tokenize_pipeline(rtext, roffs, raw=True, encoding_length=4, chunk_bytes=96 * 1024, lanes=3)
    assert enc == 4 and st["chunks"] > 4 and (pmiss == np.array(fx["missing"], dtype=np.uint32)).all()
    assert blob.tobytes() == np.concatenate(exp).astype("<u4").tobytes()
    # (4) and back: tm_decode_batch == the reference runtime's decode of the same ids (capital letters, accents and all)
    out, ooff = v.decode_packed(ids, toff, raw=False)
    if have_ref():
        ref = Reference(img)
        for k in range(0, len(docs), 3):
            assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == ref.decode(exp[k]), g["names"][k]
    if key == "gpt2":
        for k in range(len(docs)):
            if fx["missing"][k] == 0:
                assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == docs[k], g["names"][k]        # capcode 0, no normalization: the text itself
