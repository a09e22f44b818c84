"""-m gpu: the HIP path (through the C ABI) against the CPU oracle, bit-exact token ids."""
import os

import numpy as np
import pytest

import tokenmonster_amd as tm
from tokenmonster_amd import synth
from conftest import EMULATED, example_env, fuzz_text, fuzz_vocab_tokens, unit_vocab_image
from oracle_bind import Oracle, Reference, have_ref, oracle_stats

pytestmark = pytest.mark.gpu


def check_docs(vocab, orc, docs, what=""):
    text, offs = tm.pack_documents(docs)
    ids, toff, missing = vocab.tokenize_packed(text, offs)
    assert toff[0] == 0 and toff[-1] == ids.size
    for d, doc in enumerate(docs):
        exp, miss = orc.tokenize(doc)
        got = ids[int(toff[d]):int(toff[d + 1])]
        assert got.size == exp.size and (got == exp).all(), "%s doc %d (len %d): ids differ\nexp %s\ngot %s" % (
            what, d, len(doc), exp[:40], got[:40])
        assert int(missing[d]) == miss, "%s doc %d: missing %d != %d" % (what, d, int(missing[d]), miss)
    return ids, toff, missing


def test_unit_golden_vector():
    # tokenmonster-cpp/tests/unit.cpp:87-112
    v = tm.Vocab(unit_vocab_image())
    ids, missing = v.tokenize_normalized(b"ab a z")
    assert ids.tolist() == [3, 0, 1, 0] and missing == 1
    counts, miss = v.count_packed(*tm.pack_documents([b"ab a z"]))
    assert int(counts[0]) == 4 and int(miss[0]) == 1
    b, boff, _, enc = v.tokenize_serialized_packed(*tm.pack_documents([b"ab"]), encoding_length=2)
    assert enc == 2 and b.tolist() == [3, 0]


def test_fuzz_capcode1_vocab():
    # capcode level 1: delete marker 0x7F only (go/tokenmonster.go:273, :3436, :3480); tokens "\x7F "+word exist
    rng = np.random.default_rng(4242)
    toks = [t.replace(b"D", b"\x7f").replace(b"C", b"c").replace(b"W", b"w") for t in fuzz_vocab_tokens(rng, 2, 160)] + [b"\x7f"]
    img = synth.build_vocab(toks, capcode=1, charset=1)
    v, orc = tm.Vocab(img), Oracle(img)
    assert v.capcode() == 1 and v.delete_token_id() is not None
    oracle_stats(reset=True)
    docs = [fuzz_text(rng, 2, int(n)).replace(b"D", b"\x7f").replace(b"C", b"c").replace(b"W", b"w") for n in rng.integers(0, 2500, size=80)]
    check_docs(v, orc, docs, "capcode 1")
    st = oracle_stats()
    assert st["s1"] > 0 and st["s2"] > 0 and st["s1b"] + st["s2b"] + st["s3b"] > 0, st
    if have_ref():
        ref = Reference(img)
        for d in docs[::7]:
            assert v.tokenize_normalized(d)[0].tolist() == ref.tokenize_normalized(d)[0].tolist()


@pytest.mark.parametrize("capcode", [0, 2])
@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_fuzz_micro_vocab(capcode, seed):
    rng = np.random.default_rng(1000 * capcode + seed)
    toks = fuzz_vocab_tokens(rng, capcode, 60 + 40 * seed)
    img = synth.build_vocab(toks, capcode=capcode, charset=1, with_unk=(seed % 2 == 0))
    v = tm.Vocab(img)
    orc = Oracle(img)
    oracle_stats(reset=True)
    docs = [fuzz_text(rng, capcode, int(n)) for n in rng.integers(0, 3000, size=120)]
    # empty / tiny documents, lengths around multiples of the segment size (whatever it is), one long document
    docs += [b"", b"a", b" "] + [fuzz_text(rng, capcode, n) for n in (255, 256, 257, 319, 320, 321, 383, 384, 385, 511, 512, 513, 639, 640, 641, 767, 768, 769,
                                                                      1024, 1025, 70000)]
    if seed == 1:
        docs.append(fuzz_text(rng, capcode, 400_000))     # > LONG_SEGS segments: hierarchical resolve path
    check_docs(v, orc, docs, "fuzz capcode=%d seed=%d" % (capcode, seed))
    st = oracle_stats()
    # the fuzz must actually have exercised the alternatives; with capcode 2 also the forward-delete branches
    assert st["s1"] > 0 and st["s2"] > 0 and st["s3"] > 0 and st["not_found"] > 0, st
    if capcode == 2:
        assert st["s1b"] + st["s2b"] + st["s3b"] > 0, st


def test_dense_forward_delete_path():
    """K1 hands the T(p,1) words to K4 as a short per-segment side list; segments with more forward-delete states than the
    list holds use a dense array instead.  Debug bit 6 forces that path for every segment."""
    from tokenmonster_amd import _native as N
    rng = np.random.default_rng(77)
    toks = fuzz_vocab_tokens(rng, 2, 140)
    img = synth.build_vocab(toks, capcode=2, charset=1, with_unk=True)
    v = tm.Vocab(img)
    orc = Oracle(img)
    docs = [fuzz_text(rng, 2, int(n)) for n in rng.integers(0, 3000, size=60)] + [fuzz_text(rng, 2, 200_000)]
    oracle_stats(reset=True)
    old = N.lib.tm_debug_flags(64)
    try:
        check_docs(v, orc, docs, "dense (p,1) path")
    finally:
        N.lib.tm_debug_flags(old)
    st = oracle_stats()
    assert st["s1b"] + st["s2b"] + st["s3b"] > 0, st
    check_docs(v, orc, docs[:10], "side-list path")


@pytest.mark.parametrize("flags", [64, 1024, 1024 | 64, 4096, 4096 | 64, 32768, 32768 | 1024, 32768 | 64])
def test_fallback_paths_behind_test_hooks(flags):
    """Rarely taken fallback paths of the product, forced through the test hooks of tm_debug_flags: bit 10 makes the K4 tile walk
    store every id directly (the path it takes when a text averages more than one id per byte), bit 6 uses the dense T(p,1) array
    for every segment (the path of a segment with more forward-delete states than its side list holds), bit 12 hangs every
    document of more than 8 segments under a group tree of fan-out 4 (the resolve of multi-megabyte documents and of dataset strips,
    five levels deep on the 200 000-byte document here), bit 15 walks the two-plane rows in the id-staging form that vocabularies of more than
    65 536 ids use (k_emit_tiles) instead of the position-staging one (k_emit_list).  All must give the oracle's ids / histogram.  Bits outside the hooks are ignored by the default build (they need -DTM_DEVEL)."""
    from tokenmonster_amd import _native as N
    rng = np.random.default_rng(78)
    toks = fuzz_vocab_tokens(rng, 2, 140)
    img = synth.build_vocab(toks, capcode=2, charset=1, with_unk=True)
    v = tm.Vocab(img)
    orc = Oracle(img)
    docs = [fuzz_text(rng, 2, int(n)) for n in rng.integers(0, 3000, size=60)] + [b"", b"a", fuzz_text(rng, 2, 200_000)]
    data = fuzz_text(rng, 2, 120_000)
    exp_s, exp_t, exp_m = orc.score(data)
    old = N.lib.tm_debug_flags(flags)
    try:
        assert N.lib.tm_debug_flags(-1) == flags
        check_docs(v, orc, docs, "test hook flags %d" % flags)
        got_s, got_t, got_m = _score(v, data)
    finally:
        N.lib.tm_debug_flags(old)
    assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
    assert N.lib.tm_debug_flags(1 | 4 | 16 | 128 | 512) == 0 and N.lib.tm_debug_flags(0) == 0    # profiling switches: not in this build


@pytest.mark.parametrize("name", ["english-24000-consistent", "englishcode-32000-consistent", "englishcode-100256-clean",
                                  "code-4096-balanced-nocapcode"])       # BASELINE.json configs[0..3] (shapes; synthetic)
def test_synthetic_config(name):
    kind, size, capcode, norm_flag, level, seed = synth.CONFIGS[name]
    img = synth.config_vocab(name)
    v = tm.Vocab(img)
    orc = Oracle(img)
    raw, offs = synth.synth_corpus(kind, 400_000, seed=0x434F5250 + 2)
    text, noff = synth.normalize_batch(raw, offs, capcode, norm_flag)
    docs = [text[int(noff[d]):int(noff[d + 1])].tobytes() for d in range(noff.size - 1)]
    oracle_stats(reset=True)
    ids, toff, _ = check_docs(v, orc, docs, name)
    st = oracle_stats()
    assert st["s1"] > 0 and st["s2"] > 0
    # Count(): b-branches count once (quirk Q2)
    counts, _ = v.count_packed(text, noff)
    for d in range(0, len(docs), 17):
        assert int(counts[d]) == orc.count(docs[d])[0]
    # reference's own runtime, when its prebuilt library travelled with the snapshot
    if have_ref():
        ref = Reference(img)
        for d in range(0, len(docs), 7):
            exp, _ = ref.tokenize_normalized(docs[d])
            got = ids[int(toff[d]):int(toff[d + 1])]
            assert got.size == exp.size and (got == exp).all()


def _score(vocab, data, strips=None):
    """tm_score on a freshly uploaded dataset -> (scores, tokens_in_text, missing_set)"""
    import ctypes as C
    from tokenmonster_amd import _native as N
    d = np.frombuffer(bytes(data), dtype=np.uint8)
    ds = C.c_void_p()
    N.check(N.lib.tm_dataset_upload(N.ptr(d), d.size, C.byref(ds)))
    try:
        scores = np.zeros(vocab.n_ids(), dtype=np.uint32)
        tit = C.c_uint64()
        ms = np.zeros(32, dtype=np.uint8)
        if strips is None:
            N.check(N.lib.tm_score(vocab.handle, ds, None, None, 0, N.ptr(scores), C.byref(tit), N.ptr(ms)))
        else:
            so = np.array([a for a, _ in strips], dtype=np.uint64)
            sl = np.array([l for _, l in strips], dtype=np.uint64)
            N.check(N.lib.tm_score(vocab.handle, ds, N.ptr(so), N.ptr(sl), len(strips), N.ptr(scores), C.byref(tit), N.ptr(ms)))
        return scores, tit.value, ms
    finally:
        N.lib.tm_dataset_free(ds)


@pytest.mark.parametrize("capcode", [0, 2])
def test_score_histogram_micro(capcode):
    # training/trainvocab.go:925-1176: scores[id] += bytes covered, scores[deleteToken]++, tokensInText, missing bytes
    rng = np.random.default_rng(300 + capcode)
    toks = fuzz_vocab_tokens(rng, capcode, 150)
    img = synth.build_vocab(toks, capcode=capcode, charset=1)
    v, orc = tm.Vocab(img), Oracle(img)
    data = fuzz_text(rng, capcode, 90_000)
    exp_s, exp_t, exp_m = orc.score(data)
    got_s, got_t, got_m = _score(v, data)
    assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
    # strips (the pre-"midway" mode, trainvocab.go:1668-1695): each strip is walked independently
    strips = [(0, 10_000), (20_000, 4096), (50_000, 513), (70_000, 19_999), (89_999, 1), (5, 0)]
    exp_s = np.zeros(orc.n_ids(), dtype=np.uint32)
    exp_t = 0
    exp_m = np.zeros(32, dtype=np.uint8)
    for a, l in strips:
        s, t, m = orc.score(data[a:a + l])
        exp_s += s
        exp_t += t
        exp_m |= m
    got_s, got_t, got_m = _score(v, data, strips)
    assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()


def test_score_histogram_candidate_vocab():
    img = synth.synth_vocab(synth.ENGLISHCODE, 8000, capcode=2, norm_flag=1, level=5, seed=0x544D0005)
    v, orc = tm.Vocab(img), Oracle(img)
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 600_000, seed=44)
    text, _ = synth.normalize_batch(raw, offs, 2, 1)
    exp_s, exp_t, exp_m = orc.score(text)
    got_s, got_t, got_m = _score(v, text)
    assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
    assert int(got_s.sum()) >= text.size        # every byte covered once (+1 per forward delete)


def test_score_histogram_full_candidate_shape():
    """BASELINE.json configs[4] at its full vocabulary shape: 65 536 candidate ids (~113 000 index records), one 8 MiB strip
    (training/trainvocab.go:909-922 post-midway mode) against the oracle's scoring mode."""
    img = synth.config_vocab("candidates-65536")
    v, orc = tm.Vocab(img), Oracle(img)
    assert v.n_ids() == 65536
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 8 << 20, seed=0x434F5250 + 5)
    text, _ = synth.normalize_batch(raw, offs, 2, 1)
    assert text.size >= 8 << 20
    exp_s, exp_t, exp_m = orc.score(text)
    got_s, got_t, got_m = _score(v, text)
    assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
    # strips that cut the dataset at arbitrary bytes are walked independently (pre-midway mode, :1668-1695)
    strips = [(0, 1 << 20), (3 << 20, (2 << 20) + 77), ((6 << 20) + 13, 1 << 19)]
    es = np.zeros(orc.n_ids(), dtype=np.uint32)
    et, em = 0, np.zeros(32, dtype=np.uint8)
    for a, l in strips:
        s_, t_, m_ = orc.score(text[a:a + l])
        es += s_
        et += t_
        em |= m_
    got_s, got_t, got_m = _score(v, text, strips)
    assert (got_s == es).all() and got_t == et and (got_m == em).all()


@pytest.mark.parametrize("shape", ["micro-capcode2", "micro-capcode0-unk", "englishcode-8000"])
def test_score_histogram_equals_the_replay_of_the_reference_ids(shape):
    """the reference-anchored check of the scoring accumulation (training/trainvocab.go:1105-1174): the reference runtime has no scoring
    mode, but the histogram follows from the ids IT emits for the same text once every id is given back the bytes it covered
    (tests/replay.py: which token - from the reference; how many bytes - from the text and the key set).  tm_score must equal that."""
    from oracle_bind import Reference, have_ref
    from replay import histogram_from_ids
    if not have_ref():
        pytest.skip("oracle/_ref not built")
    if shape == "englishcode-8000":
        img = synth.synth_vocab(synth.ENGLISHCODE, 8000, capcode=2, norm_flag=1, level=5, seed=0x544D0005)
        raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 300_000, seed=45)
        data, _ = synth.normalize_batch(raw, offs, 2, 1)
    else:
        capcode = 2 if shape == "micro-capcode2" else 0
        rng = np.random.default_rng(4100 + capcode)
        img = synth.build_vocab(fuzz_vocab_tokens(rng, capcode, 150), capcode=capcode, charset=1, with_unk=(capcode == 0))
        data = np.frombuffer(fuzz_text(rng, capcode, 120_000), dtype=np.uint8)
    ids, missing = Reference(img).tokenize_normalized(data)
    exp_s, exp_t, exp_m = histogram_from_ids(img, data, ids, missing)
    got_s, got_t, got_m = _score(tm.Vocab(img), data)
    assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()


@pytest.mark.parametrize("shape", ["micro", "candidates"])
def test_score_ranges_of_one_walk(shape):
    """tm_score_begin / tm_score_finish: a dataset cut into byte ranges, each uploaded with a halo of the following text and scored as a
    piece of ONE whole-buffer walk (training/trainvocab.go:909-922), the ranges chained through their 80-entry exit maps exactly as
    N ranks do (tokenmonster_amd/dist.py).  The summed histogram must equal the oracle's walk over the whole buffer."""
    import ctypes as C
    from tokenmonster_amd import _native as N
    from tokenmonster_amd import dist as tmdist
    if shape == "micro":
        rng = np.random.default_rng(91)
        img = synth.build_vocab(fuzz_vocab_tokens(rng, 2, 150), capcode=2, charset=1)
        data = np.frombuffer(fuzz_text(rng, 2, 300_000), dtype=np.uint8)
        cuts = [0, 64, 200, 70_001, 70_100, 150_000, 299_900, 300_000]
    else:
        img = synth.config_vocab("candidates-65536")
        raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 3 << 20, seed=0x434F5250 + 5)
        data, _ = synth.normalize_batch(raw, offs, 2, 1)
        n = int(data.size)
        cuts = [0, n // 3 + 1, 2 * n // 3 + 2, n]          # a long range each: the group maps (k_group_compose) carry the exit map
    v, orc = tm.Vocab(img), Oracle(img)
    exp_s, exp_t, exp_m = orc.score(data)
    n_ids = v.n_ids()
    all_exits, hists = [], []
    handles = []
    for a, b in zip(cuts, cuts[1:]):
        own = data[a:min(b + tmdist.HALO, data.size)]       # the rank's bytes + halo: all it ever sees
        ds = C.c_void_p()
        N.check(N.lib.tm_dataset_upload(N.ptr(np.ascontiguousarray(own)), own.size, C.byref(ds)))
        handles.append(ds)
        eng = tmdist.HipRange(v, ds, b - a, continues=b < data.size, text_ends_in_halo=data.size - b < tmdist.HALO)
        ex = eng.begin()
        # the device's exit map agrees with the oracle's range walk on every entry state the walk can be in
        for e in (0, 2, 7, 21):
            if ex[e] != tmdist.UNREACHABLE and e % 2 == 0:
                assert int(ex[e]) == orc.score_range(data, a, b, e)[3], (a, b, e)
        all_exits.append(ex)
        entry = tmdist.resolve_entry(all_exits, len(all_exits) - 1)
        eng.finish(entry)
        s_ = np.zeros(n_ids, dtype=np.uint32)
        t_ = C.c_uint64()
        m_ = np.zeros(32, dtype=np.uint8)
        N.check(N.lib.tm_score_read(v.handle, ds, N.ptr(s_), C.byref(t_), N.ptr(m_)))
        hists.append((s_, t_.value, m_))
    got_s = sum(h[0].astype(np.uint64) for h in hists)
    got_t = sum(h[1] for h in hists)
    got_m = np.bitwise_or.reduce(np.stack([h[2] for h in hists]))
    for ds in handles:
        N.lib.tm_dataset_free(ds)
    assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
    assert tmdist.resolve_entry(all_exits, len(all_exits)) == 0     # the walk ends at the end of the text
    assert any(tmdist.resolve_entry(all_exits, r) != 0 for r in range(1, len(all_exits)))    # some cut fell inside a token


def test_serialized_auto_width_large_vocab():
    """go/tokenmonster.go:990-996: encoding_length 0 picks 3 bytes per id once the vocabulary has more than 65 536 ids
    (englishcode-100256 shape); 2- and 4-byte requests are honoured as given (:1545, :2089; 2 bytes truncates, as Go's does)."""
    img = synth.config_vocab("englishcode-100256-clean")
    v = tm.Vocab(img)
    assert v.n_ids() == 100256
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 600_000, seed=0x434F5250 + 3)
    text, noff = synth.normalize_batch(raw, offs, 2, 1)
    ids, toff, miss = v.tokenize_packed(text, noff)
    assert int(ids.max()) > 65535            # the high ids are really in use
    for enc_req, enc_exp in ((0, 3), (3, 3), (4, 4), (2, 2)):
        blob, boff, bmiss, enc = v.tokenize_serialized_packed(text, noff, enc_req)
        assert enc == enc_exp and (bmiss == miss).all()
        assert (boff == toff * np.uint64(enc)).all()
        exp = np.zeros((ids.size, enc), dtype=np.uint8)
        for b in range(min(enc, 3)):
            exp[:, b] = (ids >> (8 * b)) & 0xFF
        assert blob.tobytes() == exp.tobytes()


def test_device_normalizer_matches_host_and_reference():
    # go/tokenmonster.go:242-253 pre-step on the GPU vs the host normalizer (and the reference runtime's normalize)
    img = synth.synth_vocab(synth.ENGLISHCODE, 1200, capcode=2, norm_flag=1, level=3, seed=7)
    v = tm.Vocab(img)
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 1_500_000, seed=21)
    rng = np.random.default_rng(8)
    alphabet = list(b"aBcDeFGhij XYZ'1234567890.,-_()\n\t") + ["’", "“", "”", "—", "…"]
    extra = []
    for _ in range(3000):
        s = "".join(c if isinstance(c, str) else chr(c) for c in rng.choice(np.array(alphabet, dtype=object), size=int(rng.integers(0, 90))))
        extra.append(s.encode())
    extra += [b"", b"A", b"a", b"AB", b"Ab", b"aB", b"ABc", b"ABC", b" ABC d", b"HTTPServer2Go x", b"X's Y'S it's 'a' I'M", b"12AB34cd", b"A1B2c",
              "X’s Y’S it’s".encode(), b"A" * 200 + b"b", b"A" * 200, b"a" + b"B" * 130 + b" " + b"C" * 70 + b"d", "café Über".encode(),
              b"A" * 3000 + b"b", b"xY" * 2500, b"Q" * 5000,          # expand beyond a piece's 2 KiB slab: exact two-pass path
              b"\xff\xfe bad bytes", "  en quad".encode(),
              # capitals without a lower-case form: as a later capital of a run that ends in a lower-case letter the reference gives them no
              # marker (its pass over the run looks at the letters after lower-casing) - their documents take the host path
              "Bϒa".encode(), "B'ϔ0Ba".encode(), "BϔBa".encode(), "ϒa".encode(), "Bϒ".encode(), ("B" * 70 + "'" * 70 + "ϔ0" + "B" * 130 + "a" * 30).encode()]
    # documents whose capital / digit / apostrophe runs straddle the 64-byte chunks and 1 KiB pieces of the device pass
    for n in (60, 63, 64, 65, 1020, 1023, 1024, 1025, 2047, 2048, 2049):
        extra += [b"x" * n + b" Abc " + b"DEF'S 12a", b"x" * (n - 2) + b"AB" + b"C" * 70 + b"d", b"x" * (n - 1) + b" " + b"Q" * 130,
                  b"y" * (n - 3) + "I’M".encode() + b" OK", b"1" * n + b"a", b"z" * n + b" " + b"A" * 1100 + b"b"]
    etext, eoffs = tm.pack_documents(extra)
    from tokenmonster_amd import _native as N
    for flags in (0, 256):       # debug bit 8: the per-lane kernel (k_norm_emit<2>) instead of k_norm_emit2
        old = N.lib.tm_debug_flags(flags)
        try:
            for text_in, offs_in in ((raw, offs), (etext, eoffs)):
                got, goff, nfb = v.normalize_packed_device(text_in, offs_in)
                exp, eoff = synth.normalize_batch(text_in, offs_in, 2, 1)
                assert (goff == eoff).all(), "flags %d" % flags
                assert got.size == exp.size and (got == exp).all(), "flags %d" % flags
                assert nfb < (offs_in.size - 1) // 5 + 8          # most documents are handled on the device
                if flags == 0:
                    # the same with a small grid: every wavefront of k_norm_emit2 takes several pieces, one after the other
                    os.environ["TM_NORM_WG_PER_CU"] = "1"
                    try:
                        got1, goff1, _ = v.normalize_packed_device(text_in, offs_in)
                    finally:
                        del os.environ["TM_NORM_WG_PER_CU"]
                    assert (goff1 == eoff).all() and got1.size == exp.size and (got1 == exp).all()
        finally:
            N.lib.tm_debug_flags(old)
    # lower-case-everything flag with capcode 2 (capitals are classified as letters)
    img3 = synth.synth_vocab(synth.ENGLISHCODE, 1200, capcode=2, norm_flag=3, level=3, seed=7)
    got3, goff3, _ = tm.Vocab(img3).normalize_packed_device(etext, eoffs)
    exp3, eoff3 = synth.normalize_batch(etext, eoffs, 2, 3)
    assert (goff3 == eoff3).all() and got3.size == exp3.size and (got3 == exp3).all()
    if have_ref():
        ref = Reference(img)
        for d in range(0, len(extra), 37):
            a, b_ = int(goff[d]), int(goff[d + 1])
            assert got[a:b_].tobytes() == ref.normalize(extra[d])
    # capcode 0 + lowercase flag
    img0 = synth.build_vocab([bytes([c]) for c in range(256)], capcode=0, charset=1, norm_flag=3)
    v0 = tm.Vocab(img0)
    got, goff, _ = v0.normalize_packed_device(etext, eoffs)
    exp, eoff = synth.normalize_batch(etext, eoffs, 0, 3)
    assert (goff == eoff).all() and (got == exp).all()


def test_match_kernel_stages_the_text_from_the_normalizer_slabs():
    """tm_batch_normalize leaves the normalized text in its per-piece slabs (no compaction pass) and k_match_branch stages every segment from
    there (k_seg_src: a segment begins anywhere in a piece and may run into the next one, words straddle the seam at every alignment):
    ids == the ids of the host-normalized text == the ids with the text packed first (test hook 11, the path a batch with host-normalized
    documents takes); a batch with such documents; the packed text can still be had afterwards"""
    from tokenmonster_amd import _native as N
    img = synth.synth_vocab(synth.ENGLISHCODE, 1500, capcode=2, norm_flag=1, level=3, seed=11)
    v = tm.Vocab(img)
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 600_000, seed=33)
    docs = [raw[int(offs[d]):int(offs[d + 1])].tobytes() for d in range(offs.size - 1)]
    rng = np.random.default_rng(5)
    for n in (1, 255, 256, 257, 352, 1023, 1024, 1025, 1279, 1280, 2047, 2048, 2049, 4095, 4096, 4097, 70_000):
        for caps in (0, 1, 2, 3, 5):            # capitals move the seams between the pieces byte by byte (a marker in front of each)
            body = bytearray(b"the quick brown fox jumps over the lazy dog " * (n // 44 + 1))[:n]
            for i in rng.integers(0, max(n, 1), size=caps):
                body[int(i)] = ord("Q")
            docs.append(bytes(body))
    docs += [b"", b"A", "déjà vu Élan".encode() * 300, b"x" * 1024 + b"Y" * 3 + b"z" * 1021]
    clean = list(docs)
    host_ids, _ = v.tokenize_normalized([v.normalize(d) for d in clean])
    for flags in (0, 2048):
        old = N.lib.tm_debug_flags(flags)
        try:
            got = v.tokenize(clean)
        finally:
            N.lib.tm_debug_flags(old)
        bad = [(d, len(clean[d]), g.size, h.size) for d, (g, h) in enumerate(zip(got, host_ids)) if g.size != h.size or (g != h).any()]
        assert not bad, (flags, len(bad), bad[:12])
    # a batch in which some documents go to the host normalizer (the device part is packed, the host part placed behind it)
    mixed = clean[:200] + ["\u1e9e gro\u00dfe \U0001d400 x".encode(), "\uff21\uff22 fullwidth".encode()] + clean[200:400]
    got = v.tokenize(mixed)
    exp, _ = v.tokenize_normalized([v.normalize(d) for d in mixed])
    for d, (g, h) in enumerate(zip(got, exp)):
        assert g.size == h.size and (g == h).all(), d
    # ... and the normalized text itself, packed on request after the usual path has left it in the slabs
    text_in, offs_in = tm.pack_documents(clean)
    gtext, goff, nfb = v.normalize_packed_device(text_in, offs_in)
    etext, eoff = synth.normalize_batch(text_in, offs_in, 2, 1)
    assert nfb == 0 and (goff == eoff).all() and (gtext == etext).all()


def test_lossy_normalizer_flags_run_on_the_device():
    """a vocabulary with accents / quotemarks / collapse / trim / leadingspace / unixlines (training/README.md:110-123; the reference's own
    example is -norm "lowercase collapse trim quotemarks unixlines"): since round 6 a filter pass in front of the normalizer pass does
    these on the device too (tm_norm.hip: k_pf_*), quirks of the reference's in-place loops included - NO document of ASCII + Latin text
    goes to the host normalizer, for every one of the 256 flag values, and the bytes are the host normalizer's (= the reference runtime's:
    tests/test_builder_normalizer.py)."""
    docs = [b"  Hello   World \r\n", "\u201cQuoted\u201d  caf\u00e9  \u2018x\u2019 ".encode(), b"", b"a", b" \t ", b"No leading blank  here", b"\r\n", b"ab",
            "a  \u2018b\u2019 c  \u201cd\u201d e".encode(), "x \u2019y  z \u2019w".encode(), "\u2019".encode(), "\u00c9a  \u00f1 \u00fc na\u00efve".encode(), b"  ",
            b"one  two \r\n three\r\n\r\n  four  ", ("Title Case  And  CAPS \u2018q\u2019 \r\n" * 40).encode()]
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 60_000, seed=5)
    docs += [raw[int(roffs[d]):int(roffs[d + 1])].tobytes() for d in range(roffs.size - 1)]
    # documents of more than one 16 KiB span of the filter pass (a wavefront per span there, k_pf_long): the single drops, the quotes they
    # protect, the blanks at both ends and the characters `accents` changes on either side of a span's end
    docs += [("one  two \u2018q\u2019 \r\n caf\u00e9 " * 3000).encode(), b" \t " * 9000 + "x \u201cy\u201d".encode() * 2000 + b"  z" + b" " * 40000,
             ("a \u2019b" * 2730 + "  " + "\u2018c\u2019 \u00e9\u00f1" * 5000 + "  d \u201d").encode(), b"Q" + b"abc \r\n" * 2340 + "\u2019".encode() * 9000]
    text, offs = tm.pack_documents(docs)
    toks = [bytes([c]) for c in range(256)]
    for flag in range(256):
        for capcode in ((2, 0) if flag % 16 == 9 or flag in (255, 254, 4, 6) else (2,)):
            v = tm.Vocab(synth.build_vocab(toks, capcode=capcode, charset=1, norm_flag=flag))
            got, goff, nfb = v.normalize_packed_device(text, offs)
            # (without capcode the normalizer pass keeps lengths: a character that decomposes is the host's; trim + leadingspace cut the last
            # non-blank BYTE of a text without leading blanks, tokenmonster.cpp:274-277 - half a character in eight of these documents: malformed, the host's)
            assert nfb == 0 or capcode == 0 or ((flag & 96) == 96 and nfb <= 8), (flag, capcode, nfb)
            exp, eoff = synth.normalize_batch(text, offs, capcode, flag)
            assert (goff == eoff).all() and got.tobytes() == exp.tobytes(), (flag, capcode)
    # ... and the ids of a real vocabulary with the reference's example flags (lowercase collapse trim quotemarks unixlines)
    flag = 2 | 8 | 16 | 32 | 128
    img = synth.synth_vocab(synth.ENGLISHCODE, 1500, capcode=2, norm_flag=1, level=3, seed=3)
    img = bytes(img[:2]) + bytes([flag]) + bytes(img[3:])
    v, orc = tm.Vocab(img), Oracle(img)
    ids = v.tokenize(docs)
    for d, doc in enumerate(docs):
        assert ids[d].tolist() == orc.tokenize(synth.normalize(doc, 2, flag))[0].tolist()
    # what the filter pass leaves to the host: a three-byte mark under `accents` (the document is normalized from its ORIGINAL bytes)
    v = tm.Vocab(synth.build_vocab(toks, capcode=2, charset=1, norm_flag=4 | 16 | 32 | 64))
    odd = ["  a\u20dd  b ".encode(), b"  plain  text ", "e\u0301\ufe0f  x".encode()]
    t2, o2 = tm.pack_documents(odd)
    got, goff, nfb = v.normalize_packed_device(t2, o2)
    assert nfb == 2
    for d, doc in enumerate(odd):
        assert got[int(goff[d]):int(goff[d + 1])].tobytes() == synth.normalize(doc, 2, 4 | 16 | 32 | 64)


def _utf16(bs):
    return b"".join(bytes([c, 0]) for c in bs)


@pytest.mark.parametrize("seed", [1, 2])
def test_fuzz_utf16_vocab(seed):
    # charset 2: the forward-delete probe uses the two-byte prefix ' ' 0x00 (lilbufOffset 2, go :1031-1034, quirk Q8)
    rng = np.random.default_rng(500 + seed)
    toks8 = fuzz_vocab_tokens(rng, 2, 150)
    toks = sorted(set(_utf16(t) for t in toks8 if len(t) <= 20) | {b"D", b" ", b"a"})     # plus a few odd-length keys
    img = synth.build_vocab(toks, capcode=2, charset=2)
    v, orc = tm.Vocab(img), Oracle(img)
    assert v.charset() == 2
    oracle_stats(reset=True)
    docs = [_utf16(fuzz_text(rng, 2, int(n))) for n in rng.integers(0, 1500, size=60)]
    docs += [_utf16(fuzz_text(rng, 2, 3000))[:-1], b"", _utf16(b" abc abc")]
    ids, toff, _ = check_docs(v, orc, docs, "utf16 seed=%d" % seed)
    st = oracle_stats()
    assert st["s1"] > 0 and st["s2"] > 0
    if have_ref():
        ref = Reference(img)
        for d in range(0, len(docs), 5):
            exp, _ = ref.tokenize_normalized(docs[d])
            got = ids[int(toff[d]):int(toff[d + 1])]
            assert got.size == exp.size and (got == exp).all()


def test_utf16_cut_character_ends_the_walk_with_an_error():
    """A degenerate UTF-16 case on which the REFERENCE does not terminate: a vocabulary with one-byte keys (half a character) and a text
    that ends in half a character.  In a forward-delete state the length of a candidate is its key length minus the two-byte prefix
    ' ' 0x00 (tokenmonster.cpp:1790-1797: length1b -= lilbuf_offset), which for such keys is not positive: the walk stops advancing,
    and the reference runtime — and the oracle, which restates it — loop forever (the same vocabulary hangs them on some texts of whole
    characters too).  The device pipeline cannot loop: a state that never leaves its segment has no exit-map entry, and the call
    returns TM_E_INPUT.  Found by differential fuzzing on the emulated device; invalid input for the reference's callers."""
    from tokenmonster_amd import _native as N
    rng = np.random.default_rng(913)
    toks8 = fuzz_vocab_tokens(rng, 2, 100)
    toks = sorted(set(_utf16(t) for t in toks8 if len(t) <= 20) | {b"D", b" ", b"a"})
    img = synth.build_vocab(toks, capcode=2, charset=2)
    docs = []
    for n in rng.integers(0, 1800, size=40):
        doc = _utf16(fuzz_text(rng, 2, int(n)))
        docs.append(doc[:-1] if n % 3 == 0 else doc)
    bad = [d for d in docs if len(d) == 2975]
    assert len(bad) == 1
    v = tm.Vocab(img)
    with pytest.raises(N.TokenMonsterHipError) as e:
        v.tokenize_packed(*tm.pack_documents([bad[0]]))
    assert e.value.code == N.TM_E_INPUT
    ids, _, _ = v.tokenize_packed(*tm.pack_documents([bad[0][:-1], bad[0] + b"\x00"]))      # whole characters either side of it: fine
    assert ids.size > 0


def test_host_api_edge_cases():
    from tokenmonster_amd import _native as N
    import ctypes as C
    v = tm.Vocab(unit_vocab_image())
    # zero documents, empty documents
    ids, toff, missing = v.tokenize_packed(np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert ids.size == 0 and toff.tolist() == [0]
    ids, toff, missing = v.tokenize_packed(*tm.pack_documents([b"", b"ab", b""]))
    assert ids.tolist() == [3] and toff.tolist() == [0, 0, 1, 1]
    # TM_E_NOSPACE reports the required capacity in tok_offsets[ndocs]
    text, offs = tm.pack_documents([b"ab a z", b"abab"])
    tok_off = np.zeros(3, dtype=np.uint64)
    out = np.zeros(1, dtype=np.uint32)
    miss = np.zeros(2, dtype=np.uint32)
    rc = N.lib.tm_tokenize_batch(v.handle, N.ptr(text), N.ptr(offs), 2, N.ptr(out), 1, N.ptr(tok_off), N.ptr(miss))
    assert rc == N.TM_E_NOSPACE and int(tok_off[2]) == 6
    # serialized: encoding lengths 2, 3, 4 (go :1545/:1817/:2089) and the invalid one (go :1012)
    for enc, exp in ((2, [3, 0]), (3, [3, 0, 0]), (4, [3, 0, 0, 0])):
        b, boff, _, used = v.tokenize_serialized_packed(*tm.pack_documents([b"ab"]), encoding_length=enc)
        assert used == enc and b.tolist() == exp
    with pytest.raises(N.TokenMonsterHipError):
        v.tokenize_serialized_packed(*tm.pack_documents([b"ab"]), encoding_length=5)
    # raw-text entry point: normalize on the device, then tokenize
    img = synth.synth_vocab(synth.ENGLISHCODE, 1500, capcode=2, norm_flag=1, level=3, seed=3)
    v2, orc2 = tm.Vocab(img), Oracle(img)
    raw_docs = [b"Hello World, this is A TEST of HTTPServer2Go!", "It’s “quoted” — naïve café".encode(), b""]
    got = v2.tokenize(raw_docs)
    for d, r in zip(got, raw_docs):
        exp, _ = orc2.tokenize(synth.normalize(r, 2, 1))
        assert d.tolist() == exp.tolist()


def test_decode_matches_reference_and_round_trips():
    # go/tokenmonster.go:445-550 Decode; tokenmonster.cpp:1404-1425 decode_raw / decode
    img = synth.synth_vocab(synth.ENGLISHCODE, 3000, capcode=2, norm_flag=1, level=3, seed=31)
    v, orc = tm.Vocab(img), Oracle(img)
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 300_000, seed=17)
    text, noff = synth.normalize_batch(raw, offs, 2, 1)
    ids, toff, _ = v.tokenize_packed(text, noff)
    # decode_raw: concatenated token bytes.  reverse[id] is the LAST record with that id (quirk Q3), so the raw bytes
    # are not always the normalized text, but they are exactly what the oracle/reference produce ...
    out_raw, ooff = v.decode_packed(ids, toff, raw=True)
    nd = noff.size - 1
    ref = Reference(img) if have_ref() else None
    for d in range(nd):
        t = ids[int(toff[d]):int(toff[d + 1])]
        got = out_raw[int(ooff[d]):int(ooff[d + 1])].tobytes()
        assert got == orc.decode_raw(t)
        if ref is not None and d % 5 == 0:
            assert got == ref.decode_raw(t)
    # ... and capcode-decoding them gives back the original text (NFD form), like the reference's decode()
    out, ooff2 = v.decode_packed(ids, toff, raw=False)
    for d in range(0, nd, 3):
        got = out[int(ooff2[d]):int(ooff2[d + 1])].tobytes()
        doc = raw[int(offs[d]):int(offs[d + 1])].tobytes()
        if ref is not None:
            import ctypes as C
            t = np.ascontiguousarray(ids[int(toff[d]):int(toff[d + 1])])
            buf = np.empty(t.size * 40 + 64, dtype=np.uint8)
            n = ref.L.tmref_decode(ref.h, t.ctypes.data, t.size, buf.ctypes.data, buf.size)
            assert got == buf[:n].tobytes()
        if all(b < 0x80 for b in doc):          # ASCII documents: NFD is the identity, so decode(tokenize(x)) == x
            assert got == doc
    # out-of-range ids are skipped (tokenmonster.cpp:1407), empty documents, TM_E_NOSPACE handled by the wrapper
    weird = np.array([0xFFFFFF, 5, v.n_ids() + 7, 6], dtype=np.uint32)
    assert v.decode_packed(weird, np.array([0, 0, 4], dtype=np.uint64), raw=True)[0].tobytes() == orc.decode_raw(np.array([5, 6], dtype=np.uint32))


def test_decode_tiles_blocks_and_length_classes():
    """round 6's decode: the gather works on tiles of 2 048 ids (a document that begins exactly on a tile, an id count that is a multiple of the tile, ids
    that do not exist, a tile of 40-byte keys - more than one LDS window), the capcode decoder takes the text through LDS in blocks of 1 KiB aligned
    in the buffer (characters of two, three and four bytes across every block and chunk boundary, markers as the last byte of a block) and dispatches
    documents by length class (>= 16 KB, >= 4 KB, shorter; empty ones).  Device == host decoder, byte for byte."""
    import unicodedata
    rng = np.random.default_rng(6006)
    toks = [bytes([c]) for c in range(256)] + [b"D", b"C", b"W", b"the", b" the", b"ing", b"D ", b"x" * 40, b"y" * 39 + b"C", "é".encode(), "ж".encode(), "世".encode(), "😀".encode()]
    v = tm.Vocab(synth.build_vocab(sorted(set(toks)), capcode=2, charset=1, norm_flag=1))
    words = ["hello", "World", "HTTP", "it's", "Émile", "ŒUVRE", "Жук", "ЖУК", "世界", "😀", "x", "I", "2nd", "don’t", "NAÏVE", "naïve", " ", "\n", "ß", "ǅ"]
    def text(n):
        out, tot = [], 0
        while tot < n:
            w = words[int(rng.integers(len(words)))] + (" " if rng.random() < 0.8 else "")
            out.append(w); tot += len(w.encode())
        return "".join(out)
    docs = [text(int(n)) for n in (0, 1, 63, 64, 65, 1023, 1024, 1025, 2047, 2049, 4095, 4097, 16383, 16385, 40_000, 0, 5, 300, 70_000 if not EMULATED else 20_000)]
    docs += [text(int(n)) for n in rng.integers(0, 3000, size=40)]
    raw, offs = tm.pack_documents([d.encode() for d in docs])
    ntext, noff = synth.normalize_batch(raw, offs, 2, 1)
    ids, toff, miss = v.tokenize_packed(ntext, noff)
    assert int(miss.sum()) == 0
    # (a) as tokenized; (b) with ids that do not exist sprinkled in and the id count padded to a multiple of the tile; (c) every document begins on a tile
    def check(ids_, toff_, expect):
        out, ooff = v.decode_packed(ids_, toff_)
        for k, e in enumerate(expect):
            assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == e, (k, len(e))
    expect = [unicodedata.normalize("NFD", d).encode() for d in docs]
    check(ids, toff, expect)
    bad = np.array([v.n_ids(), 0xFFFFFF, 0xFFFFFFFF], dtype=np.uint32)
    parts, nt = [], [0]
    for k in range(len(docs)):
        t = ids[int(toff[k]):int(toff[k + 1])]
        t = np.insert(t, rng.integers(0, t.size + 1, size=3), bad) if k % 3 == 0 else t
        pad = (-(nt[-1] + t.size)) % 2048 if k % 5 == 0 else 0
        t = np.concatenate([t, np.full(pad, v.n_ids() + 1, dtype=np.uint32)])
        parts.append(t); nt.append(nt[-1] + t.size)
    check(np.ascontiguousarray(np.concatenate(parts)), np.array(nt, dtype=np.uint64), expect)
    # raw decode of a tile of 40-byte keys: 2 048 x 40 bytes = five LDS windows
    long_ids = v.tokenize_normalized([b"x" * 40 * 3000])[0][0]
    assert long_ids.size == 3000
    out, ooff = v.decode_packed(long_ids, np.array([0, 3000], dtype=np.uint64), raw=True)
    assert out.tobytes() == b"x" * 120_000


@pytest.mark.parametrize("name,mbytes", [("englishcode-32000-consistent", 256), ("englishcode-100256-clean", 64), ("code-4096-balanced-nocapcode", 64),
                                         ("english-24000-consistent", 64)])
def test_full_size_properties(name, mbytes):
    """BASELINE-size shapes cannot be re-walked by the oracle in seconds; check size-independent properties instead:
    batch-split invariance, decode round trip, a random sample against the oracle, end-to-end == host-normalized path.
    Every named vocabulary shape of BASELINE.json: the bench shape at 256 MiB, the others at 64 MiB."""
    kind, size, capcode, norm_flag, level, seed = synth.CONFIGS[name]
    v, orc = tm.Vocab(synth.config_vocab(name)), Oracle(synth.config_vocab(name))
    raw, offs = synth.synth_corpus(kind, ((4 if mbytes > 64 else 1) << 20) if EMULATED else (mbytes << 20), seed=0x434F5250 + 77)   # (the emulation leg runs ~1 MiB/s)
    text, noff = synth.normalize_batch(raw, offs, capcode, norm_flag)
    nd = noff.size - 1
    ids, toff, missing = v.tokenize_packed(text, noff)
    assert int(missing.sum()) == 0 and toff[-1] == ids.size
    # (1) the same documents tokenized in two separate batches give the same ids (no cross-document / cross-segment leakage)
    h = nd // 2
    a_ids, a_off, _ = v.tokenize_packed(text[: int(noff[h])], noff[: h + 1])
    b_ids, b_off, _ = v.tokenize_packed(text[int(noff[h]):], noff[h:] - noff[h])
    assert a_ids.size + b_ids.size == ids.size and (np.concatenate([a_ids, b_ids]) == ids).all()
    # (2) raw text through the device normalizer gives the same ids as the host-normalized path
    got = v.tokenize([raw[int(offs[d]):int(offs[d + 1])].tobytes() for d in range(0, nd, 97)])
    for k, d in enumerate(range(0, nd, 97)):
        assert (got[k] == ids[int(toff[d]):int(toff[d + 1])]).all()
    # (3) decode round trip on every ASCII document (NFD is the identity there): decode(tokenize(x)) == x
    out, ooff = v.decode_packed(ids, toff, raw=False)
    n_ascii = 0
    for d in range(nd):
        doc = raw[int(offs[d]):int(offs[d + 1])]
        if doc.size and int(doc.max()) < 0x80:
            n_ascii += 1
            assert out[int(ooff[d]):int(ooff[d + 1])].tobytes() == doc.tobytes(), "document %d does not round-trip" % d
    assert n_ascii > nd // 20            # (prose carries curly quotes and accents: a tenth of the english documents are pure ASCII, most of the code ones)
    # (4) a random sample against the oracle
    rng = np.random.default_rng(5)
    for d in rng.choice(nd, size=150, replace=False):
        exp, _ = orc.tokenize(text[int(noff[d]):int(noff[d + 1])])
        assert (ids[int(toff[d]):int(toff[d + 1])] == exp).all()


def test_batch_workspace_reuse():
    """one tm_batch reused for different uploads (normalized and raw), like a long-lived server worker would"""
    import ctypes as C
    from tokenmonster_amd import _native as N
    img = synth.synth_vocab(synth.ENGLISHCODE, 2000, capcode=2, norm_flag=1, level=3, seed=12)
    v, orc = tm.Vocab(img), Oracle(img)
    b = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, 4 << 20, 4096, C.byref(b)))
    try:
        for trial, nbytes in enumerate((300_000, 20_000, 900_000, 0, 150_000)):
            raw, offs = synth.synth_corpus(synth.ENGLISHCODE, max(nbytes, 1), seed=100 + trial) if nbytes else (np.zeros(0, np.uint8), np.zeros(1, np.uint64))
            text, noff = synth.normalize_batch(raw, offs, 2, 1)
            nd = noff.size - 1
            if trial % 2 == 0:
                N.check(N.lib.tm_batch_upload(b, N.ptr(text), N.ptr(noff), nd))
            else:
                N.check(N.lib.tm_batch_upload_raw(b, N.ptr(raw), N.ptr(offs), nd))
                N.check(N.lib.tm_batch_normalize(b, None))
            N.check(N.lib.tm_batch_run(b, None))
            ntok, nmiss = C.c_uint64(), C.c_uint64()
            N.check(N.lib.tm_batch_totals(b, C.byref(ntok), C.byref(nmiss)))
            ids = np.empty(max(int(ntok.value), 1), dtype=np.uint32)
            toff = np.zeros(nd + 1, dtype=np.uint64)
            miss = np.zeros(max(nd, 1), dtype=np.uint32)
            N.check(N.lib.tm_batch_download(b, N.ptr(ids), int(ntok.value), N.ptr(toff), N.ptr(miss)))
            for d in range(nd):
                exp, m = orc.tokenize(text[int(noff[d]):int(noff[d + 1])])
                assert (ids[int(toff[d]):int(toff[d + 1])] == exp).all() and int(miss[d]) == m
        # a batch larger than the workspace is refused, not truncated
        raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 6 << 20, seed=9)
        text, noff = synth.normalize_batch(raw, offs, 2, 1)
        assert N.lib.tm_batch_upload(b, N.ptr(text), N.ptr(noff), noff.size - 1) == N.TM_E_LIMIT
    finally:
        N.lib.tm_batch_free(b)


@pytest.mark.gpu
def test_c_example_matches_python_path(tmp_path):
    """examples/tokenize_file.c (plain C on the C ABI) prints the same ids as the Python mirror."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    rng = np.random.default_rng(5)
    toks = fuzz_vocab_tokens(rng, 0, 120)
    img = synth.build_vocab(toks, capcode=0, charset=1, with_unk=True)
    lines = [fuzz_text(rng, 0, int(n)).replace(b"\n", b" ") + b"\n" for n in rng.integers(1, 900, size=40)]
    (tmp_path / "v.vocab").write_bytes(bytes(img))
    (tmp_path / "t.txt").write_bytes(b"".join(lines))
    r = subprocess.run([os.path.join(root, "examples", "tokenize_file"), str(tmp_path / "v.vocab"), str(tmp_path / "t.txt"), "--lines"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=example_env())
    assert r.returncode == 0, r.stderr.decode(errors="replace")
    got = [[int(x) for x in ln.split()] for ln in r.stdout.decode().split("\n")[:len(lines)]]
    v = tm.Vocab(img)
    for ln, g in zip(lines, got):
        assert v.tokenize_normalized(ln)[0].tolist() == g


def european_corpus(rng, nbytes):
    """synthetic French / German / Polish / Turkish running text: accented Latin letters (U+00C0..U+017F), guillemets, ordinals, the typographic
    apostrophe; now and then a word in another script (the document then needs the host normalizer)"""
    words = ("le la les un une des et ou où à été être État États-Unis l’été d'Émile garçon français naïve cœur Œuvre sœur Noël façade Âge "
             "der die das und über Über Größe Straße STRASSE Mädchen Äpfel Österreich Übung fünf weiß Fuß müssen können ÖBB "
             "Łódź żółć gęślą jaźń Świat Zażółć İstanbul ışık Ağrı çok Şimdi "
             "«bonjour» »Guten Tag« 1º 2ª 3ème n° 20 °C 5 µm ½ ¿Qué? ¡Hola! señor AÑO Ñandú").split()
    foreign = ["Москва", "東京", "😀", "αβγ"]
    docs = []
    total = 0
    while total < nbytes:
        n = int(rng.integers(3, 400))
        ws = [str(rng.choice(words)) for _ in range(n)]
        if rng.random() < 0.04:
            ws.insert(int(rng.integers(0, n)), str(rng.choice(foreign)))
        if rng.random() < 0.2:
            ws = [w.upper() if rng.random() < 0.3 and "µ" not in w else w for w in ws]       # (the capital of µ is the GREEK letter Μ: another script)
        d = (" ".join(ws) + str(rng.choice([".", "!", " ?", "…", ""]))).encode()
        docs.append(d)
        total += len(d)
    return docs


def test_european_text_stays_on_the_device():
    """two-byte UTF-8 (Latin-1 Supplement, Latin Extended-A) is normalized by the device pass itself: NFD decomposition, lower case and
    capcode of accented letters come from a table the HOST normalizer fills (tm_norm_masks.h: NmTwo), so one `é` no longer sends its
    document to ICU.  The bytes must equal the host normalizer's, and >= 90 % of the documents must have stayed on the device."""
    rng = np.random.default_rng(2024)
    docs = european_corpus(rng, 300_000 if EMULATED else 6_000_000)
    raw, offs = tm.pack_documents(docs)
    for capcode, flag in ((2, 1), (2, 3), (2, 0), (0, 2)):
        v = tm.Vocab(synth.build_vocab([bytes([c]) for c in range(256)], capcode=capcode, charset=1, norm_flag=flag))
        got, goff, nfb = v.normalize_packed_device(raw, offs)
        exp, eoff = synth.normalize_batch(raw, offs, capcode, flag)
        assert (goff == eoff).all() and got.size == exp.size
        assert (got == exp).all(), "capcode %d flag %d" % (capcode, flag)
        if flag & 1:      # (without NFD the one capital whose lower case has another length - İ, in nearly every document here - keeps its documents on the host)
            assert nfb <= len(docs) // 10, "%d of %d documents took the host path (capcode %d flag %d)" % (nfb, len(docs), capcode, flag)
        if capcode == 2 and flag == 1:
            # and through the whole path: raw text in, ids out == ids of the host-normalized text
            ids, toff, _ = v.tokenize_packed(exp, eoff)
            gotids = v.tokenize(docs[:200])
            for k in range(200):
                assert (gotids[k] == ids[int(toff[k]):int(toff[k + 1])]).all()


def multilingual_corpus(rng, nbytes):
    """synthetic Russian / Greek / Chinese / Japanese / Hebrew / Arabic running text with English mixed in: two-byte scripts with case
    and decomposing letters (й ё Й, ά ώ), combining marks, CJK ideographs and kana; a few percent of the documents carry what the device
    left to the host at one time or still (Hangul, emoji, Latin Extended Additional: the device's by now; a voicing mark by itself, the Greek letters with two marks);
    the voiced kana of Japanese (が -> か + U+3099 under NFD) are part of the running text: the device's since round 6"""
    ru = "Привет мир Москва Россия Ёжик йод объём СЪЕЗД Киев Санкт-Петербург это русский текст и ещё й ЙОД".split()
    el = "Καλημέρα κόσμε Αθήνα Ελλάδα ΑΘΗΝΑ ά έ ή ί ό ύ ώ Ώρα το και είναι ελληνικό κείμενο".split()
    zh = ["中文", "文本", "测试", "世界", "你好", "数据", "模型", "，", "。", "ABC", "GPU"]
    ja = ["こんにちは", "世界", "カタカナ", "ひらかな", "テスト", "日本", "の", "は", "、", "。", "Tokyo", "です", "ございます", "がんばって", "データ", "プログラム", "ヴァイオリン", "ぱぴぷぺぽ"]
    he = "שלום עולם זה טקסט בעברית".split()
    ar = "مرحبا بالعالم هذا نص عربي ١٢٣".split()
    en = "the quick Brown FOX jumps over 13 lazy dogs it's NASA's iPhone".split()
    host_only = ["한국어", "か\u3099", "😀", "Ḁḁ", "ΐ"]
    langs = [ru, el, zh, ja, he, ar]
    docs, total = [], 0
    while total < nbytes:
        n = int(rng.integers(3, 300))
        lang = langs[int(rng.integers(0, len(langs)))]
        ws = [str(rng.choice(lang if rng.random() < 0.8 else en)) for _ in range(n)]
        if rng.random() < 0.03:
            ws.insert(int(rng.integers(0, n)), str(rng.choice(host_only)))
        sep = "" if lang in (zh, ja) else " "
        d = (sep.join(ws) + str(rng.choice([".", "!", "", "…"]))).encode()
        docs.append(d)
        total += len(d)
    return docs


def test_multilingual_text_stays_on_the_device():
    """the device normalizer beyond Latin (go/tokenmonster.go:233-253; classes javascript/tokenmonster.js:880-898): every two-byte script
    (Greek, Cyrillic, Hebrew, Arabic ...: NFD, case and capcode from a 1 920-entry table the HOST normalizer fills) and the three-byte
    characters the normalizer leaves alone (CJK ideographs, most kana: two bits per code point, from the host's functions too).  The bytes
    must equal the host normalizer's and >= 95 % of the documents must have stayed on the device; the ids of the whole raw path equal
    the ids of the host-normalized text."""
    rng = np.random.default_rng(2025)
    docs = multilingual_corpus(rng, 200_000 if EMULATED else 6_000_000)
    raw, offs = tm.pack_documents(docs)
    for capcode, flag in ((2, 1), (2, 3), (0, 0)):
        v = tm.Vocab(synth.build_vocab([bytes([c]) for c in range(256)], capcode=capcode, charset=1, norm_flag=flag))
        got, goff, nfb = v.normalize_packed_device(raw, offs)
        exp, eoff = synth.normalize_batch(raw, offs, capcode, flag)
        assert (goff == eoff).all() and got.size == exp.size
        assert (got == exp).all(), "capcode %d flag %d" % (capcode, flag)
        assert nfb <= len(docs) // 20, "%d of %d documents took the host path (capcode %d flag %d)" % (nfb, len(docs), capcode, flag)
        if capcode == 2 and flag == 1:
            ids, toff, _ = v.tokenize_packed(exp, eoff)
            nchk = min(200, len(docs))
            gotids = v.tokenize(docs[:nchk])
            for k in range(nchk):
                assert (gotids[k] == ids[int(toff[k]):int(toff[k + 1])]).all()
            # ... and back: the capcode decoder takes the same scripts on the device (k_dec_capcode), equal to the host's streaming decoder
            from tokenmonster_amd import _native as N
            out, ooff = v.decode_packed(ids[: int(toff[nchk])], np.ascontiguousarray(toff[: nchk + 1]))
            assert N.lib.tm_decode_host_docs() <= max(1, nchk // 10), "%d of %d documents were decoded on the host" % (N.lib.tm_decode_host_docs(), nchk)
            for k in range(nchk):
                dec = v.decoder()
                assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == dec.decode(ids[int(toff[k]):int(toff[k + 1])]) + dec.flush()


def emoji_korean_corpus(rng, nbytes):
    """web-like text: English / French sentences with AT LEAST one emoji per document (four bytes of UTF-8: emoticons, pictographs, a
    skin-tone modifier, flags), Korean running text (Hangul syllables with and without a final consonant) with English mixed in, and
    plane-2 ideographs / hieroglyphs / cuneiform here and there; one document in a hundred carries what still needs the host (a Deseret
    capital, a musical symbol that decomposes, a voicing mark by itself, a mathematical letter); the variation selector U+FE0F behind an emoji and the enclosing
    keycap - combining marks of three bytes and canonical class 0 - stay on the device"""
    en = "the quick Brown FOX jumps over 13 lazy dogs it's NASA's iPhone LOL omg so café naïve Über".split()
    ko = "한국어 텍스트 대한민국 서울 값 삶 닭 없다 읽다 가 나 다 라 마 바 사 아 자 차 카 타 파 하 안녕하세요 감사합니다 GPU 토큰".split()
    emoji = ["😀", "😂", "🚀", "🌍", "👍", "👍🏽", "🎉", "🔥", "💯", "🤖", "🦄", "🇰🇷", "🀄", "❤️", "☺️", "1️⃣"]
    astral = ["𠀀", "𠮷", "𓀀", "𒀀"]
    host_only = ["𐐀", "𝅗𝅥", "か\u3099", "𝒜"]         # (mathematical letters: their blocks of 64 code points have unassigned holes, the block table says "mixed")
    docs, total = [], 0
    while total < nbytes:
        n = int(rng.integers(2, 200))
        korean = rng.random() < 0.4
        ws = [str(rng.choice(ko if korean and rng.random() < 0.8 else en)) for _ in range(n)]
        for _ in range(1 + int(rng.integers(0, 3))):
            ws.insert(int(rng.integers(0, len(ws) + 1)), str(rng.choice(emoji)))
        if rng.random() < 0.1:
            ws.insert(int(rng.integers(0, len(ws))), str(rng.choice(astral)))
        if rng.random() < 0.01:
            ws.insert(int(rng.integers(0, len(ws))), str(rng.choice(host_only)))
        glue = "" if rng.random() < 0.2 else " "          # (now and then no spaces at all: emoji and syllables next to capitals and digits)
        d = glue.join(ws).encode()
        docs.append(d)
        total += len(d)
    return docs


def test_emoji_and_korean_text_stay_on_the_device():
    """round 5: characters beyond the Basic Multilingual Plane (emoji ...: two bits per block of 64 code points, from the host normalizer's
    own functions) pass through the device normalizer and decoder, and Hangul syllables are decomposed by arithmetic (NFD: two or three
    conjoining jamo per syllable) - one emoji no longer sends its document to ICU.  Bytes == the host normalizer's; >= 98 % of the documents
    stay on the device (the corpus plants host-only characters in 1 %); ids of the raw path == ids of the host-normalized text; the device
    decoder gives back NFD of the text."""
    rng = np.random.default_rng(2026)
    docs = emoji_korean_corpus(rng, 150_000 if EMULATED else 5_000_000)
    raw, offs = tm.pack_documents(docs)
    for capcode, flag in ((2, 1), (2, 3), (2, 0), (0, 0), (0, 1)):
        v = tm.Vocab(synth.build_vocab([bytes([c]) for c in range(256)], capcode=capcode, charset=1, norm_flag=flag))
        got, goff, nfb = v.normalize_packed_device(raw, offs)
        exp, eoff = synth.normalize_batch(raw, offs, capcode, flag)
        assert (goff == eoff).all() and got.size == exp.size
        assert (got == exp).all(), "capcode %d flag %d" % (capcode, flag)
        if not (capcode == 0 and (flag & 1)):      # (without capcode the pass keeps lengths: there the syllables that NFD decomposes take the host path)
            assert nfb <= len(docs) // 50, "%d of %d documents took the host path (capcode %d flag %d)" % (nfb, len(docs), capcode, flag)
        if capcode == 2 and flag == 1:
            ids, toff, miss = v.tokenize_packed(exp, eoff)
            assert int(miss.sum()) == 0
            nchk = min(300, len(docs))
            gotids = v.tokenize(docs[:nchk])
            for k in range(nchk):
                assert (gotids[k] == ids[int(toff[k]):int(toff[k + 1])]).all()
            from tokenmonster_amd import _native as N
            import unicodedata
            out, ooff = v.decode_packed(ids[: int(toff[nchk])], np.ascontiguousarray(toff[: nchk + 1]))
            assert N.lib.tm_decode_host_docs() <= max(1, nchk // 20), "%d of %d documents were decoded on the host" % (N.lib.tm_decode_host_docs(), nchk)
            for k in range(nchk):
                assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == unicodedata.normalize("NFD", docs[k].decode()).encode(), docs[k]


def test_japanese_text_stays_on_the_device():
    """round 6: the voiced kana (が -> か + U+3099, ぱ -> は + U+309A, ヴ ヷ ヸ ヹ ヺ ゞ ヾ) under NFD - a kana and its mark, from a table the host normalizer
    fills (tm_normalize.cpp: build_kana_table).  One of them used to send its whole document to host ICU, which is every Japanese document.
    Bytes == the host normalizer's; with NFD and capcode 2 ALL documents but those with a voicing mark of their own (or a Latin mark behind a
    voiced kana) stay on the device; ids of the raw path == ids of the host-normalized text; the decoder gives back NFD of the text."""
    import unicodedata
    rng = np.random.default_rng(2028)
    words = ["これは", "日本語", "の", "テキスト", "です", "。", "、", "ございます", "がんばって", "ください", "データ", "プログラム", "ヴァイオリン", "東京", "大学", "で", "GPU", "を", "つかう",
             "ぱぴぷぺぽ", "ばびぶべぼ", "ザジズゼゾ", "ゞ", "ヾ", "ヷヸヹヺ", "ゔ", "Tokyo", "iPhone", "it's", "2が", "Aガ", "x"]
    docs, host_docs = [], 0
    total, target = 0, 120_000 if EMULATED else 3_000_000
    while total < target:
        n = int(rng.integers(1, 300))
        d = "".join(str(rng.choice(words)) + ("" if rng.random() < 0.7 else str(rng.choice([" ", "\n", "'", "1"]))) for _ in range(n))
        if rng.random() < 0.01:
            d += str(rng.choice(["か\u3099", "は\u309a", "が\u0301"]))           # what stays with the host: the marks by themselves, a further mark behind the character's own
            host_docs += 1
        docs.append(d.encode())
        total += len(docs[-1])
    docs += [("が" * 2000).encode(), ("パ" * 341 + "b").encode(), ("x" * 1022 + "がぎ").encode(), ("x" * 1023 + "ヴ").encode()]
    raw, offs = tm.pack_documents(docs)
    for capcode, flag in ((2, 1), (2, 3), (2, 0), (0, 1), (2, 1 | 4)):
        v = tm.Vocab(synth.build_vocab([bytes([c]) for c in range(256)], capcode=capcode, charset=1, norm_flag=flag))
        got, goff, nfb = v.normalize_packed_device(raw, offs)
        exp, eoff = synth.normalize_batch(raw, offs, capcode, flag)
        assert (goff == eoff).all() and got.size == exp.size
        assert (got == exp).all(), "capcode %d flag %d" % (capcode, flag)
        if capcode == 2 and flag in (1, 3):
            assert nfb == host_docs, "%d of %d documents took the host path, %d expected (capcode %d flag %d)" % (nfb, len(docs), host_docs, capcode, flag)
        if capcode == 2 and flag == 1:
            ids, toff, miss = v.tokenize_packed(exp, eoff)
            assert int(miss.sum()) == 0
            nchk = min(200, len(docs))
            gotids = v.tokenize(docs[:nchk])
            for k in range(nchk):
                assert (gotids[k] == ids[int(toff[k]):int(toff[k + 1])]).all()
            out, ooff = v.decode_packed(ids[: int(toff[nchk])], np.ascontiguousarray(toff[: nchk + 1]))
            for k in range(nchk):
                assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == unicodedata.normalize("NFD", docs[k].decode()).encode(), docs[k]


def test_hindi_and_thai_text_stays_on_the_device():
    """round 6: the three-byte combining marks of canonical class > 0 (the virama and nukta of the Indic scripts, the tone marks and the vowels below
    of Thai ...: tm_norm_masks.h, nm_ccc3) stay where they stand under NFD unless they have to change places with a neighbouring mark - decided at the
    later of two marks from a table of classes the host normalizer fills (tm_normalize.cpp: build_ccc_table).  One virama used to send its document to
    host ICU, which is every Hindi or Thai document.  Bytes == the host normalizer's; the documents without a letter that decomposes (Bengali's two-part
    vowels, the nukta letters written as one code point) and without marks out of order stay on the device; ids of the raw path == ids of the
    host-normalized text; the decoder gives back NFD of the text."""
    import unicodedata
    rng = np.random.default_rng(2029)
    hi = "यह हिन्दी का पाठ है और इसमें कई शब्द हैं जैसे कि विश्वविद्यालय प्रौद्योगिकी स्वतंत्रता ज़िन्दगी फ़िल्म क्या क्यों नहीं भारत दिल्ली मुम्बई १२३ GPU it's".split()
    th = ["ภาษาไทย", "อยู่", "ที่", "นี่", "กรุงเทพมหานคร", "ประเทศไทย", "สวัสดี", "ครับ", "ค่ะ", "น้ำ", "ผู้", "ใหญ่", "ไม่", "ได้", "เป็น", "คุณ", "รู้", "เรื่อง", "GPU", "Bangkok", "๑๒๓"]
    # (... and the characters NFD splits in two or three three-byte ones - Bengali's and Tamil's two-part vowel signs, the nukta letters as one code point, Kannada ೋ and Sinhala ෝ: nm_dec3)
    other = ["ລາວ", "ພາສາ", "བོད་སྐད", "မြန်မာ", "ខ្មែរ", "ભાષા", "தமிழ்", "ಕನ್ನಡ", "తెలుగు", "മലയാളം", "বাংলাদেশের", "কোনো", "হবে", "மொழி", "போகிறோம்", "കൊല്ലം", "\u0958\u093f\u0932\u093e", "\u095b\u094d\u092f\u093e\u0926\u093e", "ଓଡ଼ିଆ", "ಕೋಲಾರ", "ಯೋಗ", "හෝ", "සිංහල", "საქართველო", "ქართული", "ენა"]      # (... Georgian: lower-case letters of three bytes, class L)
    odd = ["\u0f73", "\u0f75", "क\u094d\u093c", "\u0e48\u0e38", "क\u0301", "\u1e09\u0e48", "\u0958\u0301"]         # what stays with the host: Tibetan vowel signs whose first part is a mark of class > 0, marks out of order, a Latin mark among them
    docs, host_docs = [], 0
    total, target = 0, 120_000 if EMULATED else 3_000_000
    while total < target:
        n = int(rng.integers(1, 300))
        lang = hi if rng.random() < 0.5 else th
        d = (" " if lang is hi else "").join(str(rng.choice(lang if rng.random() < 0.93 else other)) for _ in range(n))
        if rng.random() < 0.01:
            d += str(rng.choice(odd))
            host_docs += 1
        docs.append(d.encode())
        total += len(docs[-1])
    docs += [("क्" * 1000).encode(), ("อยู่" * 700 + "b").encode(), ("x" * 1022 + "क्ष").encode(), ("x" * 1021 + "น้ำ").encode()]
    raw, offs = tm.pack_documents(docs)
    for capcode, flag in ((2, 1), (2, 3), (2, 0), (0, 1), (2, 1 | 4)):
        v = tm.Vocab(synth.build_vocab([bytes([c]) for c in range(256)], capcode=capcode, charset=1, norm_flag=flag))
        got, goff, nfb = v.normalize_packed_device(raw, offs)
        exp, eoff = synth.normalize_batch(raw, offs, capcode, flag)
        assert (goff == eoff).all() and got.size == exp.size
        assert (got == exp).all(), "capcode %d flag %d" % (capcode, flag)
        if flag in (0, 1, 3) and not (capcode == 0 and (flag & 1)):      # (without capcode the pass keeps lengths: a character NFD splits in two is the host's)
            assert nfb <= host_docs, "%d of %d documents took the host path, at most %d expected (capcode %d flag %d)" % (nfb, len(docs), host_docs, capcode, flag)
        if capcode == 2 and flag == 1:
            ids, toff, miss = v.tokenize_packed(exp, eoff)
            assert int(miss.sum()) == 0
            nchk = min(200, len(docs))
            gotids = v.tokenize(docs[:nchk])
            for k in range(nchk):
                assert (gotids[k] == ids[int(toff[k]):int(toff[k + 1])]).all()
            out, ooff = v.decode_packed(ids[: int(toff[nchk])], np.ascontiguousarray(toff[: nchk + 1]))
            for k in range(nchk):
                assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == unicodedata.normalize("NFD", docs[k].decode()).encode(), docs[k]


def test_vietnamese_text_stays_on_the_device():
    """round 6: Latin Extended Additional (U+1E00..U+1EFF: what Vietnamese is written in beside the two-byte letters) under NFD - a letter and one
    or two combining marks per character, from a table the host normalizer fills (tm_normalize.cpp: build_lea_table).  One such character used
    to send its whole document to host ICU.  Bytes == the host normalizer's; >= 99 % of the documents stay on the device; ids of the raw path ==
    ids of the host-normalized text; the device decoder gives back NFD of the text."""
    import unicodedata
    from fuzz_cases import VIET
    rng = np.random.default_rng(2027)
    words = ["Việt", "Nam", "tiếng", "Hà", "Nội", "Thành", "phố", "Hồ", "Chí", "Minh", "đường", "Nguyễn", "Huệ", "phở", "bò", "ĐƯỜNG", "TIẾNG", "VIỆT", "người", "được", "những", "trường", "ƯỚC", "Ắt",
             "hello", "World", "HTTP", "it's", "ớt's", "Ế", "2ế", "x", "I"]
    docs = []
    total, target = 0, 120_000 if EMULATED else 3_000_000
    while total < target:
        n = int(rng.integers(1, 400))
        parts = []
        for _ in range(n):
            r = rng.random()
            parts.append(str(rng.choice(words)) if r < 0.8 else "".join(rng.choice(VIET[:60], size=int(rng.integers(1, 6)))))
            parts.append(str(rng.choice([" ", " ", " ", ", ", ". ", "\n", "'", "1"])))
        d = "".join(parts)
        if rng.random() < 0.005:
            d += str(rng.choice(["\u1e9b", "\u1e9e", "\u1ebf\u0323"]))           # what stays with the host: a two-byte base letter, no decomposition, a further mark behind the character's own
        docs.append(d.encode())
        total += len(docs[-1])
    docs += [("\u1ebf" * 2000).encode(), ("\u1ec6" * 700 + "b").encode(), ("x" * 1022 + "\u1ebf\u1ec7").encode()]
    raw, offs = tm.pack_documents(docs)
    for capcode, flag in ((2, 1), (2, 3), (2, 0), (0, 1)):
        v = tm.Vocab(synth.build_vocab([bytes([c]) for c in range(256)], capcode=capcode, charset=1, norm_flag=flag))
        got, goff, nfb = v.normalize_packed_device(raw, offs)
        exp, eoff = synth.normalize_batch(raw, offs, capcode, flag)
        assert (goff == eoff).all() and got.size == exp.size
        assert (got == exp).all(), "capcode %d flag %d" % (capcode, flag)
        if capcode == 2 and (flag & 1):
            assert nfb <= len(docs) // 100 + 1, "%d of %d documents took the host path (capcode %d flag %d)" % (nfb, len(docs), capcode, flag)
        if capcode == 2 and flag == 1:
            ids, toff, miss = v.tokenize_packed(exp, eoff)
            assert int(miss.sum()) == 0
            nchk = min(200, len(docs))
            gotids = v.tokenize(docs[:nchk])
            for k in range(nchk):
                assert (gotids[k] == ids[int(toff[k]):int(toff[k + 1])]).all()
            out, ooff = v.decode_packed(ids[: int(toff[nchk])], np.ascontiguousarray(toff[: nchk + 1]))
            for k in range(nchk):
                assert out[int(ooff[k]):int(ooff[k + 1])].tobytes() == unicodedata.normalize("NFD", docs[k].decode()).encode(), docs[k]


def test_one_child_chains_in_the_trie():
    """round 5: a node below which the trie is a chain of one-child nodes down to the next key is walked in ONE round - the chain's string
    (a record of three table entries, tm_tables.h "tails") against the text - instead of a byte per round, in the plain walk (step A1) and
    in the forward-delete walk (step A3).  Vocabularies made of long tokens: chains of every length around the minimum (5) and the record size
    (32), keys in the middle of a chain, chains that branch late, chains entered through a suffix link; texts that match them fully, break
    off at every byte, and end inside them.  ids == the oracle's for every document."""
    rng = np.random.default_rng(55)
    words = [b"alpha", b"beta", b"gamma", b"delta", b"epsilon", b"zeta", b"eta", b"theta", b"iota", b"kappa", b"lambda", b"mu"]
    longs = []
    for _ in range(60):
        n = int(rng.integers(2, 8))
        t = b"".join(b" " + words[int(rng.integers(0, len(words)))] for _ in range(n))[:40]
        longs.append(t)
        if rng.random() < 0.5:
            longs.append(t[: int(rng.integers(3, len(t) + 1))])            # a key in the middle of the chain
        if rng.random() < 0.3 and len(t) > 12:
            longs.append(t[: len(t) - 3] + b"xyz"[: 40 - (len(t) - 3)])      # a chain that branches three bytes before its end
        if rng.random() < 0.4:
            longs.append(t[1:])                                              # the same without its first byte: suffix links lead into the chain
    for cut in range(4, 41):                                                 # one chain of every length
        longs.append((b" q" + b"abcdefghijklmnopqrstuvwxyz0123456789-+*/")[:cut])
    singles = [bytes([c]) for c in b" abcdefghijklmnopqrstuvwxyz0123456789-+*/."]
    # a chain that every position of a run of one letter stands at the head of: more chain heads under one wavefront than its task list holds
    # (the positions are marked and walked again from their first byte)
    longs += [b"z" * 40, b"zz", b"zzz", b"y" * 33, b"yy"]
    for capcode in (0, 2):
        toks = sorted(set(longs + singles + [b" " + x for x in words] + words + ([b"D"] if capcode == 2 else [])))
        img = synth.build_vocab(toks, capcode=capcode, charset=1)
        v, orc = tm.Vocab(img), Oracle(img)
        docs = []
        for _ in range(400):
            parts = []
            for _ in range(int(rng.integers(1, 30))):
                t = longs[int(rng.integers(0, len(longs)))]
                r = rng.random()
                if r < 0.5:
                    parts.append(t)
                elif r < 0.8:
                    parts.append(t[: int(rng.integers(1, len(t) + 1))])    # breaks off somewhere inside
                else:
                    k = int(rng.integers(0, len(t)))
                    parts.append(t[:k] + b"." + t[k + 1:])                   # one byte wrong
                if rng.random() < 0.2:
                    parts.append(words[int(rng.integers(0, len(words)))])    # a word without its space: forward-delete probes
            docs.append(b"".join(parts))
        docs += [longs[0][:k] for k in range(1, len(longs[0]) + 1)]          # the text ends inside a chain, at every byte
        docs += [b"z" * 1000, b"z" * 39 + b"a" + b"z" * 300 + b"y" * 500, (b"z" * 37 + b"q") * 40, b"y" * 32 + b"z" * 41 + b"y" * 34 + b" alpha" + b"z" * 700]
        text, offs = tm.pack_documents(docs)
        ids, toff, missing = v.tokenize_packed(text, offs)
        for d in range(len(docs)):
            exp, miss = orc.tokenize(docs[d])
            got = ids[int(toff[d]):int(toff[d + 1])]
            assert got.size == exp.size and (got == exp).all() and miss == int(missing[d]), (capcode, d, docs[d][:80])


@pytest.mark.parametrize("flags,wide", [(0, False), (32768, False), (1024, False), (0, True), (32768, True)])
def test_walk_with_an_id_per_byte_and_more(flags, wide):
    """K4 on texts that emit an id for every byte and, where delete tokens follow, more ids than bytes: the position-staging walk (k_emit_list)
    fills its second phase's rounds (64 slots of a segment per round: up to four and more), runs out of room in front of the byte being read and
    switches a segment to direct stores half way, and meets forward-delete states whose ids it writes itself; flags 32768 / 1024: the id-staging
    form, and every id stored directly.  All against the oracle."""
    from tokenmonster_amd import _native as N
    rng = np.random.default_rng(4242)
    alphabet = b"qrstuvwx"          # (letters the fuzz vocabulary below has no words of: every one of them is a token of its own)
    toks = [bytes([c]) for c in alphabet + b" D.\n"] + [b" " + bytes([c]) for c in alphabet] + [b"D " + bytes([c]) for c in alphabet[:4]] + [b"qr", b"st", b" qr", b"D qr"]
    toks = list(dict.fromkeys(toks + fuzz_vocab_tokens(rng, 2, 60)))       # + the fuzz vocabulary: its space-prefixed words bring the forward-delete branches
    if wide:      # more than 65 536 ids: the rows carry u32 ids (round 6: two planes for k_emit_list<true>; one plane of words under hook 15) - 66 000 tokens no text here contains
        toks += [bytes([0x7F, 0x30 + k % 40, 0x30 + (k // 40) % 40, 0x30 + k // 1600]) for k in range(66_000)]
    img = synth.build_vocab(toks, capcode=2, charset=1, with_unk=True)
    v = tm.Vocab(img)
    assert (v.n_ids() > 65536) == wide
    orc = Oracle(img)
    docs = []
    for n in (1, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 5000):
        docs.append(bytes(rng.choice(np.frombuffer(alphabet, dtype=np.uint8), size=n)))                                   # one id per byte
        docs.append(bytes(rng.choice(np.frombuffer(alphabet + b"   ", dtype=np.uint8), size=n)))                             # ' x' tokens: fewer ids than bytes
        docs.append(b"".join(bytes(rng.choice([b"D a", b"D b", b"a", b"D", b" ", b"D ab", b"b."])) for _ in range(n))[:max(n, 1)])   # markers and words of the fuzz vocabulary
        docs.append(fuzz_text(rng, 2, n))
    oracle_stats(reset=True)
    old = N.lib.tm_debug_flags(flags)
    try:
        check_docs(v, orc, docs, "an id per byte, flags %d" % flags)
    finally:
        N.lib.tm_debug_flags(old)
    st = oracle_stats()
    assert st["s1b"] + st["s2b"] + st["s3b"] > 0, st                                           # forward-delete states were walked
    assert any(orc.tokenize(d)[0].size >= 0.95 * len(d) >= 240 for d in docs)                  # and segments with (nearly) an id per byte
