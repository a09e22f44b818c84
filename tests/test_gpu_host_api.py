"""-m gpu: the host-buffer entry points a Go caller binds (tokenmonster_amd/csrc/tm_host.hip): lanes (stream + grow-only workspace
per concurrent caller, training/tokenmonsterserver.go:363-378 calls Tokenize from many goroutines at once) and the chunked
host-to-host pipeline tm_tokenize_pipeline."""
import threading

import numpy as np
import pytest

import tokenmonster_amd as tm
from oracle_bind import Oracle
from tokenmonster_amd import synth

pytestmark = pytest.mark.gpu


def _ids_from_bytes(blob, enc):
    b = blob.reshape(-1, enc).astype(np.uint32)
    out = b[:, 0] | (b[:, 1] << 8)
    if enc >= 3:
        out |= b[:, 2] << 16
    return out


@pytest.fixture(scope="module")
def setup():
    img = synth.synth_vocab(synth.ENGLISHCODE, 6000, capcode=2, norm_flag=1, level=3, seed=0x484F5354)
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 3_000_000, seed=71)
    text, offs = synth.normalize_batch(raw, roffs, 2, 1)
    return img, raw, roffs, text, offs


@pytest.mark.parametrize("raw_mode", [False, True])
@pytest.mark.parametrize("pinned", [False, True])
def test_pipeline_equals_single_batch(setup, raw_mode, pinned):
    img, raw, roffs, text, offs = setup
    v = tm.Vocab(img)
    ids, toff, miss = v.tokenize_packed(text, offs)
    src, soff = (raw, roffs) if raw_mode else (text, offs)
    keep = []
    out = None
    if pinned:
        pin = tm.PinnedBuffer(src.size)
        pin.array[:] = src
        pout = tm.PinnedBuffer(2 * ids.size + 64)
        keep += [pin, pout]
        src, out = pin.array, pout.array
    # small chunks: many chunks per lane, chunk boundaries everywhere, one document longer than a chunk
    for chunk, lanes in ((200_000, 3), (40_000, 4), (1 << 30, 1)):
        blob, boff, bmiss, enc, st = v.tokenize_pipeline(src, soff, raw=raw_mode, chunk_bytes=chunk, lanes=lanes, out=out)
        assert enc == 2 and (bmiss == miss).all()
        assert (boff == toff * np.uint64(2)).all()
        assert (_ids_from_bytes(np.asarray(blob), 2) == ids).all()
        assert st["input_pinned"] == int(pinned) and st["chunks"] >= 1
    # 4-byte form, TM_E_NOSPACE path (out=None starts with a guess), empty corpus
    blob, boff, _, enc, _ = v.tokenize_pipeline(src, soff, raw=raw_mode, encoding_length=4, chunk_bytes=300_000, out=np.empty(16, np.uint8))
    assert enc == 4 and (_ids_from_bytes(np.asarray(blob), 4) == ids).all() and int(boff[-1]) == 4 * ids.size
    blob, boff, _, _, _ = v.tokenize_pipeline(np.zeros(0, np.uint8), np.zeros(1, np.uint64), raw=raw_mode)
    assert blob.size == 0 and boff.tolist() == [0]
    del keep


def test_concurrent_callers_take_lanes(setup):
    """8 threads tokenize different batches on ONE vocabulary at once (cgo calls run on distinct OS threads): every result is
    bit-exact, repeated calls reuse the lane workspaces (no growth of device memory after the first round)."""
    img, raw, roffs, text, offs = setup
    v, orc = tm.Vocab(img), Oracle(img)
    nd = offs.size - 1
    parts = []
    for t in range(8):
        d0, d1 = nd * t // 8, nd * (t + 1) // 8
        sub_off = offs[d0:d1 + 1] - offs[d0]
        parts.append((text[int(offs[d0]):int(offs[d1])], sub_off))
    expect = [[orc.tokenize(p[int(o[d]):int(o[d + 1])])[0] for d in range(0, o.size - 1, 9)] for p, o in parts]
    errors = []

    def work(t):
        try:
            p, o = parts[t]
            for _ in range(6):
                ids, toff, _ = v.tokenize_packed(p, o)
                counts, _ = v.count_packed(p, o)
                for k, d in enumerate(range(0, o.size - 1, 9)):
                    got = ids[int(toff[d]):int(toff[d + 1])]
                    if got.size != expect[t][k].size or (got != expect[t][k]).any():
                        raise AssertionError("thread %d doc %d differs" % (t, d))
        except Exception as e:     # noqa: BLE001
            errors.append(e)

    ths = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors[:2]


def test_vocab_save_writes_back_the_loaded_image(setup, tmp_path):
    """Save (go/tokenmonster.go:2602-2653): the library never mutates a vocabulary, so the saved file is the loaded image, byte for byte,
    and loads again"""
    import ctypes as C
    from tokenmonster_amd import _native as N
    img = setup[0]
    v = tm.Vocab(img)
    p, n = C.c_void_p(), C.c_size_t()
    N.check(N.lib.tm_vocab_image(v.handle, C.byref(p), C.byref(n)))
    assert C.string_at(p.value, n.value) == bytes(img)
    path = str(tmp_path / "saved.vocab")
    N.check(N.lib.tm_vocab_save(v.handle, path.encode()))
    assert open(path, "rb").read() == bytes(img)
    v2 = tm.load(path)
    assert v2.n_ids() == v.n_ids() and v2.n_info() == v.n_info()


def test_pipeline_capital_heavy_text_retries_with_a_larger_workspace(setup):
    """capcode turns "A.B.C" into 2.5 times as many bytes: a lane's workspace (1.5 x the raw size plus headroom) is too small, the
    normalizer answers TM_E_LIMIT and the lane uploads the chunk again into the worst-case workspace (tm_host.hip: lane_compute) —
    here on a chunk that had been prefetched behind a lower-case one into a workspace that looked large enough."""
    img = setup[0]
    v = tm.Vocab(img)
    rng = np.random.default_rng(5)
    docs = [b"plain lower case words again and again " * 1300 for _ in range(80)]                       # 4 MB that do not grow
    docs += [b".".join(bytes([65 + int(c)]) for c in rng.integers(0, 26, 25_000)) for _ in range(100)]   # 5 MB that grow 2.5 x
    raw = np.frombuffer(b"".join(docs), dtype=np.uint8).copy()
    roffs = np.zeros(len(docs) + 1, dtype=np.uint64)
    roffs[1:] = np.cumsum([len(x) for x in docs])
    text, offs = synth.normalize_batch(raw, roffs, 2, 1)
    assert text.size > 1.7 * raw.size          # (the point of the test)
    ids, toff, miss = v.tokenize_packed(text, offs)
    for chunk, lanes in ((2 << 20, 1), (2 << 20, 2), (1 << 30, 1)):
        blob, boff, bmiss, enc, st = v.tokenize_pipeline(raw, roffs, raw=True, chunk_bytes=chunk, lanes=lanes)
        assert (boff == toff * np.uint64(enc)).all() and (bmiss == miss).all()
        assert (_ids_from_bytes(np.asarray(blob), enc) == ids).all()


def test_small_transfers_survive_a_wrapping_mailbox(setup):
    """Counters, offsets and server-sized batches travel through a pinned mailbox and a copy kernel (tm_kernels.hip: small_d2h /
    small_h2d).  Test hook bit 13 shrinks the mailbox to 64 KiB with 16 KiB transfers, so that this corpus wraps it many times and
    also takes the copy-engine path for what no longer fits."""
    from tokenmonster_amd import _native as N
    img, raw, roffs, text, offs = setup
    v = tm.Vocab(img)
    ids, toff, miss = v.tokenize_packed(text, offs)
    old = N.lib.tm_debug_flags(8192)
    try:
        assert N.lib.tm_debug_flags(-1) == 8192
        v2 = tm.Vocab(img)
        ids2, toff2, miss2 = v2.tokenize_packed(text, offs)
        assert (ids2 == ids).all() and (toff2 == toff).all() and (miss2 == miss).all()
        for k in range(0, offs.size - 1, 97):                     # server-sized batches: the text itself goes through the mailbox
            a, b = int(offs[k]), int(offs[min(k + 3, offs.size - 1)])
            sub_off = (offs[k:min(k + 3, offs.size - 1) + 1] - offs[k]).astype(np.uint64)
            i3, t3, _ = v2.tokenize_packed(text[a:b], sub_off)
            assert (i3 == ids[int(toff[k]):int(toff[k]) + i3.size]).all()
        blob, boff, bmiss, enc, _ = v2.tokenize_pipeline(raw, roffs, raw=True, chunk_bytes=40_000, lanes=4)
        assert (boff == toff * np.uint64(enc)).all() and (bmiss == miss).all()
        assert (_ids_from_bytes(np.asarray(blob), enc) == ids).all()
    finally:
        N.lib.tm_debug_flags(old)


def test_vocabulary_freed_under_an_asynchronous_pass():
    """tm_vocab_free parks the device block of a vocabulary for the next tm_vocab_load (the trainvocab worker loads and frees one per
    candidate).  A scoring pass launched through an asynchronous entry point may still be in flight then: the block must not be refilled
    under its kernels.  Candidate A is scored through tm_score_device (returns with kernels queued) and freed at once, candidate B (other
    tokens, same size: it takes a parked block - A's, if nothing protects it) is loaded and scored before anybody has synchronized;
    A's histogram must be A's."""
    import ctypes as C
    from tokenmonster_amd import _native as N
    from conftest import EMULATED
    img_a = synth.synth_vocab(synth.ENGLISHCODE, 2000 if EMULATED else 9000, capcode=2, norm_flag=1, level=5, seed=0x41414141)
    img_b = synth.synth_vocab(synth.ENGLISHCODE, 2000 if EMULATED else 9000, capcode=2, norm_flag=1, level=5, seed=0x42424242)
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, (256 << 10) if EMULATED else (32 << 20), seed=72)     # (the emulated device runs ~1 MiB/s and has nothing in flight)
    text, _ = synth.normalize_batch(raw, roffs, 2, 1)
    data = np.ascontiguousarray(text)
    ds1, ds2 = C.c_void_p(), C.c_void_p()
    N.check(N.lib.tm_dataset_upload(N.ptr(data), int(data.size), C.byref(ds1)))
    N.check(N.lib.tm_dataset_upload(N.ptr(data), int(data.size), C.byref(ds2)))
    try:
        ref = tm.Vocab(img_a)
        n_ids = ref.n_ids()
        full_a, tit, ms = np.zeros(n_ids, dtype=np.uint32), C.c_uint64(), np.zeros(32, dtype=np.uint8)
        N.check(N.lib.tm_score(ref.handle, ds1, None, None, 0, N.ptr(full_a), C.byref(tit), N.ptr(ms)))
        full_a_tokens = tit.value
        head = min(1 << 20, int(data.size) // 2)            # the oracle pins the head of the synchronous pass
        exp = Oracle(img_a).score(text[:head])
        so, sl, got = np.array([0], dtype=np.uint64), np.array([head], dtype=np.uint64), np.zeros(n_ids, dtype=np.uint32)
        N.check(N.lib.tm_score(ref.handle, ds1, N.ptr(so), N.ptr(sl), 1, N.ptr(got), C.byref(tit), N.ptr(ms)))
        assert (got == exp[0]).all() and tit.value == exp[1]
        ref.close()
        for trial in range(1 if EMULATED else 4):
            tm.Vocab(img_b).close()                         # a parked block of the right size is waiting
            a = tm.Vocab(img_a)
            dev_hist, words = C.c_void_p(), C.c_uint64()
            N.check(N.lib.tm_score_device(a.handle, ds1, None, None, 0, None, C.byref(dev_hist), C.byref(words)))
            a.close()                                       # kernels of the pass may still be running
            b = tm.Vocab(img_b)                             # takes a parked block
            assert b.n_ids() == n_ids
            sb = np.zeros(n_ids, dtype=np.uint32)
            N.check(N.lib.tm_score(b.handle, ds2, None, None, 0, N.ptr(sb), C.byref(tit), N.ptr(ms)))
            sa = np.zeros(n_ids, dtype=np.uint32)
            N.check(N.lib.tm_score_read(b.handle, ds1, N.ptr(sa), C.byref(tit), N.ptr(ms)))     # (the vocabulary only says how many ids there are)
            assert tit.value == full_a_tokens and (sa == full_a).all(), "trial %d: the histogram of the freed vocabulary is not its own" % trial
            assert not (sb == full_a).all()
            b.close()
    finally:
        N.lib.tm_dataset_free(ds1)
        N.lib.tm_dataset_free(ds2)


def test_vocabulary_block_export_import(setup):
    """tm_vocab_block_export / tm_vocab_block_import: the device block of a vocabulary, copied into an empty vocabulary of the same shape,
    tokenizes and scores like the original (what dist.broadcast_vocab does between ranks with one RCCL broadcast); what needs host tables
    is refused."""
    import ctypes as C
    from tokenmonster_amd import _native as N
    img, raw, roffs, text, offs = setup
    a = tm.Vocab(img)
    desc, src_ptr, nbytes = a.export_block()
    b, dst_ptr, nbytes_b = tm.Vocab.import_block(desc, 0)
    assert nbytes_b == nbytes and dst_ptr != src_ptr
    N.check(N.lib.tm_device_copy(C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes))
    ia, oa, ma = a.tokenize_packed(text, offs)
    out_a, _ = a.decode_packed(ia, oa, raw=True)
    a.close()                                                # the copy stands on its own
    ib, ob, mb = b.tokenize_packed(text, offs)
    assert (ia == ib).all() and (oa == ob).all() and (ma == mb).all()
    assert b.n_ids() == len(Oracle(img).score(text[:10])[0])
    out, ooff = b.decode_packed(ib, ob, raw=True)            # the device gather works (reverse tables are in the block) ...
    assert out.tobytes() == out_a.tobytes()
    with pytest.raises(N.TokenMonsterHipError):               # ... the streaming decoder and Save need host tables
        tm.Decoder(b)


def test_latin_text_is_decoded_on_the_device():
    """Capcode decoding of accented Latin text (two-byte characters, combining marks, curly quotes) happens in k_dec_capcode:
    tm_decode_host_docs() counts the documents of the last call that were left to the host decoder."""
    import ctypes as C
    from tokenmonster_amd import _native as N
    import numpy as np
    from tokenmonster_amd import synth
    img = synth.build_vocab([bytes([c]) for c in range(256)], capcode=0, charset=1)
    a = tm.Vocab(b"\x02" + bytes(img[1:]))                    # 256 one-byte tokens, header switched to capcode 2: the decoder's input is the document
    all_ids = np.arange(a.n_ids(), dtype=np.uint32)
    rb, ro = a.decode_packed(all_ids, np.arange(a.n_ids() + 1, dtype=np.uint64), raw=True)
    id_of = np.zeros(256, dtype=np.uint32)
    for i in range(a.n_ids()):
        if int(ro[i + 1] - ro[i]) == 1:
            id_of[int(rb[int(ro[i])])] = i
    docs = ["Wdécouvert Dà Cparis, Wcœur Dde l\u2019été: Cgarçon".encode(), "Wstraße Cøre Cn\u0303andu\u0301 Wñu".encode(), ("Cé" * 40 + " Wàb c").encode(), b"Wplain ascii"]
    text, toff = tm.pack_documents(docs)
    tok = id_of[text]
    out, ooff = a.decode_packed(tok, toff, raw=False)
    assert N.lib.tm_decode_host_docs() == 0
    for d, doc in enumerate(docs):
        dec = a.decoder()
        exp = dec.decode(tok[int(toff[d]):int(toff[d + 1])]) + dec.flush()
        assert out[int(ooff[d]):int(ooff[d + 1])].tobytes() == exp, doc
    assert out[int(ooff[1]):int(ooff[2])].tobytes() == "STRAßE Øre N\u0303andu\u0301 ÑU".encode()      # (ß has no simple upper case)
    # what is left to the host decoder: an upper-case form of another length (ı -> I), case in three bytes (fullwidth Latin), malformed UTF-8;
    # another two-byte lead for the upper-case form (ÿ -> Ÿ) and caseless three-byte scripts stay on the device since round 4
    for other, on_host in [("W\u4e2d\u6587 Cの".encode(), 0), ("Cÿ Wÿz".encode(), 0), ("Wж Cσ Wֆ".encode(), 0), ("Cı".encode(), 1), ("C\uff41".encode(), 1), (b"W\xc3(", 1)]:
        t2, o2 = tm.pack_documents([other, b"Wascii"])
        out2, oo2 = a.decode_packed(id_of[t2], o2, raw=False)
        assert N.lib.tm_decode_host_docs() == on_host, other
        dec = a.decoder()
        assert out2[:int(oo2[1])].tobytes() == dec.decode(id_of[t2][:int(o2[1])]) + dec.flush(), other


@pytest.mark.parametrize("capcode,with_unk", [(2, False), (0, True)])
def test_vocab_build_equals_build_image_then_load(capcode, with_unk):
    """tm_vocab_build (token list -> tables -> device in one call: the trainvocab worker's per-candidate step, training/trainvocab.go:530-907)
    gives the vocabulary tm_vocab_load(tm_build_vocab(same list)) gives: the same .vocab bytes (written from its records on request), the same
    ids, the same scoring histogram, and the same device block byte for byte; tm_vocab_build_all replicates it over the members of a handle."""
    import ctypes as C
    from conftest import fuzz_text, fuzz_vocab_tokens
    from tokenmonster_amd import _native as N, multi
    from tokenmonster_amd.vocab import VocabBlock
    from test_gpu_parity import _score
    rng = np.random.default_rng(900 + capcode)
    toks = fuzz_vocab_tokens(rng, capcode, 400)
    img = synth.build_vocab(toks, capcode=capcode, charset=1, with_unk=with_unk)
    a = tm.Vocab(img)
    b = tm.Vocab.from_tokens(toks, capcode=capcode, charset=1, with_unk=with_unk)
    assert b.image() == bytes(img)
    ma, pa, na = a.export_block()
    mb, pb, nb = b.export_block()
    assert ma == mb and na == nb
    ha, hb = np.empty(na, dtype=np.uint8), np.empty(nb, dtype=np.uint8)
    for ptr, host in ((pa, ha), (pb, hb)):
        pin = tm.PinnedBuffer(host.size)
        N.check(N.lib.tm_device_copy(C.c_void_p(pin.array.ctypes.data), C.c_void_p(ptr), host.size))
        host[:] = pin.array
    total = 256 + sum(((x + 255) & ~255) for x in VocabBlock.from_buffer_copy(ma).part_bytes)
    assert (ha[:total] == hb[:total]).all()                       # the tables themselves (the tail of a block is allocation slack)
    data = np.frombuffer(fuzz_text(rng, capcode, 50_000), dtype=np.uint8)
    offs = np.array([0, 10_000, 10_000, 31_111, 50_000], dtype=np.uint64)
    ia, oa, xa = a.tokenize_packed(data, offs)
    ib, ob, xb = b.tokenize_packed(data, offs)
    assert (ia == ib).all() and (oa == ob).all() and (xa == xb).all()
    sa, sb = _score(a, data), _score(b, data)
    assert (sa[0] == sb[0]).all() and sa[1] == sb[1] and (sa[2] == sb[2]).all()
    g = multi.Devices([0, 0, 0])
    try:
        vs = multi.VocabSet.from_tokens(g, toks, capcode=capcode, charset=1, with_unk=with_unk)
        ds = multi.DatasetSet(g, data)
        sm = ds.score(vs)
        assert (sm[0] == sa[0]).all() and sm[1] == sa[1] and (sm[2] == sa[2]).all()
        ds.close(); vs.close()
    finally:
        g.close()


def test_vocabularies_come_and_go_between_calls():
    """a vocabulary per call, loaded into the parked block of the one before (the trainvocab worker's life, training/trainvocab.go:1827-1829): the
    next load must not trip over the last kernels of the vocabulary that was freed - its events used to outlive the streams they were recorded on,
    and the runtime's answer to a wait on such an event surfaced as a 'kernel launch' error several calls later"""
    from conftest import fuzz_text, fuzz_vocab_tokens
    from oracle_bind import Oracle
    rng = np.random.default_rng(4242)
    for i in range(60):
        toks = fuzz_vocab_tokens(rng, 2, 80 + i)
        img = synth.build_vocab(toks, capcode=2, charset=1)
        v = tm.Vocab(img)
        docs = [fuzz_text(rng, 2, int(n)) for n in rng.integers(1, 3000, size=12)]
        ids, _ = v.tokenize_normalized(docs)
        if i % 10 == 0:
            orc = Oracle(img)
            for d, doc in enumerate(docs):
                exp, _ = orc.tokenize(doc)
                assert ids[d].size == exp.size and (ids[d] == exp).all()
        del v


@pytest.mark.parametrize("capcode,charset", [(2, 1), (0, 1), (2, 2)])
def test_tables_laid_out_by_use_give_the_same_results(capcode, charset):
    """tm_vocab_tune renumbers the trie's nodes by how often a sample uses them and writes every table again: ids, counts, scoring histogram,
    decode and the saved image are those of the untuned vocabulary - on the sample, on other text, after a second tuning on another sample"""
    from conftest import fuzz_text, fuzz_vocab_tokens
    from test_gpu_parity import _score
    rng = np.random.default_rng(7100 + 10 * capcode + charset)
    toks = fuzz_vocab_tokens(rng, capcode, 600)
    if charset == 2:
        toks = sorted({bytes(b for ch in t for b in (ch, 0))[:40] for t in toks if len(t) <= 20})
    img = synth.build_vocab(toks, capcode=capcode, charset=charset, with_unk=(capcode == 0))
    plain, tuned = tm.Vocab(img), tm.Vocab(img)

    def text(n):
        t = fuzz_text(rng, capcode, n)
        return bytes(b for ch in t[: n // 2] for b in (ch, 0)) if charset == 2 else t
    docs = [text(int(n)) for n in rng.integers(0, 4000, size=80)] + [b"", text(70_000)]
    data = np.frombuffer(text(60_000), dtype=np.uint8)
    exp_ids, exp_miss = plain.tokenize_normalized(docs)
    exp_cnt = plain.tokenize_count(docs)
    exp_score = _score(plain, data)
    for sample in (b"".join(docs[:40]), text(200_000), b"", text(3)):
        tuned.tune(sample)
        got_ids, got_miss = tuned.tokenize_normalized(docs)
        assert all(g.size == e.size and (g == e).all() for g, e in zip(got_ids, exp_ids)) and (got_miss == exp_miss).all()
        got_cnt = tuned.tokenize_count(docs)
        assert all((np.asarray(g) == np.asarray(e)).all() for g, e in zip(got_cnt, exp_cnt))
        s = _score(tuned, data)
        assert (s[0] == exp_score[0]).all() and s[1] == exp_score[1] and (s[2] == exp_score[2]).all()
        assert tuned.image() == plain.image()
    text0, offs0 = tm.pack_documents(docs)
    ids0, toff0, _ = plain.tokenize_packed(text0, offs0)
    a, ao = plain.decode_packed(ids0, toff0, raw=True)
    b, bo = tuned.decode_packed(ids0, toff0, raw=True)
    assert (a == b).all() and (ao == bo).all()
    # tm_vocab_load_sample: the layout by use from the start (also with an empty sample = plain tm_vocab_load)
    for sample in (b"".join(docs[:40]), b""):
        v = tm.Vocab(img, sample=np.frombuffer(sample, dtype=np.uint8))
        got_ids, got_miss = v.tokenize_normalized(docs)
        assert all(g.size == e.size and (g == e).all() for g, e in zip(got_ids, exp_ids)) and (got_miss == exp_miss).all()
        s = _score(v, data)
        assert (s[0] == exp_score[0]).all() and s[1] == exp_score[1] and (s[2] == exp_score[2]).all()
        assert v.image() == plain.image()
        b, bo = v.decode_packed(ids0, toff0, raw=True)
        assert (a == b).all() and (ao == bo).all()
        v.tune(text(50_000))                    # ... and laid out once more on another sample
        got_ids, got_miss = v.tokenize_normalized(docs)
        assert all(g.size == e.size and (g == e).all() for g, e in zip(got_ids, exp_ids)) and (got_miss == exp_miss).all()


def _pinned_copy(a):
    p = tm.PinnedBuffer(max(int(a.size), 16))
    p.array[: a.size] = a
    return p


def test_ring_equals_single_batch():
    """tm_tokenize_pipeline on page-locked buffers runs on the RING (tm_host.hip: no host round trip inside a chunk - the kernels behind the
    normalizer pass are launched over a bound and look the segment / id counts up on the device).  Chunks of every size of the ramp, several
    id widths, the TM_E_NOSPACE answer, two calls in a row on the same slots."""
    from conftest import EMULATED
    img = synth.synth_vocab(synth.ENGLISHCODE, 3000, capcode=2, norm_flag=1, level=3, seed=0x52494E47)
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 400_000 if EMULATED else 6_000_000, seed=73)
    text, offs = synth.normalize_batch(raw, roffs, 2, 1)
    v = tm.Vocab(img)
    ids, toff, miss = v.tokenize_packed(text, offs)
    pin, pout = _pinned_copy(raw), tm.PinnedBuffer(4 * ids.size + 64)
    for chunk, lanes, width in ((30_000, 2, 0), (30_000, 4, 3), (90_000, 3, 4), (9_000, 4, 2)):
        blob, boff, bmiss, enc, st = v.tokenize_pipeline(pin.array[: raw.size], roffs, raw=True, encoding_length=width, chunk_bytes=chunk, lanes=lanes, out=pout.array)
        assert st["ring"] == 1 and st["ring_exact_chunks"] == 0 and st["chunks"] > 3, st
        assert enc == (width or 2) and (bmiss == miss).all()
        assert (boff == toff * np.uint64(enc)).all()
        assert (_ids_from_bytes(np.asarray(blob), enc) == ids).all()
        assert st["normalized_bytes"] == text.size and st["host_fallback_docs"] == 0
    small = tm.PinnedBuffer(64)
    blob, boff, _, enc, st = v.tokenize_pipeline(pin.array[: raw.size], roffs, raw=True, chunk_bytes=30_000, out=small.array)      # (the wrapper retries with a pageable buffer of the size asked for)
    assert int(boff[-1]) == 2 * ids.size and (_ids_from_bytes(np.asarray(blob), 2) == ids).all()
    # test hook 16: the thin kernels of a chunk one by one (what a chunk of more than 2^18 pieces takes) instead of fused
    from tokenmonster_amd import _native as N
    old = N.lib.tm_debug_flags(65536)
    try:
        blob, boff, bmiss, enc, st = v.tokenize_pipeline(pin.array[: raw.size], roffs, raw=True, chunk_bytes=30_000, lanes=2, out=pout.array)
        assert st["ring"] == 1 and st["ring_exact_chunks"] == 0
        assert (boff == toff * np.uint64(enc)).all() and (bmiss == miss).all() and (_ids_from_bytes(np.asarray(blob), enc) == ids).all()
    finally:
        N.lib.tm_debug_flags(old)


def test_ring_takes_vocabularies_with_byte_level_flags():
    """A vocabulary with quotemarks / collapse / trim / leadingspace / unixlines / accents (training/README.md:110-123) runs on the ring too: its
    filter pass (tm_norm.hip: k_pf_*) is enqueued in front of the normalizer pass, which is then launched over the pieces of the RAW documents - a
    bound - and takes the count of the filtered documents' pieces from the device.  ids == the host normalizer's text through the batch path."""
    from conftest import EMULATED
    img0 = synth.synth_vocab(synth.ENGLISHCODE, 3000, capcode=2, norm_flag=1, level=3, seed=0x52494E47)
    raw0, roffs0 = synth.synth_corpus(synth.ENGLISHCODE, 200_000 if EMULATED else 3_000_000, seed=76)
    docs = [bytes(raw0[int(roffs0[d]):int(roffs0[d + 1])]) for d in range(roffs0.size - 1)]
    extra = [b"  Hello   World \r\n", "\u201cQuoted\u201d  caf\u00e9  \u2018x\u2019 ".encode(), b"", b" \t ", b"one  two \r\n three\r\n\r\n  four  ",
             ("some  plain  words \u2018q\u2019 \r\n" * 40).encode(), ("a \u2019b" * 2730 + "  " + "\u2018c\u2019 \u00e9\u00f1" * 2000 + "  d \u201d x").encode(),
             b" " * 3000 + b"padded  both   ends\r\n" + b" " * 5000, b"x"]
    for i, e in enumerate(extra):
        docs.insert((i * len(docs)) // len(extra), e)
    raw = np.frombuffer(b"".join(docs), dtype=np.uint8).copy()
    roffs = np.zeros(len(docs) + 1, dtype=np.uint64)
    roffs[1:] = np.cumsum([len(x) for x in docs])
    pin = _pinned_copy(raw)
    for flag in (2 | 8 | 16 | 32 | 128, 255, 1 | 64, 1 | 4 | 16):
        img = bytes(img0[:2]) + bytes([flag]) + bytes(img0[3:])
        v = tm.Vocab(img)
        text, offs = synth.normalize_batch(raw, roffs, 2, flag)
        ids, toff, miss = v.tokenize_packed(text, offs)
        pout = tm.PinnedBuffer(2 * ids.size + 64)
        for chunk, lanes in ((25_000, 3), (70_000, 2)):
            blob, boff, bmiss, enc, st = v.tokenize_pipeline(pin.array[: raw.size], roffs, raw=True, chunk_bytes=chunk, lanes=lanes, out=pout.array)
            # (the long document of marks grows by more than the 25 % the segment kernels of a chunk are launched over when neither accents nor
            # lowercase shrinks it again, and is then a chunk for the exact path - as without a filter pass)
            assert st["ring"] == 1 and st["ring_exact_chunks"] <= (0 if flag & 6 else 1) and st["chunks"] > 3, (flag, st)
            assert st["host_fallback_docs"] == 0 and st["normalized_bytes"] == text.size, (flag, st)
            assert (boff == toff * np.uint64(enc)).all() and (bmiss == miss).all(), flag
            assert (_ids_from_bytes(np.asarray(blob), enc) == ids).all(), flag


def test_ring_hands_chunks_to_the_exact_path():
    """A chunk the one-pass form of the ring cannot finish by itself - a document for the host normalizer, a document of more than 512 segments,
    text that capcode more than doubles (beyond the bound the segment kernels were launched over) - costs nothing behind its normalizer pass
    and is run through the exact path by the ring's finisher; the ids land in document order all the same."""
    from conftest import EMULATED
    img = synth.synth_vocab(synth.ENGLISHCODE, 3000, capcode=2, norm_flag=1, level=3, seed=0x52494E47)
    v = tm.Vocab(img)
    rng = np.random.default_rng(11)
    base_raw, base_off = synth.synth_corpus(synth.ENGLISHCODE, 150_000, seed=74)
    docs = [bytes(base_raw[int(base_off[d]):int(base_off[d + 1])]) for d in range(base_off.size - 1)]
    n0 = len(docs)
    docs.insert(n0 // 5, "𐐀𐐨 Deseret has case beyond the BMP: the host normalizer's 𐐁𐐩 ".encode() * 20)                              # a cased script of plane 1: host normalizer
    docs.insert(n0 // 2, b"one long document of plain words that goes on and on " * 3200)                               # 170 KB: more than 512 segments
    docs.insert(4 * n0 // 5, b".".join(bytes([65 + int(c)]) for c in rng.integers(0, 26, 9_000)))                        # grows 2.5 x under capcode
    docs.append(b"")
    raw = np.frombuffer(b"".join(docs), dtype=np.uint8).copy()
    roffs = np.zeros(len(docs) + 1, dtype=np.uint64)
    roffs[1:] = np.cumsum([len(x) for x in docs])
    text, offs = synth.normalize_batch(raw, roffs, 2, 1)
    ids, toff, miss = v.tokenize_packed(text, offs)
    pin, pout = _pinned_copy(raw), tm.PinnedBuffer(2 * ids.size + 64)
    for chunk in (20_000, 60_000):
        blob, boff, bmiss, enc, st = v.tokenize_pipeline(pin.array[: raw.size], roffs, raw=True, chunk_bytes=chunk, lanes=3, out=pout.array)
        assert st["ring"] == 1 and 3 <= st["ring_exact_chunks"] < st["chunks"], st
        assert st["host_fallback_docs"] == 1
        assert (boff == toff * np.uint64(enc)).all() and (bmiss == miss).all()
        assert (_ids_from_bytes(np.asarray(blob), enc) == ids).all()
        assert st["normalized_bytes"] == text.size


@pytest.mark.parametrize("capcode", [2, 0])
def test_resident_decode_equals_the_host_buffer_decode(capcode):
    """tm_batch_decode: the ids a batch holds decoded where they lie (ids in HBM -> text in HBM, what bench.py --workload decode times) give
    the text tm_decode_batch gives for the same ids - device-decoded documents and those left to the host decoder alike - and, for a
    vocabulary with every byte as a token, the NFD form of the raw text back."""
    from conftest import EMULATED
    img = synth.synth_vocab(synth.ENGLISHCODE, 3000, capcode=capcode, norm_flag=1, level=3, seed=0x44454344)
    v = tm.Vocab(img)
    base_raw, base_off = synth.synth_corpus(synth.ENGLISHCODE, 60_000 if EMULATED else 1_500_000, seed=75)
    docs = [bytes(base_raw[int(base_off[d]):int(base_off[d + 1])]) for d in range(base_off.size - 1)]
    docs += ["Découvert À Paris, le CŒUR de l\u2019été: Garçon ÉTÉ".encode(), "中文 と Ελληνικά ЖУК жук".encode(), b"", "I\u0131 \uff21\uff41 mixed CASE".encode(), b"ALL CAPS WORDS HERE and Title Case"]
    raw = np.frombuffer(b"".join(docs), dtype=np.uint8).copy()
    roffs = np.zeros(len(docs) + 1, dtype=np.uint64)
    roffs[1:] = np.cumsum([len(x) for x in docs])
    text, offs = synth.normalize_batch(raw, roffs, capcode, 1)
    ids, toff, _ = v.tokenize_packed(text, offs)
    for raw_mode in (False, True):
        exp, eoff = v.decode_packed(ids, toff, raw=raw_mode)
        got, goff, host_docs = v.roundtrip_resident(raw, roffs, raw=raw_mode)
        assert (goff == eoff).all() and got.tobytes() == exp.tobytes()
        if capcode == 2 and not raw_mode:
            assert 1 <= host_docs <= 3                      # (the dotless i / fullwidth letters: upper-case forms the device leaves to the host)
        else:
            assert host_docs == 0
    plain, poff = synth.normalize_batch(raw, roffs, 0, 1)     # NFD without capcode: what Decode gives back (no token is missing in this vocabulary's text)
    got, goff, _ = v.roundtrip_resident(raw, roffs)
    if int(v.tokenize_packed(text, offs)[2].sum()) == 0:
        assert (goff == poff).all() and got.tobytes() == plain.tobytes()
