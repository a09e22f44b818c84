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
    under its kernels.  Candidate A is scored asynchronously into a caller-owned device buffer and freed at once, candidate B (other
    tokens, same size: it takes A's block) is loaded and scored before anybody has synchronized; A's histogram must be A's."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    from conftest import EMULATED
    if EMULATED:
        pytest.skip("the emulated device completes every launch at once: nothing can be in flight")
    from tokenmonster_amd import _native as N, dist as tmdist
    img_a = synth.synth_vocab(synth.ENGLISHCODE, 9000, capcode=2, norm_flag=1, level=5, seed=0x41414141)
    img_b = synth.synth_vocab(synth.ENGLISHCODE, 9000, capcode=2, norm_flag=1, level=5, seed=0x42424242)
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 48 << 20, seed=72)
    text, _ = synth.normalize_batch(raw, roffs, 2, 1)
    ds = C.c_void_p()
    N.check(N.lib.tm_dataset_upload(N.ptr(np.ascontiguousarray(text)), int(text.size), C.byref(ds)))
    try:
        stream = torch.cuda.Stream()
        for trial in range(3):
            a, b = tm.Vocab(img_a), tm.Vocab(img_b)
            wa, wb = a.n_ids() + 260, b.n_ids() + 260
            b.close()                                       # a parked block of B's size is waiting
            hist_a = torch.zeros(wa, dtype=torch.int32, device="cuda")
            N.check(N.lib.tm_score_device_into(a.handle, ds, None, None, 0, C.c_void_p(stream.cuda_stream), C.c_void_p(hist_a.data_ptr()), wa))
            a.close()                                       # kernels of the pass may still be running
            b2 = tm.Vocab(img_b)                            # takes a parked block - A's, if nothing protects it
            hist_b = torch.zeros(wb, dtype=torch.int32, device="cuda")
            N.check(N.lib.tm_score_device_into(b2.handle, ds, None, None, 0, C.c_void_p(stream.cuda_stream), C.c_void_p(hist_b.data_ptr()), wb))
            torch.cuda.synchronize()
            if trial == 0:
                orc_a, orc_b = Oracle(img_a), Oracle(img_b)
                n = 2 << 20                                  # the oracle walks 15 MB/s: pin the head, compare the whole with a second, synchronous pass
                exp_a = orc_a.score(text[:n])
                so, sl = np.array([0], dtype=np.uint64), np.array([n], dtype=np.uint64)
                ref_a = tm.Vocab(img_a)
                got = np.zeros(ref_a.n_ids(), dtype=np.uint32)
                tit = C.c_uint64()
                ms = np.zeros(32, dtype=np.uint8)
                N.check(N.lib.tm_score(ref_a.handle, ds, N.ptr(so), N.ptr(sl), 1, N.ptr(got), C.byref(tit), N.ptr(ms)))
                assert (got == exp_a[0]).all() and tit.value == exp_a[1]
                full_a = np.zeros(ref_a.n_ids(), dtype=np.uint32)
                N.check(N.lib.tm_score(ref_a.handle, ds, None, None, 0, N.ptr(full_a), C.byref(tit), N.ptr(ms)))
                full_a_tokens = tit.value
            sa, ta, _ = tmdist.decode_histogram(hist_a.cpu().numpy(), wa - 260)
            assert ta == full_a_tokens and (sa == full_a).all(), "trial %d: the histogram of the freed vocabulary is not its own" % trial
    finally:
        N.lib.tm_dataset_free(ds)
