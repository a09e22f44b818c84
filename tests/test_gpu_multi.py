"""The in-library multi-device driver (include/tokenmonster_hip.h "several devices", tm_multi.hip): what a ONE-process host — the Go
library, tokenmonsterserver (training/tokenmonsterserver.go:363-378), trainvocab's workers (training/trainvocab.go:1827-1829, :909-922)
— calls for N > 1 GPUs.  A gpurun box has one GPU, so N members of a tm_devices handle sit on device 0 ("virtual devices"):
every code path but the RCCL call itself runs — replication of the vocabulary block, the chunk queue over several devices' lanes, ranges
+ halos, the exit-map chain, the reduction (peer copy + add instead of ncclAllReduce: RCCL refuses two ranks on one device) — and RCCL
is exercised on a communicator of hipGetDeviceCount() ranks in its own test."""
import os

import numpy as np
import pytest

import tokenmonster_amd as tm
from tokenmonster_amd import multi, synth
from conftest import EMULATED, fuzz_text, fuzz_vocab_tokens
from oracle_bind import Oracle

pytestmark = pytest.mark.gpu


def _micro(seed, nbytes):
    rng = np.random.default_rng(seed)
    img = synth.build_vocab(fuzz_vocab_tokens(rng, 2, 150), capcode=2, charset=1)
    return img, np.frombuffer(fuzz_text(rng, 2, nbytes), dtype=np.uint8)


@pytest.mark.parametrize("members", [1, 2, 3, 8])
def test_score_multi_equals_the_whole_buffer_walk(members):
    img, data = _micro(500 + members, 333_337)
    orc = Oracle(img)
    exp_s, exp_t, exp_m = orc.score(data)
    g = multi.Devices([0] * members)
    try:
        assert len(g) == members and g.device(members - 1) == 0
        vs = multi.VocabSet(g, img)
        ds = multi.DatasetSet(g, data)
        rr = ds.ranges()
        assert sum(r for r, _ in rr) == data.size and all(h == 128 for _, h in rr[:-1]) and rr[-1][1] == 0
        for _ in range(2):                        # a second pass reuses every workspace
            got_s, got_t, got_m = ds.score(vs)
            assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
        # ... and equals the single-device entry point on the same bytes
        v = tm.Vocab(img)
        from test_gpu_parity import _score
        one_s, one_t, one_m = _score(v, data)
        assert (one_s == got_s).all() and one_t == got_t and (one_m == got_m).all()
        n, why = g.rccl_ranks()
        assert (n == 0 and why) if members > 1 or EMULATED else True      # members on one device: no communicator, and the handle says why
        ds.close(); vs.close()
    finally:
        g.close()


def test_score_multi_tiny_and_empty_datasets():
    img, data = _micro(77, 9_000)
    orc = Oracle(img)
    g = multi.Devices([0, 0, 0, 0])
    try:
        vs = multi.VocabSet(g, img)
        for n in (0, 1, 63, 4095, 4096, 8191, 9_000):          # fewer than 4 KiB per member: fewer members take part, the rest run the identity
            ds = multi.DatasetSet(g, data[:n])
            exp_s, exp_t, exp_m = orc.score(data[:n])
            got_s, got_t, got_m = ds.score(vs)
            assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all(), n
            ds.close()
        vs.close()
    finally:
        g.close()


def test_score_multi_candidate_shape():
    """BASELINE.json configs[4]'s vocabulary shape (65 536 candidate ids) over 8 members: long ranges (group maps carry the exit map)"""
    if EMULATED:
        pytest.skip("8 MiB through the emulated device takes minutes")
    img = synth.config_vocab("candidates-65536")
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 8 << 20, seed=0x434F5250 + 5)
    data, _ = synth.normalize_batch(raw, offs, 2, 1)
    exp_s, exp_t, exp_m = Oracle(img).score(data)
    g = multi.Devices([0] * 8)
    try:
        vs, ds = multi.VocabSet(g, img), multi.DatasetSet(g, data)
        got_s, got_t, got_m = ds.score(vs)
        assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
        ds.close(); vs.close()
    finally:
        g.close()


@pytest.mark.parametrize("members,raw", [(2, True), (3, False)])
def test_pipeline_multi_equals_one_device(members, raw):
    img = synth.synth_vocab(synth.ENGLISHCODE, 4096, capcode=2, norm_flag=1, level=3, seed=0x534D4F4B)
    rawtext, offs = synth.synth_corpus(synth.ENGLISHCODE, 3_000_000 if not EMULATED else 300_000, seed=12)
    if not raw:
        rawtext, offs = synth.normalize_batch(rawtext, offs, 2, 1)
    v = tm.Vocab(img)
    chunk = (256 << 10) if not EMULATED else (32 << 10)
    one = v.tokenize_pipeline(rawtext, offs, raw=raw, chunk_bytes=chunk, lanes=2)
    g = multi.Devices([0] * members)
    try:
        vs = multi.VocabSet(g, img)
        got = vs.tokenize_pipeline(rawtext, offs, raw=raw, chunk_bytes=chunk, lanes_per_device=2)
        assert got[3] == one[3] and (got[1] == one[1]).all() and (got[0] == one[0]).all() and (got[2] == one[2]).all()
        assert got[4]["lanes"] == min(2 * members, got[4]["chunks"]) and got[4]["chunks"] >= 4
        # the replicas are vocabularies of their own: every one tokenizes by itself
        import ctypes as C
        from tokenmonster_amd import _native as N
        text, noff = (rawtext, offs) if not raw else synth.normalize_batch(rawtext, offs, 2, 1)
        nd = min(64, noff.size - 1)
        sub = np.ascontiguousarray(noff[:nd + 1])
        ref_ids, ref_off, _ = v.tokenize_packed(text[:int(sub[nd])], sub)
        for i in range(members):
            out = np.empty(ref_ids.size + 16, dtype=np.uint32)
            toff = np.zeros(nd + 1, dtype=np.uint64)
            miss = np.zeros(nd, dtype=np.uint32)
            N.check(N.lib.tm_tokenize_batch(vs.member(i), N.ptr(np.ascontiguousarray(text[:int(sub[nd])])), N.ptr(sub), nd, N.ptr(out), out.size, N.ptr(toff), N.ptr(miss)))
            assert (toff == ref_off).all() and (out[:ref_ids.size] == ref_ids).all()
        vs.close()
    finally:
        g.close()


@pytest.mark.parametrize("members", [2, 3, 5])
def test_a_member_that_fails_between_the_meetings_brings_everybody_home(members):
    """round-4 advice: a failure raised right after the first meeting of the members' threads must not change what a slower member is
    told AT that meeting (it would leave early, and the failing member would wait for it at the second meeting forever).  Test hook 14
    makes the last member give up there; the call has to return its error - under a watchdog, because the bug is a hang."""
    import threading
    from tokenmonster_amd import _native as N
    img, data = _micro(321 + members, 120_000)
    g = multi.Devices([0] * members)
    try:
        vs, ds = multi.VocabSet(g, img), multi.DatasetSet(g, data)
        good = ds.score(vs)
        old = N.lib.tm_debug_flags(16384)
        box = {}
        try:
            for _ in range(20):                                      # the interleaving is a race: many tries
                def call():
                    try:
                        ds.score(vs)
                        box["r"] = "returned ok"
                    except N.TokenMonsterHipError as ex:
                        box["r"] = str(ex)
                t = threading.Thread(target=call, daemon=True)
                t.start()
                t.join(60)
                assert not t.is_alive(), "tm_score_multi hangs when a member fails after the first meeting"
                assert "test hook 14" in box["r"], box["r"]
        finally:
            N.lib.tm_debug_flags(old)
        again = ds.score(vs)                                         # ... and the handle is as good as before
        assert (again[0] == good[0]).all() and again[1] == good[1]
        ds.close(); vs.close()
    finally:
        g.close()


def test_devices_handle_rules():
    from tokenmonster_amd import _native as N
    with pytest.raises(N.TokenMonsterHipError):
        multi.Devices([0, 99])
    g = multi.Devices(1)
    assert len(g) == 1 and g.device(0) == 0 and g.device(5) == -1
    g.close()
    os.environ["TM_VIRTUAL_DEVICES"] = "3"
    try:
        g = multi.Devices()
        assert len(g) == 3 and all(g.device(i) == 0 for i in range(3))
        g.close()
    finally:
        del os.environ["TM_VIRTUAL_DEVICES"]


def test_rccl_allreduce_on_the_visible_devices():
    """the RCCL leg proper: a communicator over hipGetDeviceCount() devices (one on a gpurun box; TM_RCCL=1 makes a one-member handle run
    the collective anyway), ncclAllReduce(sum, uint32) of the histogram inside tm_score_multi"""
    if EMULATED:
        pytest.skip("no RCCL on the emulated device")
    from tokenmonster_amd import _native as N
    img, data = _micro(611, 200_000)
    exp_s, exp_t, exp_m = Oracle(img).score(data)
    os.environ["TM_RCCL"] = "1"
    try:
        g = multi.Devices()                        # every visible device
        n, why = g.rccl_ranks()
        assert n == len(g) == N.lib.tm_device_count(), why
        vs, ds = multi.VocabSet(g, img), multi.DatasetSet(g, data)
        got_s, got_t, got_m = ds.score(vs)
        assert (got_s == exp_s).all() and got_t == exp_t and (got_m == exp_m).all()
        ds.close(); vs.close(); g.close()
    finally:
        del os.environ["TM_RCCL"]


def test_vocab_set_tune_keeps_every_result():
    """tm_vocab_set_tune: member 0 lays its tables out by use, the other members take over the block and the scalars that move with it
    (node values): the sharded scoring pass and the multi-device pipeline give what they gave before"""
    img, data = _micro(901, 150_000)
    g = multi.Devices([0, 0, 0])
    try:
        vs = multi.VocabSet(g, img)
        ds = multi.DatasetSet(g, data)
        before = ds.score(vs)
        raw = bytes(data[:60_000])
        docs = [raw[i:i + 3000] for i in range(0, len(raw), 3000)]
        text, offs = tm.pack_documents(docs)
        b0 = vs.tokenize_pipeline(text, offs, raw=False, chunk_bytes=20_000)
        for sample in (data[:50_000], data[50_000:], data[:0]):
            vs.tune(sample)
            after = ds.score(vs)
            assert (after[0] == before[0]).all() and after[1] == before[1] and (after[2] == before[2]).all()
            b1 = vs.tokenize_pipeline(text, offs, raw=False, chunk_bytes=20_000)
            assert (b1[0] == b0[0]).all() and (b1[1] == b0[1]).all()
        ds.close(); vs.close()
    finally:
        g.close()
