"""CPU, world_size 2 and 3 over gloo: the N>1 paths — document sharding (no collective) and the scoring pass as byte ranges of ONE
whole-buffer walk (exit-map exchange + histogram all-reduce, tokenmonster_amd/dist.py) — with the oracle standing in for the
per-rank device pass.  The assertion is the one SURVEY.md 8(e) and the north star ask for: the summed histogram equals the
oracle's walk over the WHOLE buffer (training/trainvocab.go:909-922), not a sum of independent strips."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import fuzz_text, fuzz_vocab_tokens
from tokenmonster_amd import dist as tmdist
from tokenmonster_amd import synth


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class OracleRange:
    """engine for dist.score_ranges_exact backed by the oracle: the rank holds its own bytes + the halo of the next rank's, exactly
    what a GPU rank uploads.  begin() walks the range once per entry state (the device gets all 80 from one pass of its match
    kernel); entry states whose forward-delete look-up cannot succeed are reported unreachable like the device does."""

    def __init__(self, orc, own_plus_halo, own_len):
        self.orc, self.buf, self.own_len = orc, own_plus_halo, own_len

    def begin(self):
        ex = np.full(tmdist.ENTRY_STATES, tmdist.UNREACHABLE, dtype=np.uint8)
        for e in range(tmdist.ENTRY_STATES):
            if (e >> 1) > self.own_len:
                continue
            if e & 1:          # (offset, forwardDelete = 1) exists only where ' ' + text has a match (go :1088-1095)
                i = e >> 1
                if not self.orc.longest(b" " + bytes(self.buf[i:i + 39]))[2]:
                    continue
            ex[e] = self.orc.score_range(self.buf, 0, self.own_len, e)[3]
        return ex

    def finish(self, entry):
        s, t, m, _ = self.orc.score_range(self.buf, 0, self.own_len, entry)
        return torch.from_numpy(tmdist.encode_histogram(s, t, m).view(np.int32).copy())


def _worker(rank, world, port, img, data, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_bind import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle(img)
    lo, hi = tmdist.shard_strips(len(data), rank, world)
    own = np.frombuffer(data[lo:hi], dtype=np.uint8)                       # what this rank was given: ITS range only
    halo = tmdist.exchange_halo(own, rank, world)                          # ... plus the first bytes of the next rank's, over the wire
    assert halo.tobytes() == (data[hi:hi + tmdist.HALO] if rank + 1 < world else b"")
    engine = OracleRange(orc, np.concatenate([own, halo]), own.size)
    words = tmdist.score_ranges_exact(engine, rank, world)
    tmdist.allreduce_histogram(words)
    np.save(os.path.join(out_dir, "hist%d.npy" % rank), words.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ranges_of_one_walk_equal_the_whole_buffer(tmp_path, world):
    rng = np.random.default_rng(21 + world)
    toks = fuzz_vocab_tokens(rng, 2, 140)
    img = synth.build_vocab(toks, capcode=2, charset=1)
    data = fuzz_text(rng, 2, 30011)          # range boundaries fall inside tokens and inside forward-delete pairs
    port = _free_port()
    mp.spawn(_worker, args=(world, port, img, data, str(tmp_path)), nprocs=world, join=True)
    from oracle_bind import Oracle
    orc = Oracle(img)
    hs = [np.load(tmp_path / ("hist%d.npy" % r)) for r in range(world)]
    for h in hs[1:]:
        assert (h == hs[0]).all()
    scores, tokens, missing = tmdist.decode_histogram(hs[0], orc.n_ids())
    exp_s, exp_t, exp_m = orc.score(data)                                  # ONE walk over the whole buffer
    assert (scores == exp_s).all() and tokens == exp_t and (missing == exp_m).all()


def test_range_walk_chains_to_the_whole_walk():
    """the oracle's own range form: any cut points, chained through the exit states, give the whole walk"""
    from oracle_bind import Oracle
    rng = np.random.default_rng(5)
    toks = fuzz_vocab_tokens(rng, 2, 150)
    img = synth.build_vocab(toks, capcode=2, charset=1, with_unk=True)
    orc = Oracle(img)
    data = fuzz_text(rng, 2, 20000)
    exp_s, exp_t, exp_m = orc.score(data)
    inside = 0          # cuts that fell inside a token (entry state != 0): the case independent strips get wrong
    rnd = [0] + [c for c in np.cumsum(rng.integers(64, 600, size=60)).tolist() if c < 19900] + [20000]
    for cuts in ([0, 20000], [0, 777, 20000], [0, 64, 128, 4097, 19936, 20000], rnd):
        s = np.zeros_like(exp_s)
        t, m, e = 0, np.zeros(32, dtype=np.uint8), 0
        for a, b in zip(cuts, cuts[1:]):
            s_, t_, m_, e = orc.score_range(data, a, b, e)
            inside += e != 0
            s += s_
            t += t_
            m |= m_
        assert (s == exp_s).all() and t == exp_t and (m == exp_m).all() and e == 0, cuts
    assert inside > 5


def test_sharded_documents_concatenate_to_the_single_rank_ids():
    """batch tokenize needs no collective: ranks take contiguous document ranges (shard_documents) and the ids of the ranks,
    concatenated in rank order, are the ids of one rank tokenizing everything (documents never interact,
    training/tokenmonsterserver.go:371)"""
    from oracle_bind import Oracle
    img = synth.synth_vocab(synth.ENGLISHCODE, 1500, capcode=2, norm_flag=1, level=3, seed=77)
    orc = Oracle(img)
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 120_000, seed=4)
    text, offs = synth.normalize_batch(raw, roffs, 2, 1)
    nd = offs.size - 1
    whole = [orc.tokenize(text[int(offs[d]):int(offs[d + 1])])[0] for d in range(nd)]
    for world in (2, 3, 8):
        got = []
        for r in range(world):
            d0, d1 = tmdist.shard_documents(offs, r, world)
            sub = offs[d0:d1 + 1] - offs[d0]
            part = text[int(offs[d0]):int(offs[d1])]
            got += [orc.tokenize(part[int(sub[k]):int(sub[k + 1])])[0] for k in range(d1 - d0)]
        assert len(got) == nd and all((a == b).all() for a, b in zip(got, whole))


def test_shard_documents_balanced_and_complete():
    rng = np.random.default_rng(2)
    lens = rng.integers(0, 5000, size=1000)
    offsets = np.zeros(1001, dtype=np.uint64)
    np.cumsum(lens, out=offsets[1:])
    for world in (1, 2, 3, 8):
        parts = [tmdist.shard_documents(offsets, r, world) for r in range(world)]
        assert parts[0][0] == 0 and parts[-1][1] == 1000
        for a, b in zip(parts, parts[1:]):
            assert a[1] == b[0]
        sizes = [int(offsets[b] - offsets[a]) for a, b in parts]
        assert max(sizes) - min(sizes) <= 2 * 5000
    assert tmdist.shard_documents(np.zeros(1, dtype=np.uint64), 0, 2) == (0, 0)


def test_histogram_codec_roundtrip():
    scores = np.arange(1000, dtype=np.uint32) * 4000000
    ms = np.zeros(32, dtype=np.uint8)
    ms[3] = 0x81
    w = tmdist.encode_histogram(scores, (1 << 40) + 12345, ms)
    s, t, m = tmdist.decode_histogram(w, 1000)
    assert (s == scores).all() and t == (1 << 40) + 12345 and (m == ms).all()
