"""CPU, world_size 2 over gloo: the N>1 paths — document sharding (no collective) and the scoring histogram
all-reduce — with the oracle standing in for the per-rank device pass."""
import os
import socket
import sys

import numpy as np
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


def _worker(rank, world, port, img, data, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_bind import Oracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    orc = Oracle(img)
    lo, hi = tmdist.shard_strips(len(data), rank, world)
    scores, tit, ms = orc.score(data[lo:hi])
    words = torch.from_numpy(tmdist.encode_histogram(scores, tit, ms).view(np.int32).copy())
    tmdist.allreduce_histogram(words)
    np.save(os.path.join(out_dir, "hist%d.npy" % rank), words.numpy())
    dist.destroy_process_group()


def test_histogram_allreduce_world2(tmp_path):
    rng = np.random.default_rng(21)
    toks = fuzz_vocab_tokens(rng, 2, 120)
    img = synth.build_vocab(toks, capcode=2, charset=1)
    data = fuzz_text(rng, 2, 60000)
    port = _free_port()
    mp.spawn(_worker, args=(2, port, img, data, str(tmp_path)), nprocs=2, join=True)
    from oracle_bind import Oracle
    orc = Oracle(img)
    h0, h1 = np.load(tmp_path / "hist0.npy"), np.load(tmp_path / "hist1.npy")
    assert (h0 == h1).all()
    n_ids = orc.n_ids()
    scores, tokens, missing = tmdist.decode_histogram(h0, n_ids)
    # expected: the two strips scored independently and summed (what trainvocab workers do with strips)
    lo0, hi0 = tmdist.shard_strips(len(data), 0, 2)
    lo1, hi1 = tmdist.shard_strips(len(data), 1, 2)
    assert lo0 == 0 and hi0 == lo1 and hi1 == len(data)
    s0, t0, m0 = orc.score(data[lo0:hi0])
    s1, t1, m1 = orc.score(data[lo1:hi1])
    assert (scores == s0 + s1).all() and tokens == t0 + t1 and (missing == (m0 | m1)).all()


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
