"""(named to run LAST under `pytest -x`: it is the one -m gpu test that starts further processes, each importing torch.)
-m gpu: the scoring pass of N ranks as N PROCESSES, each with its own handle on the device, talking over torch.distributed —
what bench.py --workload score --gpus N does, with gloo standing in for RCCL so that it runs on a one-GPU box (the ranks share
device 0) and, on the emulated device (TM_EMU=1, tools/emu), on none.  Every rank is given its own byte range only; it fetches the
halo from its neighbour (dist.exchange_halo), runs tm_score_begin, all-gathers the 80 exit states, finishes from its true entry
state (dist.score_ranges_exact with the HipRange engine) and all-reduces the histogram words.  Only rank 0 builds the vocabulary's
tables: the others receive its device block (dist.broadcast_vocab).  The result must equal the oracle's
ONE walk over the whole buffer (training/trainvocab.go:909-922)."""
import os
import socket
import sys

import numpy as np
import pytest

import conftest  # noqa: F401  (first: under TM_EMU=1 it redirects the library before tokenmonster_amd is imported, also in the ranks)
from conftest import fuzz_text, fuzz_vocab_tokens

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, img, data, out_dir):
    import ctypes as C
    import torch
    import torch.distributed as dist
    import tokenmonster_amd as tm
    from tokenmonster_amd import _native as N
    from tokenmonster_amd import dist as tmdist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # the box's hostname may not resolve: loopback, explicitly
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=180))
    try:
        N.check(N.lib.tm_set_device(0))
        # ONE rank turns the candidate into tables; the others take the finished device block (tm_vocab_block_export / _import, one broadcast)
        v = tmdist.broadcast_vocab(tm.Vocab(img) if rank == 0 else None, 0, rank, device=0, on_device=not conftest.EMULATED)
        lo, hi = tmdist.shard_strips(len(data), rank, world)
        own = np.frombuffer(data[lo:hi], dtype=np.uint8)
        halo = tmdist.exchange_halo(own, rank, world)
        buf = np.ascontiguousarray(np.concatenate([own, halo]))
        ds = C.c_void_p()
        N.check(N.lib.tm_dataset_upload(N.ptr(buf), int(buf.size), C.byref(ds)))
        engine = tmdist.HipRange(v, ds, own.size, continues=rank + 1 < world)
        for _ in range(2):                                   # twice: a pass must leave nothing behind that changes the next one
            tmdist.score_ranges_exact(engine, rank, world)
            n_ids = v.n_ids()
            s = np.zeros(n_ids, dtype=np.uint32)
            t = C.c_uint64()
            m = np.zeros(32, dtype=np.uint8)
            N.check(N.lib.tm_score_read(v.handle, ds, N.ptr(s), C.byref(t), N.ptr(m)))
            words = torch.from_numpy(tmdist.encode_histogram(s, t.value, m).view(np.int32).copy())
            tmdist.allreduce_histogram(words)
        np.save(os.path.join(out_dir, "hist%d.npy" % rank), words.numpy())
        N.lib.tm_dataset_free(ds)
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3])
def test_rank_processes_score_one_whole_buffer_walk(tmp_path, world):
    import torch.multiprocessing as mp
    from tokenmonster_amd import dist as tmdist
    from tokenmonster_amd import synth
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from oracle_bind import Oracle
    rng = np.random.default_rng(400 + world)
    img = synth.build_vocab(fuzz_vocab_tokens(rng, 2, 160), capcode=2, charset=1)
    data = fuzz_text(rng, 2, 200_003)              # ~780 segments per rank at world 2: the ranges resolve through the group tree
    mp.spawn(_rank, args=(world, _free_port(), img, data, str(tmp_path)), nprocs=world, join=True)
    orc = Oracle(img)
    hs = [np.load(tmp_path / ("hist%d.npy" % r)) for r in range(world)]
    for h in hs[1:]:
        assert (h == hs[0]).all()
    scores, tokens, missing = tmdist.decode_histogram(hs[0], orc.n_ids())
    exp_s, exp_t, exp_m = orc.score(data)
    assert (scores == exp_s).all() and tokens == exp_t and (missing == exp_m).all()
