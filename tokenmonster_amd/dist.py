"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" = RCCL over xGMI).

Batch tokenize shards by documents and needs NO collective.  The trainvocab scoring pass shards the
dataset into byte ranges and merges the per-rank histograms with ONE all-reduce(sum) of
n_ids + 4 + 256 uint32 words (include/tokenmonster_hip.h: tm_score_device)."""
import numpy as np


def shard_documents(offsets, rank, world):
    """contiguous range of documents [d0, d1) for `rank`, balanced by BYTES (not by document count)"""
    offsets = np.asarray(offsets, dtype=np.uint64)
    nd = offsets.size - 1
    total = int(offsets[nd]) if nd > 0 else 0
    cuts = [int(np.searchsorted(offsets, np.uint64(total * r // world), side="left")) for r in range(world + 1)]
    cuts[0], cuts[world] = 0, nd
    for r in range(1, world + 1):
        cuts[r] = max(cuts[r], cuts[r - 1])
    return cuts[rank], cuts[rank + 1]


ENTRY_STATES = 80        # 40 offsets x forwardDelete {0, 1}: tmh::ENT
HALO = 128               # bytes of the following text a rank must see: longest token (40) + look-ahead (41), rounded up
UNREACHABLE = 0xFF


def shard_strips(n_bytes, rank, world, align=4):
    """byte range of one contiguous dataset for `rank` (trainvocab cuts strips on multiples of 4, trainvocab.go:1674).
    With score_ranges_exact below the ranges are pieces of ONE whole-buffer walk (exact); walked as independent strips
    they are the approximation the reference makes at its strip boundaries before "midway"."""
    per = (n_bytes // world) // align * align
    lo = per * rank
    hi = n_bytes if rank == world - 1 else per * (rank + 1)
    return lo, hi


def allreduce_histogram(words, group=None):
    """in-place all-reduce(sum) of a histogram tensor of uint32 words viewed as int32 (two's-complement addition is
    the same bit pattern as unsigned addition).  `words`: torch int32 tensor on the rank's device."""
    import torch.distributed as dist
    dist.all_reduce(words, op=dist.ReduceOp.SUM, group=group)
    return words


def decode_histogram(words, n_ids):
    """uint32 words -> (scores u32[n_ids], tokens_in_text, missing_set[32]); inverse of k_hist_finish's layout"""
    w = np.asarray(words).view(np.uint32)
    scores = w[:n_ids].copy()
    limbs = w[n_ids:n_ids + 4].astype(np.uint64)
    tokens = int(limbs[0] + (limbs[1] << np.uint64(16)) + (limbs[2] << np.uint64(32)) + (limbs[3] << np.uint64(48)))
    missing = np.zeros(32, dtype=np.uint8)
    for k in np.nonzero(w[n_ids + 4:n_ids + 260])[0]:
        missing[k >> 3] |= np.uint8(1 << (k & 7))
    return scores, tokens, missing


def encode_histogram(scores, tokens, missing_set):
    """host-side construction of the same layout (used by tests and by CPU ranks)"""
    n_ids = len(scores)
    w = np.zeros(n_ids + 260, dtype=np.uint32)
    w[:n_ids] = scores
    for k in range(4):
        w[n_ids + k] = (tokens >> (16 * k)) & 0xFFFF
    for b in range(256):
        if missing_set[b >> 3] & (1 << (b & 7)):
            w[n_ids + 4 + b] = 1
    return w


# ---- exact data-parallel scoring: ONE whole-buffer walk (training/trainvocab.go:909-922) cut into one byte range per rank ------------
# The walk's whole state at a token boundary is (position, forwardDelete): at a range boundary that is one of 80 entry states
# (2 * offset-into-the-next-range + forwardDelete, offset < 40).  What a range does to the state is a map of 80 entries that the match
# kernel produces anyway (exit maps, composed per range by k_group_compose / k_doc_exits).  Protocol, per scoring pass:
#   1. every rank runs the match kernel over its range (which looks HALO bytes into the next rank's text) -> exits[80]
#   2. all-gather 80 bytes per rank; rank r chains exits[0][0] -> exits[1][.] -> ... -> its own entry state
#   3. every rank finishes its pass from that state; ONE all-reduce(sum) of the histogram words
# Engines: HipRange (tm_score_begin / tm_score_finish on this rank's GPU); the CPU tests use an oracle-backed engine with the same
# two methods.


def exchange_halo(own, rank, world, group=None, backend_device="cpu"):
    """the first HALO bytes of rank+1's range (empty for the last rank).  `own`: uint8 numpy array, every rank >= HALO bytes."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return np.zeros(0, dtype=np.uint8)
    if own.size < HALO:
        raise ValueError("every rank's range must hold at least %d bytes" % HALO)
    mine = torch.from_numpy(np.ascontiguousarray(own[:HALO]).copy()).to(backend_device)
    heads = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(heads, mine, group=group)
    return heads[rank + 1].cpu().numpy() if rank + 1 < world else np.zeros(0, dtype=np.uint8)


def resolve_entry(all_exits, rank):
    """all_exits[r][e] = exit state of rank r's range when entered in state e -> entry state of `rank` (rank 0 enters in 0)"""
    e = 0
    for r in range(rank):
        e = int(all_exits[r][e])
        if e == UNREACHABLE or e >= ENTRY_STATES:
            raise RuntimeError("range of rank %d cannot be entered in the state the walk reaches it in" % r)
    return e


def score_ranges_exact(engine, rank, world, group=None, backend_device="cpu"):
    """one scoring pass over this rank's range as a piece of the whole-buffer walk; returns what engine.finish returns (for HipRange the
    device histogram is already all-reduced in place by the caller's choice: see bench.py).  engine.begin() -> 80 exit states;
    engine.finish(entry_state) -> result."""
    import torch
    import torch.distributed as dist
    exits = np.asarray(engine.begin(), dtype=np.uint8)
    assert exits.size == ENTRY_STATES
    if world > 1:
        mine = torch.from_numpy(exits.copy()).to(backend_device)
        allx = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allx, mine, group=group)
        entry = resolve_entry([t.cpu().numpy() for t in allx], rank)
    else:
        entry = 0
    return engine.finish(entry)


class HipRange:
    """engine for score_ranges_exact on this rank's GPU.  `dataset` holds the rank's bytes followed by the halo; the range is
    [0, own_len) of it."""

    def __init__(self, vocab, dataset, own_len, continues, stream=None, dst=None, dst_words=0, text_ends_in_halo=False):
        """continues: text follows the range (the dataset then holds >= HALO bytes of it; text_ends_in_halo: or fewer, and they are all there is)"""
        self.vocab, self.ds, self.own_len, self.continues = vocab, dataset, int(own_len), (2 if text_ends_in_halo else 1) if continues else 0
        self.stream, self.dst, self.dst_words = stream, dst, dst_words

    def begin(self):
        import ctypes as C
        from . import _native as N
        ex = np.zeros(ENTRY_STATES, dtype=np.uint8)
        N.check(N.lib.tm_score_begin(self.vocab.handle, self.ds, 0, self.own_len, self.continues, C.c_void_p(self.stream or 0), N.ptr(ex)))
        return ex

    def finish(self, entry):
        import ctypes as C
        from . import _native as N
        N.check(N.lib.tm_score_finish(self.vocab.handle, self.ds, int(entry), C.c_void_p(self.stream or 0), C.c_void_p(self.dst or 0), self.dst_words))
        return entry


# ---- one rank builds a candidate's tables, the others take the finished device block -----------------------------------------------
# In the data-parallel scoring mode every rank scores its range against the SAME candidate; tm_build_vocab + tm_vocab_load cost ~50 ms
# of one host thread, ten times the pass they feed at 8 GPUs.  So the candidates are built round-robin (rank r prepares candidates
# r, r + N, ... ahead of time on its host threads) and the finished block - a few MB - goes to the others in one broadcast.

class _DeviceBytes:
    """`nbytes` bytes at a raw device pointer, as something torch.as_tensor can alias without a copy"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def _alias(ptr, nbytes, on_device):
    import torch
    if on_device:
        return torch.as_tensor(_DeviceBytes(ptr, nbytes), device="cuda")
    import ctypes as C       # on_device=False: the pointer is a host address (the CPU harness of tests/ hands out host memory as device memory)
    return torch.from_numpy(np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)))


def broadcast_vocab(vocab, src, rank, group=None, device=0, on_device=True):
    """`vocab`: the candidate on rank `src` (anything elsewhere).  Returns a vocabulary on every rank: the original on `src`, an imported
    one (tm_vocab_block_import: device tables only) elsewhere, filled by ONE broadcast of the device block (RCCL over xGMI with the nccl
    backend; gloo stages it through the host)."""
    import torch
    import torch.distributed as dist
    from .vocab import Vocab, VocabBlock
    import ctypes as C
    where = "cuda" if on_device and dist.get_backend(group) == "nccl" else "cpu"
    n_meta = C.sizeof(VocabBlock)
    if rank == src:
        desc, ptr, nbytes = vocab.export_block()
        meta = torch.frombuffer(bytearray(desc), dtype=torch.uint8).to(where)
    else:
        meta = torch.zeros(n_meta, dtype=torch.uint8, device=where)
    dist.broadcast(meta, src, group=group)
    if rank != src:
        vocab, ptr, nbytes = Vocab.import_block(bytes(meta.cpu().numpy().tobytes()), device)
    block = _alias(ptr, nbytes, on_device)
    if on_device:
        torch.cuda.synchronize()          # (the exporter's upload ran on the library's copy stream)
    dist.broadcast(block, src, group=group)
    if on_device:
        torch.cuda.synchronize()
    return vocab
