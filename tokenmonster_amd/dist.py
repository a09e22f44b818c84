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


def shard_strips(n_bytes, rank, world, align=4):
    """byte range of one contiguous dataset for `rank` (trainvocab cuts strips on multiples of 4, trainvocab.go:1674).
    Each rank walks its range as an independent strip - the same approximation the reference makes at strip
    boundaries; world == 1 is exact."""
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
