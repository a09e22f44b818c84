"""Several GPUs behind one handle, driven from inside the library (include/tokenmonster_hip.h, "several devices"; tm_multi.hip).

This is the path a single-process host (the Go library, tokenmonsterserver, trainvocab) uses for N > 1 GPUs: no torch, no ranks —
the library runs one host thread per device and does the one collective of the path (the histogram all-reduce of the scoring pass,
training/trainvocab.go:909-922 + 1105-1174) with RCCL itself.  tokenmonster_amd/dist.py is the one-process-per-GPU form of the same
protocol over torch.distributed."""
import ctypes as C

import numpy as np

from . import _native as N
from .vocab import PipelineStats


class Devices:
    """tm_devices: `devices` = None (every visible device, or TM_VIRTUAL_DEVICES members on device 0), an int (the first n), or a
    list of device numbers (a device may appear more than once: virtual devices on a one-GPU box)."""

    def __init__(self, devices=None):
        self._h = C.c_void_p()
        if devices is None or isinstance(devices, int):
            N.check(N.lib.tm_devices_open(int(devices or 0), C.byref(self._h)))
        else:
            lst = np.ascontiguousarray(devices, dtype=np.int32)
            N.check(N.lib.tm_devices_open_list(N.ptr(lst), lst.size, C.byref(self._h)))

    handle = property(lambda self: self._h)

    def __len__(self):
        return N.lib.tm_devices_count(self._h)

    def device(self, member):
        return N.lib.tm_devices_device(self._h, member)

    def rccl_ranks(self):
        """-> (ranks of the RCCL communicator the handle uses, reason if there is none)"""
        why = C.c_char_p()
        n = N.lib.tm_devices_rccl_ranks(self._h, C.byref(why))
        return n, (why.value or b"").decode()

    def close(self):
        if getattr(self, "_h", None):
            N.lib.tm_devices_close(self._h)
            self._h = None


class VocabSet:
    """tm_vocab_set: one replica of a vocabulary per member of `devices`"""

    def __init__(self, devices, image):
        self.devices = devices
        self._h = C.c_void_p()
        buf = np.frombuffer(bytes(image), dtype=np.uint8)
        N.check(N.lib.tm_vocab_load_all(devices.handle, N.ptr(buf), buf.size, C.byref(self._h)))

    @classmethod
    def from_tokens(cls, devices, tokens, capcode=0, charset=1, norm_flag=0, level=5, with_unk=False):
        """tm_vocab_build_all: a candidate's tables built once from its token list, the device block replicated to every member"""
        tokens = [bytes(t) for t in tokens]
        blob = np.frombuffer(b"".join(tokens), dtype=np.uint8) if tokens else np.zeros(0, np.uint8)
        off = np.zeros(len(tokens) + 1, dtype=np.uint32)
        np.cumsum([len(t) for t in tokens], out=off[1:])
        s = cls.__new__(cls)
        s.devices, s._h = devices, C.c_void_p()
        N.check(N.lib.tm_vocab_build_all(devices.handle, N.ptr(blob), N.ptr(off), len(tokens), None, capcode, charset, norm_flag, level, 1 if with_unk else 0, C.byref(s._h)))
        return s

    handle = property(lambda self: self._h)

    def member(self, i):
        return C.c_void_p(N.lib.tm_vocab_set_member(self._h, i))

    def n_ids(self):
        return N.lib.tm_vocab_n_ids(self.member(0))

    def tune(self, normalized_sample):
        """tm_vocab_set_tune: tables by use on every member"""
        data = N.as_u8(normalized_sample)
        N.check(N.lib.tm_vocab_set_tune(self._h, N.ptr(data), data.size))

    def tokenize_pipeline(self, text, offsets, raw=True, encoding_length=0, chunk_bytes=0, lanes_per_device=0, out=None):
        """tm_tokenize_pipeline_multi -> (serialized ids u8, byte_offsets u64[D+1], missing u32[D], encoding length, stats)"""
        text = N.as_u8(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nd = offsets.size - 1
        boff = np.zeros(nd + 1, dtype=np.uint64)
        missing = np.zeros(max(nd, 1), dtype=np.uint32)
        enc = C.c_uint32()
        stats = PipelineStats()
        if out is None:
            out = np.empty(int(text.size) + 8 * nd + 64, dtype=np.uint8)
        while True:
            rc = N.lib.tm_tokenize_pipeline_multi(self._h, N.ptr(text), N.ptr(offsets), nd, 1 if raw else 0, encoding_length, chunk_bytes, lanes_per_device,
                                                  N.ptr(out), out.size, N.ptr(boff), N.ptr(missing), C.byref(enc), C.byref(stats))
            if rc == N.TM_E_NOSPACE:
                out = np.empty(int(boff[nd]), dtype=np.uint8)
                continue
            N.check(rc)
            st = {k: getattr(stats, k) for k, _ in PipelineStats._fields_}
            return out[: int(boff[nd])], boff, missing[:nd], enc.value, st

    def close(self):
        if getattr(self, "_h", None):
            N.lib.tm_vocab_set_free(self._h)
            self._h = None


class DatasetSet:
    """tm_dataset_set: a normalized dataset cut into one byte range (+ 128-byte halo) per member"""

    def __init__(self, devices, normalized):
        self.devices = devices
        data = N.as_u8(normalized)
        self._h = C.c_void_p()
        N.check(N.lib.tm_dataset_upload_sharded(devices.handle, N.ptr(data), data.size, C.byref(self._h)))

    handle = property(lambda self: self._h)

    def ranges(self):
        """[(bytes of the member's range, bytes of halo behind it)]"""
        out = []
        for i in range(len(self.devices)):
            halo = C.c_uint64()
            out.append((int(N.lib.tm_dataset_set_range(self._h, i, C.byref(halo))), int(halo.value)))
        return out

    def score(self, vocab_set):
        """tm_score_multi: ONE whole-buffer walk over all members -> (scores u32[n_ids], tokens_in_text, missing_set u8[32])"""
        scores = np.zeros(vocab_set.n_ids(), dtype=np.uint32)
        ntok = C.c_uint64()
        missing = np.zeros(32, dtype=np.uint8)
        N.check(N.lib.tm_score_multi(vocab_set.handle, self._h, N.ptr(scores), C.byref(ntok), N.ptr(missing)))
        return scores, int(ntok.value), missing

    def close(self):
        if getattr(self, "_h", None):
            N.lib.tm_dataset_set_free(self._h)
            self._h = None
