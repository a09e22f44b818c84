"""Vocab: host-side mirror of the reference's tokenize API over the HIP library.

Method names follow python/tokenmonster.py (tokenize :410, tokenize_count :497, len, max_token_length,
capcode, charset, unk_token_id ...) and go/tokenmonster.go:953-1014; every tokenize variant is a batch
call into libtokenmonster_hip.so.  Decoding, vocabulary editing, YAML import/export and the training
CLIs are out of scope and are not provided."""
import ctypes as C

import numpy as np

from . import _native as N
from . import synth as _synth


def pack_documents(docs):
    """list of bytes-like -> (text u8[n], offsets u64[len+1]) : the packed layout of the C ABI"""
    docs = [bytes(d) if not isinstance(d, (bytes, bytearray)) else d for d in docs]
    offsets = np.zeros(len(docs) + 1, dtype=np.uint64)
    if docs:
        np.cumsum([len(d) for d in docs], out=offsets[1:])
    text = np.frombuffer(b"".join(docs), dtype=np.uint8) if docs else np.zeros(0, np.uint8)
    return text, offsets


class VocabBlock(C.Structure):
    """tm_vocab_block (include/tokenmonster_hip.h): what a process needs besides the bytes of a vocabulary's device block"""
    _fields_ = [("bytes", C.c_uint64), ("part_bytes", C.c_uint64 * 8)] + [(n, C.c_uint32) for n in (
        "idle_off", "n_da", "n_info", "max_len", "off", "bstart", "spl_hint", "link_off", "direct_off", "delete_id", "unk_id",
        "n_ids", "vocab_size", "capcode", "charset", "norm_flag", "level", "reserve", "n_nodes", "pad")]


class Vocab:
    def __init__(self, image, sample=None):
        """tm_vocab_load; with `sample` (normalized text the vocabulary will be used on) tm_vocab_load_sample: tables laid out by use from the start"""
        self._image = bytes(image)
        h = C.c_void_p()
        buf = np.frombuffer(self._image, dtype=np.uint8)
        if sample is None:
            N.check(N.lib.tm_vocab_load(N.ptr(buf), buf.size, C.byref(h)))
        else:
            data = np.ascontiguousarray(sample, dtype=np.uint8)
            N.check(N.lib.tm_vocab_load_sample(N.ptr(buf), buf.size, N.ptr(data), data.size, C.byref(h)))
        self._h = h

    # ---- the device block from process to process (tm_vocab_block_export / _import: the data-parallel scoring mode) ----
    def export_block(self):
        """-> (description as bytes, device pointer of the block, its size): send the description, broadcast the block"""
        m, p = VocabBlock(), C.c_void_p()
        N.check(N.lib.tm_vocab_block_export(self._h, C.byref(m), C.byref(p)))
        return bytes(m), int(p.value), int(m.bytes)

    @classmethod
    def import_block(cls, description, device=0):
        """an empty vocabulary of the described shape on `device` -> (Vocab, device pointer to fill with the exporter's block before use)"""
        m = VocabBlock.from_buffer_copy(description)
        h, p = C.c_void_p(), C.c_void_p()
        N.check(N.lib.tm_vocab_block_import(C.byref(m), int(device), C.byref(h), C.byref(p)))
        v = cls.__new__(cls)
        v._image, v._h = b"", h
        return v, int(p.value), int(m.bytes)

    @classmethod
    def from_tokens(cls, tokens, capcode=0, charset=1, norm_flag=0, level=5, with_unk=False, special=None, device=0):
        """a vocabulary straight from a token list (tm_vocab_build: the trainvocab worker's per-candidate step, training/trainvocab.go:530-907),
        without the .vocab image in between; `image()` writes it on request"""
        tokens = [bytes(t) for t in tokens]
        blob = np.frombuffer(b"".join(tokens), dtype=np.uint8) if tokens else np.zeros(0, np.uint8)
        off = np.zeros(len(tokens) + 1, dtype=np.uint32)
        np.cumsum([len(t) for t in tokens], out=off[1:])
        sp = None if special is None else np.ascontiguousarray(np.asarray(special, dtype=np.uint8))
        h = C.c_void_p()
        N.check(N.lib.tm_vocab_build(N.ptr(blob), N.ptr(off), len(tokens), N.ptr(sp), capcode, charset, norm_flag, level, 1 if with_unk else 0, int(device), C.byref(h)))
        v = cls.__new__(cls)
        v._image, v._h = b"", h
        return v

    def image(self):
        """the bytes of the .vocab file of this vocabulary (tm_vocab_image)"""
        p, n = C.c_void_p(), C.c_size_t()
        N.check(N.lib.tm_vocab_image(self._h, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value)

    def close(self):
        if getattr(self, "_h", None):
            N.lib.tm_vocab_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- metadata (python/tokenmonster.py:Vocab properties; go/tokenmonster.go:2366-2597) ----
    def __len__(self):
        return N.lib.tm_vocab_size(self._h)

    @property
    def handle(self):
        return self._h

    def max_token_length(self):
        return N.lib.tm_vocab_max_token_length(self._h)

    def capcode(self):
        return N.lib.tm_vocab_capcode(self._h)

    def charset(self):
        return N.lib.tm_vocab_charset(self._h)

    def normalization_code(self):
        return N.lib.tm_vocab_normalization(self._h)

    def n_info(self):
        return N.lib.tm_vocab_n_info(self._h)

    def n_ids(self):
        return N.lib.tm_vocab_n_ids(self._h)

    def tune(self, normalized_sample):
        """tm_vocab_tune: the tables laid out by use on a sample of NORMALIZED text (results unchanged; worth it for large vocabularies)"""
        data = N.as_u8(normalized_sample)
        N.check(N.lib.tm_vocab_tune(self._h, N.ptr(data), data.size))

    def unk_token_id(self):
        u = N.lib.tm_vocab_unk(self._h)
        return None if u == N.TM_NONE else u

    def delete_token_id(self):
        u = N.lib.tm_vocab_delete_token(self._h)
        return None if u == N.TM_NONE else u

    # ---- normalization pre-step (go/tokenmonster.go:953 Normalize) ----
    def normalize(self, data):
        return _synth.normalize(data, self.capcode(), self.normalization_code())

    def normalize_packed_device(self, raw_text, raw_offsets):
        """norm.Normalize + capcode.Encode of packed raw documents ON THE GPU (tm_batch_normalize; documents with
        other non-ASCII content fall back to the host normalizer inside the library)
        -> (normalized text u8, offsets u64[D+1], number of host-fallback documents)"""
        raw_text = N.as_u8(raw_text)
        raw_offsets = np.ascontiguousarray(raw_offsets, dtype=np.uint64)
        nd = raw_offsets.size - 1
        cap = int(raw_text.size * 4 + 16 * nd + 1024)
        b = C.c_void_p()
        N.check(N.lib.tm_batch_create(self._h, cap, max(nd, 1), C.byref(b)))
        try:
            N.check(N.lib.tm_batch_upload_raw(b, N.ptr(raw_text), N.ptr(raw_offsets), nd))
            N.check(N.lib.tm_batch_normalize(b, None))
            n = int(N.lib.tm_batch_normalized_bytes(b))
            text = np.empty(max(n, 1), dtype=np.uint8)
            offs = np.zeros(nd + 1, dtype=np.uint64)
            N.check(N.lib.tm_batch_download_text(b, N.ptr(text), n, N.ptr(offs)))
            return text[:n], offs, int(N.lib.tm_batch_host_fallback_docs(b))
        finally:
            N.lib.tm_batch_free(b)

    # ---- tokenize ----
    def tokenize_packed(self, text, offsets):
        """normalized packed documents -> (ids u32[T], tok_offsets u64[D+1], missing u32[D])"""
        text = N.as_u8(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nd = offsets.size - 1
        tok_off = np.zeros(nd + 1, dtype=np.uint64)
        missing = np.zeros(max(nd, 1), dtype=np.uint32)
        cap = int(text.size // 2 + 2 * nd + 64)
        while True:
            out = np.empty(cap, dtype=np.uint32)
            rc = N.lib.tm_tokenize_batch(self._h, N.ptr(text), N.ptr(offsets), nd, N.ptr(out), cap, N.ptr(tok_off), N.ptr(missing))
            if rc == N.TM_E_NOSPACE:
                cap = int(tok_off[nd])
                continue
            N.check(rc)
            return out[: int(tok_off[nd])], tok_off, missing[:nd]

    def tokenize_normalized(self, docs):
        """list of already-normalized documents -> list of uint32 arrays (Vocab.tokenize, go :1017)"""
        single = isinstance(docs, (bytes, bytearray, np.ndarray))
        text, offsets = pack_documents([docs] if single else docs)
        ids, off, missing = self.tokenize_packed(text, offsets)
        res = [ids[int(off[d]): int(off[d + 1])] for d in range(offsets.size - 1)]
        return (res[0], int(missing[0])) if single else (res, missing)

    def tokenize(self, docs):
        """raw text -> ids (go :959 Tokenize = normalize + tokenize; python/tokenmonster.py:410).  The normalization
        pre-step runs on the device too (tm_batch_upload_raw + tm_batch_normalize)."""
        single = isinstance(docs, (bytes, bytearray, str))
        lst = [docs] if single else list(docs)
        lst = [d.encode("utf-8") if isinstance(d, str) else d for d in lst]
        text, offsets = pack_documents(lst)
        nd = offsets.size - 1
        b = C.c_void_p()
        N.check(N.lib.tm_batch_create(self._h, int(text.size * 4 + 16 * nd + 1024), max(nd, 1), C.byref(b)))
        try:
            N.check(N.lib.tm_batch_upload_raw(b, N.ptr(text), N.ptr(offsets), nd))
            N.check(N.lib.tm_batch_normalize(b, None))
            N.check(N.lib.tm_batch_run(b, None))
            ntok = C.c_uint64()
            N.check(N.lib.tm_batch_totals(b, C.byref(ntok), None))
            ids = np.empty(max(int(ntok.value), 1), dtype=np.uint32)
            off = np.zeros(nd + 1, dtype=np.uint64)
            N.check(N.lib.tm_batch_download(b, N.ptr(ids), int(ntok.value), N.ptr(off), None))
        finally:
            N.lib.tm_batch_free(b)
        res = [ids[int(off[d]): int(off[d + 1])] for d in range(nd)]
        return res[0] if single else res

    def decode_packed(self, ids, tok_offsets, raw=False):
        """Decode (go/tokenmonster.go:445) of packed id streams -> (bytes u8, offsets u64[D+1]); raw=True: decode_raw"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        tok_offsets = np.ascontiguousarray(tok_offsets, dtype=np.uint64)
        nd = tok_offsets.size - 1
        ooff = np.zeros(nd + 1, dtype=np.uint64)
        cap = int(ids.size * 8 + 64)
        while True:
            out = np.empty(cap, dtype=np.uint8)
            rc = N.lib.tm_decode_batch(self._h, N.ptr(ids), N.ptr(tok_offsets), nd, 1 if raw else 0, N.ptr(out), cap, N.ptr(ooff))
            if rc == N.TM_E_NOSPACE:
                cap = int(ooff[nd])
                continue
            N.check(rc)
            return out[: int(ooff[nd])], ooff

    def roundtrip_resident(self, raw_text, raw_offsets, raw=False):
        """raw documents -> ids -> text without the ids leaving the device: tm_batch_upload_raw + tm_batch_normalize + tm_batch_run, then
        tm_batch_decode on the ids the batch holds -> (decoded text u8, offsets u64[D+1], documents the device left to the host decoder)"""
        raw_text = N.as_u8(raw_text)
        raw_offsets = np.ascontiguousarray(raw_offsets, dtype=np.uint64)
        nd = raw_offsets.size - 1
        b = C.c_void_p()
        N.check(N.lib.tm_batch_create(self._h, int(raw_text.size * 4 + 16 * nd + 1024), max(nd, 1), C.byref(b)))
        try:
            N.check(N.lib.tm_batch_upload_raw(b, N.ptr(raw_text), N.ptr(raw_offsets), nd))
            N.check(N.lib.tm_batch_normalize(b, None))
            N.check(N.lib.tm_batch_run(b, None))
            nbytes, host_docs = C.c_uint64(), C.c_uint32()
            N.check(N.lib.tm_batch_decode(b, 1 if raw else 0, None, C.byref(nbytes), C.byref(host_docs)))
            ooff = np.zeros(nd + 1, dtype=np.uint64)
            cap = int(raw_text.size * 2 + 64)
            while True:
                out = np.empty(cap, dtype=np.uint8)
                rc = N.lib.tm_batch_decoded_download(b, N.ptr(out), cap, N.ptr(ooff))
                if rc == N.TM_E_NOSPACE:
                    cap = int(ooff[nd])
                    continue
                N.check(rc)
                return out[: int(ooff[nd])], ooff, int(host_docs.value)
        finally:
            N.lib.tm_batch_free(b)

    def decode(self, ids):
        """one id sequence -> bytes (python/tokenmonster.py:341 decode)"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out, _ = self.decode_packed(ids, np.array([0, ids.size], dtype=np.uint64))
        return out.tobytes()

    def decoder(self):
        return Decoder(self)

    def count_packed(self, text, offsets):
        """Count (go :971 / :1281): b-branches count once (quirk Q2) -> (counts u64[D], missing u32[D])"""
        text = N.as_u8(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nd = offsets.size - 1
        counts = np.zeros(max(nd, 1), dtype=np.uint64)
        missing = np.zeros(max(nd, 1), dtype=np.uint32)
        N.check(N.lib.tm_count_batch(self._h, N.ptr(text), N.ptr(offsets), nd, N.ptr(counts), N.ptr(missing)))
        return counts[:nd], missing[:nd]

    def tokenize_count(self, docs):
        single = isinstance(docs, (bytes, bytearray, str))
        lst = [docs] if single else list(docs)
        lst = [d.encode("utf-8") if isinstance(d, str) else d for d in lst]
        text, offsets = pack_documents(lst)
        ntext, noff = _synth.normalize_batch(text, offsets, self.capcode(), self.normalization_code())
        counts, _ = self.count_packed(ntext, noff)
        return int(counts[0]) if single else counts

    def tokenize_serialized_packed(self, text, offsets, encoding_length=0):
        """TokenizeToSerialized (go :986) on normalized packed documents -> (bytes u8, byte_offsets, missing, enc)"""
        text = N.as_u8(text)
        offsets = np.ascontiguousarray(offsets, dtype=np.uint64)
        nd = offsets.size - 1
        boff = np.zeros(nd + 1, dtype=np.uint64)
        missing = np.zeros(max(nd, 1), dtype=np.uint32)
        enc = C.c_uint32()
        cap = int(text.size * 2 + 8 * nd + 64)
        while True:
            out = np.empty(cap, dtype=np.uint8)
            rc = N.lib.tm_tokenize_batch_serialized(self._h, N.ptr(text), N.ptr(offsets), nd, encoding_length, N.ptr(out), cap,
                                                    N.ptr(boff), N.ptr(missing), C.byref(enc))
            if rc == N.TM_E_NOSPACE:
                cap = int(boff[nd])
                continue
            N.check(rc)
            return out[: int(boff[nd])], boff, missing[:nd], enc.value


    def tokenize_pipeline(self, text, offsets, raw=True, encoding_length=0, chunk_bytes=0, lanes=0, out=None):
        """host-to-host tokenization of a large packed corpus (tm_tokenize_pipeline): chunks of documents run
        H2D | normalize + tokenize + serialize | D2H on several lanes at once.  -> (serialized ids u8, byte_offsets u64[D+1],
        missing u32[D], encoding length, stats dict).  `text` / `out` may be pinned arrays from pinned_empty()."""
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
            rc = N.lib.tm_tokenize_pipeline(self._h, N.ptr(text), N.ptr(offsets), nd, 1 if raw else 0, encoding_length, chunk_bytes, lanes,
                                            N.ptr(out), out.size, N.ptr(boff), N.ptr(missing), C.byref(enc), C.byref(stats))
            if rc == N.TM_E_NOSPACE:
                out = np.empty(int(boff[nd]), dtype=np.uint8)
                continue
            N.check(rc)
            st = {k: getattr(stats, k) for k, _ in PipelineStats._fields_}
            return out[: int(boff[nd])], boff, missing[:nd], enc.value, st


class Decoder:
    """streaming decoder (go/tokenmonster.go:552-700 NewDecoder; python/tokenmonster.py Decoder): feed ids a few at a time, get the
    text that is complete so far; partial UTF-8 sequences and the capcode state carry over to the next call"""

    def __init__(self, vocab):
        self._vocab = vocab
        self._h = C.c_void_p()
        N.check(N.lib.tm_decoder_new(vocab.handle, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            N.lib.tm_decoder_free(self._h)
            self._h = None

    def _call(self, fn, *args):
        n = C.c_uint64()
        out = np.empty(4096, dtype=np.uint8)
        rc = fn(self._h, *args, N.ptr(out), out.size, C.byref(n))
        if rc == N.TM_E_NOSPACE:         # the ids were consumed; fetch the text with a buffer of the reported size
            out = np.empty(int(n.value), dtype=np.uint8)
            rc = N.lib.tm_decoder_decode(self._h, None, 0, N.ptr(out), out.size, C.byref(n))
        N.check(rc)
        return out[: int(n.value)].tobytes()

    def decode(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        return self._call(N.lib.tm_decoder_decode, N.ptr(ids), ids.size)

    def decode_serialized(self, data, encoding_length=0):
        d = N.as_u8(data)
        return self._call(N.lib.tm_decoder_decode_serialized, N.ptr(d), d.size, encoding_length)

    def flush(self):
        n = C.c_uint64()
        out = np.empty(64, dtype=np.uint8)
        rc = N.lib.tm_decoder_flush(self._h, N.ptr(out), out.size, C.byref(n))
        if rc == N.TM_E_NOSPACE:         # the remainder (a run of continuation bytes can be long) has been kept: fetch it with a buffer of the reported size
            out = np.empty(int(n.value), dtype=np.uint8)
            rc = N.lib.tm_decoder_decode(self._h, None, 0, N.ptr(out), out.size, C.byref(n))
        N.check(rc)
        return out[: int(n.value)].tobytes()


class PipelineStats(C.Structure):
    _fields_ = [("chunks", C.c_uint32), ("lanes", C.c_uint32), ("input_pinned", C.c_int), ("output_pinned", C.c_int),
                ("normalized_bytes", C.c_uint64), ("host_fallback_docs", C.c_uint32), ("ring", C.c_uint32), ("ring_exact_chunks", C.c_uint32)]


class PinnedBuffer:
    """page-locked host memory (tm_host_alloc) viewed as a numpy uint8 array; freed with the object"""

    def __init__(self, nbytes):
        self._p = N.lib.tm_host_alloc(max(int(nbytes), 16))
        if not self._p:
            raise MemoryError("tm_host_alloc(%d) failed" % nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(int(nbytes), 16)).from_address(self._p))[: int(nbytes)]

    def __del__(self):
        if getattr(self, "_p", None):
            self.array = None
            N.lib.tm_host_free(self._p)
            self._p = None


def load(path_or_bytes):
    """load a .vocab file (go/tokenmonster.go:2656 Load; python/tokenmonster.py load) onto the current GPU"""
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return Vocab(path_or_bytes)
    with open(path_or_bytes, "rb") as f:
        return Vocab(f.read())
