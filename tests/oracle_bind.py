"""ctypes bindings of the CPU checkers under oracle/ (TEST INFRASTRUCTURE: only tests/, smoke() and
bench.py's cpu_baseline leg import this).

  Oracle     oracle/libtm_oracle.so   our plain-C restatement (oracle/tm_oracle.c)
  Reference  oracle/_ref/libtmref.so  the reference's own C++ runtime compiled unmodified from
                                      /root/reference/tokenmonster-cpp (oracle/Makefile)
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "libtm_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libtmref.so")
REF_UNIT = os.path.join(ORACLE_DIR, "_ref", "unit")
REF_BENCH = os.path.join(ORACLE_DIR, "_ref", "bench")      # the reference's own tests/bench.cpp


def build_oracles():
    subprocess.run(["make", "-C", ORACLE_DIR, "all"], check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)


def have_ref():
    return os.path.exists(REF_SO)


def _u8(a):
    return np.ascontiguousarray(np.frombuffer(bytes(a), dtype=np.uint8) if not isinstance(a, np.ndarray) else a, dtype=np.uint8)


class Oracle:
    def __init__(self, image):
        if not os.path.exists(ORACLE_SO):
            build_oracles()
        L = C.CDLL(ORACLE_SO)
        L.tmo_load.restype = C.c_void_p
        L.tmo_load.argtypes = [C.c_void_p, C.c_size_t]
        L.tmo_free.argtypes = [C.c_void_p]
        L.tmo_last_error.restype = C.c_char_p
        for n in ("tmo_vocab_size", "tmo_n_info", "tmo_max_token_length", "tmo_n_reverse", "tmo_capcode"):
            getattr(L, n).restype = C.c_uint32
            getattr(L, n).argtypes = [C.c_void_p]
        L.tmo_tokenize.restype = C.c_longlong
        L.tmo_tokenize.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_longlong)]
        L.tmo_count.restype = C.c_longlong
        L.tmo_count.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_longlong)]
        L.tmo_score.restype = None
        L.tmo_score.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
        L.tmo_decode_raw.restype = C.c_longlong
        L.tmo_decode_raw.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.tmo_longest.restype = C.c_int
        L.tmo_longest.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self.L = L
        img = _u8(image)
        self.h = L.tmo_load(img.ctypes.data, img.size)
        if not self.h:
            raise RuntimeError("oracle load failed: " + L.tmo_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.tmo_free(self.h)
            self.h = None

    def n_ids(self):
        return self.L.tmo_n_reverse(self.h)

    def n_info(self):
        return self.L.tmo_n_info(self.h)

    def tokenize(self, data):
        d = _u8(data)
        cap = 2 * d.size + 8
        out = np.empty(cap, dtype=np.uint32)
        miss = C.c_longlong()
        n = self.L.tmo_tokenize(self.h, d.ctypes.data, d.size, out.ctypes.data, cap, C.byref(miss))
        return out[:n].copy(), miss.value

    def count(self, data):
        d = _u8(data)
        miss = C.c_longlong()
        n = self.L.tmo_count(self.h, d.ctypes.data, d.size, C.byref(miss))
        return n, miss.value

    def score(self, data, scores=None):
        d = _u8(data)
        if scores is None:
            scores = np.zeros(self.n_ids(), dtype=np.uint32)
        tit = C.c_uint64(0)
        ms = np.zeros(32, dtype=np.uint8)
        self.L.tmo_score(self.h, d.ctypes.data, d.size, scores.ctypes.data, C.byref(tit), ms.ctypes.data)
        return scores, tit.value, ms

    def score_range(self, data, lo, hi, entry_state=0):
        """scoring mode over bytes [lo, hi) of ONE walk over `data`, entered in `entry_state` (2 * offset + forwardDelete)
        -> (scores, tokens_in_text, missing_set, exit_state)"""
        d = _u8(data)
        self.L.tmo_score_range.restype = None
        self.L.tmo_score_range.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.c_void_p, C.POINTER(C.c_uint64),
                                           C.c_void_p, C.POINTER(C.c_uint32)]
        scores = np.zeros(self.n_ids(), dtype=np.uint32)
        tit = C.c_uint64(0)
        ms = np.zeros(32, dtype=np.uint8)
        ex = C.c_uint32(0)
        self.L.tmo_score_range(self.h, d.ctypes.data, d.size, lo + (entry_state >> 1), entry_state & 1, hi, scores.ctypes.data, C.byref(tit),
                               ms.ctypes.data, C.byref(ex))
        return scores, tit.value, ms, ex.value

    def score_mt(self, data, threads, strip=1 << 20, warm=1024):
        """tmo_score of the whole text on `threads` threads, exact (strips entered in guessed states, chained and redone where a guess was
        wrong) -> (scores, tokens_in_text, missing_set, strips redone)"""
        d = _u8(data)
        self.L.tmo_score_strips_mt.restype = C.c_longlong
        self.L.tmo_score_strips_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]
        scores = np.zeros(self.n_ids(), dtype=np.uint32)
        tit = C.c_uint64(0)
        ms = np.zeros(32, dtype=np.uint8)
        redone = self.L.tmo_score_strips_mt(self.h, d.ctypes.data, d.size, strip, warm, threads, scores.ctypes.data, C.byref(tit), ms.ctypes.data)
        return scores, tit.value, ms, int(redone)

    def decode_raw(self, toks):
        t = np.ascontiguousarray(toks, dtype=np.uint32)
        cap = 40 * t.size + 8
        out = np.empty(cap, dtype=np.uint8)
        n = self.L.tmo_decode_raw(self.h, t.ctypes.data, t.size, out.ctypes.data, cap)
        return out[:n].tobytes()

    def longest(self, key):
        k = _u8(key)
        i, l = C.c_uint32(), C.c_uint32()
        f = self.L.tmo_longest(self.h, k.ctypes.data, k.size, C.byref(i), C.byref(l))
        return (i.value, l.value, bool(f))


class Reference:
    """the reference's own C++ runtime (oracle/_ref/libtmref.so)"""

    def __init__(self, image):
        L = C.CDLL(REF_SO)
        L.tmref_load.restype = C.c_void_p
        L.tmref_load.argtypes = [C.c_char_p]
        L.tmref_free.argtypes = [C.c_void_p]
        L.tmref_last_error.restype = C.c_char_p
        for n in ("tmref_tokenize_normalized", "tmref_tokenize"):
            getattr(L, n).restype = C.c_longlong
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        L.tmref_count_normalized.restype = C.c_longlong
        L.tmref_count_normalized.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int)]
        for n in ("tmref_normalize",):
            getattr(L, n).restype = C.c_longlong
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        for n in ("tmref_decode", "tmref_decode_raw"):
            getattr(L, n).restype = C.c_longlong
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        for n in ("tmref_vocab_size", "tmref_max_token_length", "tmref_capcode", "tmref_charset", "tmref_normalization"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p]
        self.L = L
        with tempfile.NamedTemporaryFile(suffix=".vocab", delete=False) as f:
            f.write(bytes(image))
            path = f.name
        try:
            self.h = L.tmref_load(path.encode())
        finally:
            os.unlink(path)
        if not self.h:
            raise RuntimeError("reference load failed: " + L.tmref_last_error().decode())

    def __del__(self):
        if getattr(self, "h", None):
            self.L.tmref_free(self.h)
            self.h = None

    def _tok(self, fn, data):
        d = _u8(data)
        cap = 2 * d.size + 8
        out = np.empty(cap, dtype=np.uint32)
        miss = C.c_int()
        n = fn(self.h, d.ctypes.data, d.size, out.ctypes.data, cap, C.byref(miss))
        if n < 0:
            raise RuntimeError("reference tokenize failed")
        return out[:n].copy(), miss.value

    def tokenize_normalized(self, data):
        return self._tok(self.L.tmref_tokenize_normalized, data)

    def tokenize(self, data):
        return self._tok(self.L.tmref_tokenize, data)

    def count_normalized(self, data):
        d = _u8(data)
        miss = C.c_int()
        n = self.L.tmref_count_normalized(self.h, d.ctypes.data, d.size, C.byref(miss))
        return n, miss.value

    def normalize(self, data):
        d = _u8(data)
        cap = 4 * d.size + 64
        out = np.empty(cap, dtype=np.uint8)
        n = self.L.tmref_normalize(self.h, d.ctypes.data, d.size, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("reference normalize failed")
        return out[:n].tobytes()

    def decode_raw(self, toks):
        t = np.ascontiguousarray(toks, dtype=np.uint32)
        cap = 40 * t.size + 8
        out = np.empty(cap, dtype=np.uint8)
        n = self.L.tmref_decode_raw(self.h, t.ctypes.data, t.size, out.ctypes.data, cap)
        return out[:n].tobytes()

    def tokenize_docs_mt(self, text, offsets, raw, threads):
        """all documents, `threads` std::threads inside the shim (one document per call, like the server's goroutines) -> #tokens"""
        t = _u8(text)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        self.L.tmref_tokenize_docs_mt.restype = C.c_longlong
        self.L.tmref_tokenize_docs_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32]
        n = self.L.tmref_tokenize_docs_mt(self.h, t.ctypes.data, o.ctypes.data, o.size - 1, 1 if raw else 0, threads)
        if n < 0:
            raise RuntimeError("reference tokenize (multi-threaded) failed")
        return int(n)

    def verify_docs_mt(self, text, offsets, raw, threads, ids, toff, missing=None):
        """every document through the reference on `threads` threads, its ids (and `missing`) compared with ids[toff[d]:toff[d+1]]
        -> (documents that differ, lowest differing document or None, the reference's token total)"""
        t = _u8(text)
        o = np.ascontiguousarray(offsets, dtype=np.uint64)
        i = np.ascontiguousarray(ids, dtype=np.uint32)
        f = np.ascontiguousarray(toff, dtype=np.uint64)
        m = None if missing is None else np.ascontiguousarray(missing, dtype=np.uint32)
        self.L.tmref_verify_docs_mt.restype = C.c_longlong
        self.L.tmref_verify_docs_mt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.POINTER(C.c_uint32), C.POINTER(C.c_longlong)]
        fb, nt = C.c_uint32(), C.c_longlong()
        bad = self.L.tmref_verify_docs_mt(self.h, t.ctypes.data, o.ctypes.data, o.size - 1, 1 if raw else 0, threads, i.ctypes.data, f.ctypes.data,
                                          None if m is None else m.ctypes.data, C.byref(fb), C.byref(nt))
        if bad < 0:
            raise RuntimeError("reference verification (multi-threaded) failed")
        return int(bad), (None if bad == 0 else int(fb.value)), int(nt.value)

    def decode(self, toks):
        """Vocab::decode (tokenmonster.cpp:1404-1425): decode_raw + capcode / charset post-processing"""
        t = np.ascontiguousarray(toks, dtype=np.uint32)
        cap = 40 * t.size + 8
        out = np.empty(cap, dtype=np.uint8)
        n = self.L.tmref_decode(self.h, t.ctypes.data, t.size, out.ctypes.data, cap)
        if n < 0:
            raise RuntimeError("reference decode failed")
        return out[:n].tobytes()


class ReferenceDecoder:
    """the reference's streaming Decoder (tokenmonster.cpp:1509-1721) on a Reference vocabulary"""

    def __init__(self, ref):
        self.ref, L = ref, ref.L
        L.tmref_decoder_new.restype = C.c_void_p
        L.tmref_decoder_new.argtypes = [C.c_void_p]
        L.tmref_decoder_free.argtypes = [C.c_void_p]
        L.tmref_decoder_decode.restype = C.c_longlong
        L.tmref_decoder_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.tmref_decoder_decode_serialized.restype = C.c_longlong
        L.tmref_decoder_decode_serialized.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_size_t]
        L.tmref_decoder_flush.restype = C.c_longlong
        L.tmref_decoder_flush.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        self.h = L.tmref_decoder_new(ref.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.ref.L.tmref_decoder_free(self.h)
            self.h = None

    def decode(self, toks):
        t = np.ascontiguousarray(toks, dtype=np.uint32)
        out = np.empty(40 * t.size + 64, dtype=np.uint8)
        n = self.ref.L.tmref_decoder_decode(self.h, t.ctypes.data, t.size, out.ctypes.data, out.size)
        assert n >= 0
        return out[:n].tobytes()

    def decode_serialized(self, data, enc):
        d = _u8(data)
        out = np.empty(40 * d.size + 64, dtype=np.uint8)
        n = self.ref.L.tmref_decoder_decode_serialized(self.h, d.ctypes.data, d.size, enc, out.ctypes.data, out.size)
        assert n >= 0
        return out[:n].tobytes()

    def flush(self):
        out = np.empty(64, dtype=np.uint8)
        n = self.ref.L.tmref_decoder_flush(self.h, out.ctypes.data, out.size)
        assert n >= 0
        return out[:n].tobytes()


STAT_NAMES = ["s1", "s2", "s3", "s1b", "s2b", "s3b", "fast_exit", "no_lookahead_match", "not_found"]


def oracle_stats(reset=True):
    L = C.CDLL(ORACLE_SO)
    out = (C.c_uint64 * 9)()
    L.tmo_stats(out, 1 if reset else 0)
    return dict(zip(STAT_NAMES, [int(x) for x in out]))
