"""ctypes binding of libtokenmonster_hip.so (include/tokenmonster_hip.h, include/tm_build.h).

The library is the product path.  If it is missing the import raises: there is no Python or CPU
fallback for tokenization."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtokenmonster_hip.so")

TM_OK, TM_E_INVALID, TM_E_NODEVICE, TM_E_HIP, TM_E_NOSPACE, TM_E_LIMIT, TM_E_INPUT, TM_E_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7
TM_NONE = 0xFFFFFF
TM_NUM_KERNELS = 5
KIND_ENGLISH, KIND_ENGLISHCODE, KIND_CODE = 0, 1, 2

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
vp = C.c_void_p

# name -> (restype, argtypes): every symbol the two public headers declare
SIGNATURES = {
    "tm_last_error": (C.c_char_p, []),
    "tm_device_count": (C.c_int, []),
    "tm_set_device": (C.c_int, [C.c_int]),
    "tm_vocab_load": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
    "tm_vocab_load_on": (C.c_int, [vp, C.c_size_t, C.c_int, C.POINTER(vp)]),
    "tm_vocab_block_export": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "tm_vocab_block_import": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp)]),
    "tm_device_copy": (C.c_int, [vp, vp, C.c_uint64]),
    "tm_vocab_free": (None, [vp]),
    "tm_vocab_tune": (C.c_int, [vp, vp, C.c_uint64]),
    "tm_vocab_load_sample": (C.c_int, [vp, C.c_size_t, vp, C.c_uint64, C.POINTER(vp)]),
    "tm_vocab_set_tune": (C.c_int, [vp, vp, C.c_uint64]),
    "tm_vocab_size": (C.c_uint32, [vp]),
    "tm_vocab_n_info": (C.c_uint32, [vp]),
    "tm_vocab_n_ids": (C.c_uint32, [vp]),
    "tm_vocab_max_token_length": (C.c_uint32, [vp]),
    "tm_vocab_capcode": (C.c_uint32, [vp]),
    "tm_vocab_charset": (C.c_uint32, [vp]),
    "tm_vocab_normalization": (C.c_uint32, [vp]),
    "tm_vocab_unk": (C.c_uint32, [vp]),
    "tm_vocab_delete_token": (C.c_uint32, [vp]),
    "tm_vocab_device_bytes": (C.c_uint64, [vp]),
    "tm_tokenize_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, C.c_uint64, vp, vp]),
    "tm_count_batch": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp]),
    "tm_count_batch_raw": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp]),
    "tm_tokenize_batch_serialized": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_uint32, vp, C.c_uint64, vp, vp, u32p]),
    "tm_tokenize_pipeline": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, vp, C.c_uint64, vp, vp, u32p, vp]),
    "tm_host_alloc": (vp, [C.c_size_t]),
    "tm_device_numa_node": (C.c_int, [C.c_int]),
    "tm_host_free": (None, [vp]),
    "tm_host_register": (C.c_int, [vp, C.c_size_t]),
    "tm_host_unregister": (C.c_int, [vp]),
    "tm_batch_create": (C.c_int, [vp, C.c_uint64, C.c_uint32, C.POINTER(vp)]),
    "tm_batch_free": (None, [vp]),
    "tm_batch_upload": (C.c_int, [vp, vp, vp, C.c_uint32]),
    "tm_batch_upload_raw": (C.c_int, [vp, vp, vp, C.c_uint32]),
    "tm_batch_normalize": (C.c_int, [vp, vp]),
    "tm_batch_normalized_bytes": (C.c_uint64, [vp]),
    "tm_batch_host_fallback_docs": (C.c_uint32, [vp]),
    "tm_batch_download_text": (C.c_int, [vp, vp, C.c_uint64, vp]),
    "tm_batch_run": (C.c_int, [vp, vp]),
    "tm_batch_run_timed": (C.c_int, [vp, vp, f32p]),
    "tm_kernel_name": (C.c_char_p, [C.c_int]),
    "tm_debug_flags": (C.c_int, [C.c_int]),
    "tm_batch_totals": (C.c_int, [vp, u64p, u64p]),
    "tm_batch_download": (C.c_int, [vp, vp, C.c_uint64, vp, vp]),
    "tm_batch_device_tokens": (vp, [vp]),
    "tm_batch_device_tok_offsets": (vp, [vp]),
    "tm_batch_device_bytes": (C.c_uint64, [vp]),
    "tm_decode_batch": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_int, vp, C.c_uint64, vp]),
    "tm_decode_host_docs": (C.c_uint32, []),
    "tm_batch_decode": (C.c_int, [vp, C.c_int, vp, u64p, u32p]),
    "tm_batch_decode_timed": (C.c_int, [vp, C.c_int, vp, u64p, u32p, f32p]),
    "tm_batch_decoded_download": (C.c_int, [vp, vp, C.c_uint64, vp]),
    "tm_decoder_new": (C.c_int, [vp, C.POINTER(vp)]),
    "tm_decoder_free": (None, [vp]),
    "tm_decoder_decode": (C.c_int, [vp, vp, C.c_uint64, vp, C.c_uint64, u64p]),
    "tm_decoder_decode_serialized": (C.c_int, [vp, vp, C.c_uint64, C.c_uint32, vp, C.c_uint64, u64p]),
    "tm_decoder_flush": (C.c_int, [vp, vp, C.c_uint64, u64p]),
    "tm_dataset_upload": (C.c_int, [vp, C.c_uint64, C.POINTER(vp)]),
    "tm_dataset_upload_on": (C.c_int, [vp, C.c_uint64, C.c_int, C.POINTER(vp)]),
    "tm_dataset_device": (C.c_int, [vp]),
    "tm_dataset_free": (None, [vp]),
    "tm_score": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, u64p, vp]),
    "tm_score_device": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, C.POINTER(vp), u64p]),
    "tm_score_device_into": (C.c_int, [vp, vp, vp, vp, C.c_uint32, vp, vp, C.c_uint64]),
    "tm_score_begin": (C.c_int, [vp, vp, C.c_uint64, C.c_uint64, C.c_int, vp, vp]),
    "tm_score_finish": (C.c_int, [vp, vp, C.c_uint32, vp, vp, C.c_uint64]),
    "tm_score_read": (C.c_int, [vp, vp, vp, u64p, vp]),
    "tm_devices_open": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "tm_devices_open_list": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "tm_devices_count": (C.c_int, [vp]),
    "tm_devices_device": (C.c_int, [vp, C.c_int]),
    "tm_devices_rccl_ranks": (C.c_int, [vp, C.POINTER(C.c_char_p)]),
    "tm_devices_close": (None, [vp]),
    "tm_vocab_load_all": (C.c_int, [vp, vp, C.c_size_t, C.POINTER(vp)]),
    "tm_vocab_set_count": (C.c_int, [vp]),
    "tm_vocab_set_member": (vp, [vp, C.c_int]),
    "tm_vocab_set_free": (None, [vp]),
    "tm_tokenize_pipeline_multi": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_int, C.c_uint32, C.c_uint64, C.c_uint32, vp, C.c_uint64, vp, vp, u32p, vp]),
    "tm_dataset_upload_sharded": (C.c_int, [vp, vp, C.c_uint64, C.POINTER(vp)]),
    "tm_dataset_set_range": (C.c_uint64, [vp, C.c_int, u64p]),
    "tm_dataset_set_free": (None, [vp]),
    "tm_score_multi": (C.c_int, [vp, vp, vp, u64p, vp]),
    # tm_build.h
    "tm_free": (None, [vp]),
    "tm_build_vocab": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int,
                                 C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "tm_vocab_build": (C.c_int, [vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(vp)]),
    "tm_vocab_build_all": (C.c_int, [vp, vp, vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]),
    "tm_vocab_image": (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "tm_vocab_save": (C.c_int, [vp, C.c_char_p]),
    "tm_tok_read": (C.c_int, [vp, C.c_size_t, vp, C.POINTER(vp), C.POINTER(vp), u32p, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), u32p]),
    "tm_tok_write": (C.c_int, [vp, vp, vp, C.c_uint32, vp, vp, vp, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "tm_normalize": (C.c_int, [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "tm_denormalize": (C.c_int, [vp, C.c_size_t, C.c_uint32, C.POINTER(vp), C.POINTER(C.c_size_t)]),
    "tm_normalize_batch": (C.c_int, [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(vp), vp]),
}


# include/tm_testsupport.h: libtm_testsupport.so (synthetic vocabularies / corpora for tests and bench.py; not the product)
SUPPORT_LIB_PATH = os.path.join(_HERE, "libtm_testsupport.so")
SUPPORT_SIGNATURES = {
    "tm_synth_corpus": (C.c_int, [C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint32, vp, vp, C.c_uint32, u32p, u64p]),
    "tm_synth_vocab": (C.c_int, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int,
                                 C.POINTER(vp), C.POINTER(C.c_size_t)]),
}


class TokenMonsterHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libtokenmonster_hip: error %d: %s" % (code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s is missing - run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the HIP tokenizer)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()
_support = None


def support_lib():
    """libtm_testsupport.so, loaded on first use (tests, tools and bench.py only)"""
    global _support
    if _support is None:
        if not os.path.exists(SUPPORT_LIB_PATH):
            raise ImportError("%s is missing - run `python -c 'import __graft_entry__ as g; g.build()'`" % SUPPORT_LIB_PATH)
        sl = C.CDLL(SUPPORT_LIB_PATH)
        for name, (res, args) in SUPPORT_SIGNATURES.items():
            fn = getattr(sl, name)
            fn.restype = res
            fn.argtypes = args
        _support = sl
    return _support


def check(rc):
    if rc != TM_OK:
        raise TokenMonsterHipError(rc, (lib.tm_last_error() or b"").decode(errors="replace"))


def ptr(a):
    """raw pointer of a contiguous numpy array (or None)"""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data


def take(p, n):
    """copy a malloc'd buffer returned by the library into bytes and free it"""
    try:
        return C.string_at(p.value, n) if n else b""
    finally:
        lib.tm_free(p)


def as_u8(data):
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data, dtype=np.uint8)
    return np.frombuffer(bytes(data), dtype=np.uint8)
