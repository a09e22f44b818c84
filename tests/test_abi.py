"""CPU: the C-ABI library loads and exports every symbol the public headers declare (no compute calls)."""
import ctypes as C
import os
import re

from tokenmonster_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tm_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = C.CDLL(N.LIB_PATH)
    names = declared_symbols("tokenmonster_hip.h") + declared_symbols("tm_build.h")
    assert len(names) > 30
    for n in names:
        assert hasattr(lib, n), "libtokenmonster_hip.so does not export %s" % n
        assert n in N.SIGNATURES, "%s is declared in a header but not bound in _native.SIGNATURES" % n
    for n in N.SIGNATURES:
        assert n in names, "%s is bound but not declared in a public header" % n


def test_error_path_without_compute():
    # malformed .vocab is rejected before any device work
    h = C.c_void_p()
    rc = N.lib.tm_vocab_load(b"\x07\x07\x07", 3, C.byref(h))
    assert rc == N.TM_E_INVALID and b"truncated" in N.lib.tm_last_error()
    assert N.lib.tm_kernel_name(1) == b"match_branch"


def test_no_oracle_in_product():
    # the product package must not reference the oracle (voids parity claims otherwise)
    pkg = os.path.join(ROOT, "tokenmonster_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle_bind" not in src and "libtm_oracle" not in src and "libtmref" not in src, f
