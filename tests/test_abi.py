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
    # the synthetic generators are test / benchmark support: their own library, absent from the product's
    sup = C.CDLL(N.SUPPORT_LIB_PATH)
    for n in declared_symbols("tm_testsupport.h"):
        assert hasattr(sup, n) and n in N.SUPPORT_SIGNATURES and n not in N.SIGNATURES
        assert not hasattr(lib, n), "libtokenmonster_hip.so still exports the test-support symbol %s" % n


def test_every_declared_symbol_has_a_go_binding():
    """go/*.go (cgo, build tag `hip`) cannot be compiled here - no Go toolchain - but it can be kept in step: every tm_* symbol the public headers
    declare is called somewhere in it, and it calls nothing the headers do not declare (a renamed or removed entry point shows up here)."""
    godir = os.path.join(ROOT, "go")
    src = "".join(open(os.path.join(godir, f)).read() for f in sorted(os.listdir(godir)) if f.endswith(".go"))
    src = re.sub(r"//[^\n]*", "", src)
    used = set(re.findall(r"\bC\.(tm_[a-z0-9_]+)\s*\(", src))
    names = set(declared_symbols("tokenmonster_hip.h") + declared_symbols("tm_build.h"))
    missing = sorted(names - used)
    assert not missing, "no Go binding calls %s" % missing
    unknown = sorted(used - names)
    assert not unknown, "go/*.go calls %s, which no public header declares" % unknown


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


def test_no_emulation_in_product():
    """tools/emu (the kernel sources built for the host) is test infrastructure: the product's Python layer must not know of it, the
    product library must not have been built with it (no emulation runtime symbols), and without a GPU the product refuses to work."""
    import subprocess
    pkg = os.path.join(ROOT, "tokenmonster_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f), errors="replace").read()
            assert "emu" not in src.lower().replace("enumerate", ""), f
    r = subprocess.run(["nm", "-D", "--defined-only", os.path.join(pkg, "libtokenmonster_hip.so")], stdout=subprocess.PIPE)
    syms = r.stdout.decode(errors="replace")
    assert "emu_switch" not in syms and "launch_impl" not in syms
    from tokenmonster_amd import _native as N
    if N.lib.tm_device_count() == 0:          # (this container: no GPU)
        import tokenmonster_amd as tm
        from conftest import unit_vocab_image
        try:
            tm.Vocab(unit_vocab_image())
            raise AssertionError("the product library worked without a GPU")
        except N.TokenMonsterHipError as e:
            assert e.code in (N.TM_E_NODEVICE, N.TM_E_HIP)


def test_headers_are_plain_c_and_example_links(tmp_path):
    """include/*.h must be usable from C (the cgo stub of INTEGRATION.md compiles them as C), and the library must link
    into a program that knows nothing of Python or torch."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include "tokenmonster_hip.h"\n#include "tm_build.h"\nint main(void) { return tm_device_count() < 0; }\n')
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                        "-L", os.path.join(root, "tokenmonster_amd"), "-ltokenmonster_hip",
                        "-Wl,-rpath," + os.path.join(root, "tokenmonster_amd")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode(errors="replace")
    r = subprocess.run(["make", "-C", os.path.join(root, "examples")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode(errors="replace")


def test_library_leaves_none_of_its_own_symbols_undefined():
    """a declaration with the wrong linkage (extern "C" against C++) links into a shared library without a word and fails at the first
    program that links against it"""
    import subprocess
    from tokenmonster_amd import build as _b
    lib = os.path.join(os.path.dirname(os.path.abspath(_b.__file__)), "libtokenmonster_hip.so")
    out = subprocess.run(["nm", "-D", "--undefined-only", lib], stdout=subprocess.PIPE, check=True).stdout.decode()
    own = [l.split()[-1] for l in out.splitlines() if l.split() and (l.split()[-1].startswith(("tm_", "tmh", "_ZN3tmh")) or "tmh" in l.split()[-1])]
    assert own == [], own
