"""Builds libtokenmonster_hip.so (HIP kernels + C ABI, gfx950) in-tree with hipcc.

The shared library is the product; there is no Python/CPU fallback for it."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtokenmonster_hip.so")
# test / benchmark support (synthetic vocabularies and corpora): its own library, not part of the product
SUPPORT_LIB = os.path.join(HERE, "libtm_testsupport.so")
SUPPORT_SOURCES = [os.path.join(HERE, "testsupport", "tm_synth.cpp")]
SOURCES = ["tm_vocab.hip", "tm_kernels.hip", "tm_score.hip", "tm_norm.hip", "tm_decode.hip", "tm_host.hip", "tm_decoder.hip", "tm_formats.hip", "tm_multi.hip", "tm_build.cpp", "tm_normalize.cpp"]
HEADERS = ["tm_device.h", "tm_internal.h", "tm_tables.h", "tm_pipeline.h", "tm_norm_masks.h", "../../include/tokenmonster_hip.h", "../../include/tm_build.h"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libtokenmonster_hip.so cannot be built")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    hipcc = _hipcc()
    common = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
              "-Wall", "-Wno-unused-result"] + os.environ.get("TM_EXTRA_FLAGS", "").split()
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    procs = []
    for s in srcs:
        o = os.path.join(CSRC, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + deps[len(srcs):]):
            # (the plain C++ units see the HIP headers too - tm_device.h's host structures use its vector types - but no device code)
            rocm_inc = os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(hipcc))), "include")
            cmd = [hipcc] + common + (["-x", "hip"] if s.endswith(".hip") else ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I", rocm_inc]) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, out.decode(errors="replace")))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", LIB] + objs + ["-licuuc", "-licui18n", "-lz", "-ldl"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout.decode(errors="replace"))
    if force or _stale(SUPPORT_LIB, SUPPORT_SOURCES + [LIB, os.path.join(CSRC, "tm_internal.h"), os.path.join(ROOT, "include", "tm_testsupport.h")]):
        cmd = [hipcc, "-O2", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", os.path.join(ROOT, "include"), "-I", CSRC, "-Wall"] + SUPPORT_SOURCES + [
            "-o", SUPPORT_LIB, "-L" + HERE, "-ltokenmonster_hip", "-Wl,-rpath,$ORIGIN"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("test-support library failed to build:\n" + r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
