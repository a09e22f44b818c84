"""Builds tools/emu/libtokenmonster_emu.so: the SAME sources as libtokenmonster_hip.so, compiled for the host against
tools/emu/hip/hip_runtime.h (work-items as fibers, see that header).  Test infrastructure / development aid — the product
never loads it (tokenmonster_amd/_native.py binds libtokenmonster_hip.so and nothing else)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "tokenmonster_amd", "csrc")
LIB = os.path.join(HERE, "libtokenmonster_emu.so")
BUILD = os.path.join(HERE, "build")
SOURCES = ["tm_vocab.hip", "tm_kernels.hip", "tm_score.hip", "tm_norm.hip", "tm_decode.hip", "tm_host.hip", "tm_decoder.hip", "tm_formats.hip", "tm_multi.hip",
           "tm_build.cpp", "tm_normalize.cpp"]


def _cxx():
    for cand in (os.environ.get("TM_EMU_CXX"), "/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++"), shutil.which("g++")):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("no host C++ compiler for the emulation build")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False, extra=()):
    cxx = _cxx()
    os.makedirs(BUILD, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [
        os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "tokenmonster_hip.h"), os.path.join(ROOT, "include", "tm_build.h"),
        os.path.abspath(__file__)]
    common = ["-O1", "-g", "-std=c++17", "-fPIC", "-fno-omit-frame-pointer", "-I", HERE, "-I", os.path.join(ROOT, "include"), "-I", CSRC,
              "-Wall", "-Wno-unused-result", "-Wno-unused-variable", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-pass-failed",
              "-x", "c++"] + list(extra) + os.environ.get("TM_EMU_EXTRA_FLAGS", "").split()
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "emu_runtime.cpp")]
    objs, procs = [], []
    for s in srcs:
        o = os.path.join(BUILD, os.path.basename(s) + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cxx] + common + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((s, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("emulation build failed on %s:\n%s" % (s, out.decode(errors="replace")[-6000:]))
        if verbose and out:
            print(out.decode(errors="replace"), file=sys.stderr)
    if force or procs or _stale(LIB, objs):
        cmd = [cxx, "-shared", "-fPIC", "-o", LIB] + objs + ["-licuuc", "-licui18n", "-lz", "-lpthread", "-ldl"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
        if r.returncode != 0:
            raise RuntimeError("emulation link failed:\n" + r.stdout.decode(errors="replace"))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
