"""CPU: the walk tables tm_vocab_load builds (byte trie as two-slot hash buckets, direct two-byte map, suffix links, child
filters; tokenmonster_amd/csrc/tm_tables.h) checked without a GPU by tools/tables_check.cpp: structural invariants the kernels
rely on, and the walk of k_match_branch step A1 replayed against a brute-force longest-prefix search over the .vocab keys
(pansearch LongestSubstring semantics, tokenmonster-cpp/src/tokenmonster.cpp:786-877)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_walk_tables_against_brute_force(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    exe = str(tmp_path / "tables_check")
    libdir = os.path.join(ROOT, "tokenmonster_amd")
    r = subprocess.run([hipcc, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(libdir, "csrc"),
                        os.path.join(ROOT, "tools", "tables_check.cpp"), "-o", exe, "-L" + libdir, "-ltokenmonster_hip", "-ltm_testsupport", "-Wl,-rpath," + libdir],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-3000:]
    r = subprocess.run([exe, str(384 * 1024)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0 and "\n0 failures" in out, out[-3000:]
