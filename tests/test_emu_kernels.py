"""CPU leg: the LOGIC of the HIP kernels without a GPU.

tools/emu compiles the kernel sources of libtokenmonster_hip.so a second time, for the host (work-items are fibers that meet
at every cross-lane operation and barrier: tools/emu/hip/hip_runtime.h), and the -m gpu parity tests run against that library
in a child process (tests/conftest.py under TM_EMU=1).  This is how a kernel change is checked where no GPU can be had; it says
nothing about what the gfx950 compiler makes of the code or about speed, and it is no parity claim — those rest on the real
-m gpu run.  The subset below takes about two and a half minutes; `TM_EMU=1 python -m pytest tests -m gpu` runs everything (≈ 10 minutes)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAST = ("unit_golden_vector or fuzz_micro_vocab or fuzz_capcode1 or fuzz_utf16 or dense_forward_delete or fallback_paths or "
        "score_histogram_micro or (score_ranges_of_one_walk and micro) or host_api_edge_cases or golden_fixture_through_hip or "
        "device_normalizer_equals_reference_js or device_decode_equals_reference_js or capcode_decode or "
        "normalizer_against_the_host or (against_the_oracle and 1007) or jobs_5_to_9 or (raw_text_to_ids and 501) or come_and_go")


def test_emulation_library_exports_the_c_abi():
    import ctypes as C
    sys.path.insert(0, os.path.join(ROOT, "tools", "emu"))
    import build_emu
    from tokenmonster_amd import _native as N
    lib = C.CDLL(build_emu.build())
    for name in N.SIGNATURES:
        getattr(lib, name)


def test_gpu_parity_subset_on_the_emulated_device():
    env = dict(os.environ, TM_EMU="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "tests/test_gpu_golden.py", "tests/test_gpu_fuzz.py", "tests/test_gpu_server_jobs.py", "tests/test_gpu_host_api.py", "-q", "-m", "gpu", "-x", "-k", FAST,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=1500)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    m = re.search(r"(\d+) passed", out)
    assert m and int(m.group(1)) >= 30, out[-2000:]
