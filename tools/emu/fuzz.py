#!/usr/bin/env python3
"""tools/emu/fuzz.py [seconds] [first_seed] [norm|decode|raw] — differential fuzzing of the kernel logic on the EMULATED device (tools/emu) against the
oracle (default) or of the device normalizer against the host normalizer (`norm`), or of the device capcode decoder against the host decoder (`decode`): the cases of tests/fuzz_cases.py, as many seeds as
the time allows.  Device allocations end at guard pages there, so an out-of-bounds access of a kernel is a crash, not silence.
Development aid: run it after touching a kernel when no GPU is at hand."""
import os
import sys
import time

os.environ["TM_EMU"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import conftest  # noqa: E402,F401  (under TM_EMU=1: libtokenmonster_hip.so -> libtokenmonster_emu.so, before tokenmonster_amd is imported)
import fuzz_cases  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    case = {"norm": fuzz_cases.one_norm, "decode": fuzz_cases.one_decode, "raw": fuzz_cases.one_raw}.get(sys.argv[3] if len(sys.argv) > 3 else "", fuzz_cases.one)
    t0 = time.time()
    n = nbytes = 0
    while time.time() - t0 < budget:
        nbytes += case(seed)
        seed += 1
        n += 1
    print("fuzz ok: %d cases, %.1f MB, seeds up to %d, %.0f s" % (n, nbytes / 1e6, seed - 1, time.time() - t0))


if __name__ == "__main__":
    main()
