#!/usr/bin/env python3
"""tools/gpu_fuzz.py [seconds] [first_seed] — the differential fuzz cases of tests/fuzz_cases.py on the REAL device (tools/emu/fuzz.py runs them on the
emulated one): tokenizer, raw text -> ids in one device pass, device normalizer, device capcode decoder, round robin, as many seeds as the time allows."""
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.pop("TM_EMU", None)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import conftest  # noqa: E402,F401  (arms the test hooks)
import fuzz_cases  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cases = [("tokenizer", fuzz_cases.one), ("raw", fuzz_cases.one_raw), ("normalizer", fuzz_cases.one_norm), ("decoder", fuzz_cases.one_decode)]
    done = {n: [0, 0] for n, _ in cases}
    t0 = time.time()
    while time.time() - t0 < budget:
        for name, fn in cases:
            done[name][1] += fn(seed)
            done[name][0] += 1
        seed += 1
    print("gpu fuzz ok: seeds up to %d, %.0f s: %s" % (seed - 1, time.time() - t0, ", ".join("%s %d cases %.1f MB" % (n, c, b / 1e6) for n, (c, b) in done.items())))


if __name__ == "__main__":
    main()
