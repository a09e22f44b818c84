#!/usr/bin/env python3
"""cases.json for go_oracle/dump.go: the seeded token lists of tests/golden/make_builder_golden.py (capcode 0/1/2, UTF-8 and UTF-16,
special tokens, long multi-word tokens) plus the three hand-worked lists of tests/test_builder_handworked.py, each with a few raw
documents spelled from its own tokens.      python go_oracle/make_cases.py > go_oracle/cases.json"""
import base64
import json
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_builder_golden as mbg     # noqa: E402


def b64(b):
    return base64.b64encode(bytes(b)).decode()


def main():
    rng = random.Random(11)
    out = []
    extra = [
        ("handworked-priority10", [bytes([c]) for c in sorted(set(b" helowrdD"))] + [b" he", b" hello", b" hello world"], 2, 1, None),
        ("handworked-swap", [bytes([c]) for c in sorted(set(b" fobarD"))] + [b"fo", b"foo", b"foo ", b"foo bar"], 2, 1, None),
        ("handworked-suffix", [bytes([c]) for c in sorted(set(b" bo's"))] + [b"D", b"bo", b"bob", b"bob'", b"bob's"], 2, 1, None),
        ("handworked-underscore", [bytes([c]) for c in range(32, 127)] + [b"foo", b"foo_", b"_bar", b"bar", b"foo_bar"], 0, 1, None),
    ]
    for name, toks, capcode, charset, special in mbg.cases() + extra:
        docs = []
        pool = [t for i, t in enumerate(toks) if not (special and special[i])]
        for _ in range(12):
            d = b"".join(rng.choice(pool) for _ in range(rng.randint(1, 60)))
            if charset == 1:
                d = d.decode("utf-8", errors="ignore").encode()       # raw documents must be valid text for Tokenize's normalizer
            docs.append(d)
        out.append({"name": name, "tokens_b64": [b64(t) for t in toks], "special": list(special) if special else [], "capcode": capcode,
                    "charset": charset, "normalization": "", "docs_b64": [b64(d) for d in docs]})
    json.dump(out, sys.stdout)


if __name__ == "__main__":
    main()
