#!/usr/bin/env python3
"""Generates tests/golden/*.json with the REFERENCE's own C++ runtime (oracle/_ref/libtmref.so, compiled
unmodified from /root/reference/tokenmonster-cpp by oracle/Makefile).  Run in the build container only
(the GPU box has no /root/reference); the JSON files are committed.

    python tests/golden/make_golden.py

Each file: {"vocab_b64": .vocab image, "docs_b64": [normalized documents], "ids": [[...]], "missing": [...],
"count": [...]} — ids/missing/count are what tokenmonster::Vocab::tokenize_normalized /
tokenize_count_normalized return (tokenmonster.cpp:1723, :1993)."""
import base64
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import fuzz_text, fuzz_vocab_tokens, unit_vocab_image  # noqa: E402
from oracle_bind import Reference  # noqa: E402
from tokenmonster_amd import synth  # noqa: E402


def case(name, img, docs, note):
    ref = Reference(img)
    ids, missing, count = [], [], []
    for d in docs:
        t, m = ref.tokenize_normalized(d)
        c, _ = ref.count_normalized(d)
        ids.append([int(x) for x in t])
        missing.append(int(m))
        count.append(int(c))
    out = {"note": note, "generator": "tests/golden/make_golden.py via oracle/_ref/libtmref.so",
           "vocab_b64": base64.b64encode(img).decode(), "docs_b64": [base64.b64encode(d).decode() for d in docs],
           "ids": ids, "missing": missing, "count": count}
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f)
    print(name, len(docs), "docs", sum(len(x) for x in ids), "tokens")


def main():
    case("unit_cpp", unit_vocab_image(), [b"ab a z", b"ab", b"", b"zzz", b" a b ab"],
         "the vocabulary and first vector of tokenmonster-cpp/tests/unit.cpp:50-112")
    for capcode, seed in ((0, 11), (2, 12), (2, 13)):
        rng = np.random.default_rng(seed)
        toks = fuzz_vocab_tokens(rng, capcode, 120)
        img = synth.build_vocab(toks, capcode=capcode, charset=1, with_unk=(seed == 13))
        docs = [fuzz_text(rng, capcode, int(n)) for n in rng.integers(0, 1500, size=24)] + [fuzz_text(rng, capcode, 5000)]
        case("fuzz_capcode%d_seed%d" % (capcode, seed), img, docs, "random micro-vocabulary over a tiny alphabet")
    img = synth.synth_vocab(synth.ENGLISHCODE, 2048, capcode=2, norm_flag=1, level=3, seed=0x474F4C44)
    raw, offs = synth.synth_corpus(synth.ENGLISHCODE, 60_000, seed=5)
    text, noff = synth.normalize_batch(raw, offs, 2, 1)
    docs = [text[int(noff[d]):int(noff[d + 1])].tobytes() for d in range(noff.size - 1)]
    case("englishcode2048", img, docs, "synthetic englishcode vocabulary of 2048 ids, synthetic mixed documents")
    # raw -> normalized pairs from the reference's normalize() (with oracle/capcode/capcode.hpp: capcode parity UNPINNED)
    ref = Reference(img)
    raws = [b"Hello World", b"hello WORLD and HTTPServer2Go", "Café Über naïve".encode(), b"it's John's 3rd 42nd",
            b"  MiXeD CaSe\tTabs\nNewLine(x){y}", b"", b"ALLCAPS", b"a", b"A", b"x.Y", "“Quoted” — dash".encode()]
    pairs = [{"raw_b64": base64.b64encode(r).decode(), "norm_b64": base64.b64encode(ref.normalize(r)).decode()} for r in raws]
    with open(os.path.join(HERE, "normalize_capcode2_nfd.json"), "w") as f:
        json.dump({"note": "Vocab::normalize (NFD + capcode level 2) of the reference runtime built with oracle/capcode/capcode.hpp",
                   "capcode": 2, "norm_flag": 1, "pairs": pairs}, f)
    print("normalize pairs", len(pairs))


if __name__ == "__main__":
    main()
