#!/usr/bin/env python3
"""Regression pins for the vocabulary builder (tokenmonster_amd/csrc/tm_build.cpp): md5 of the .vocab image it writes for a set of
seeded token lists (capcode 0/1/2, UTF-8 and UTF-16, special tokens, with and without an unk token).  The images were produced by
the map-per-concept implementation that mirrors go/tokenmonster.go:3423-3793 line by line; the single-table implementation that
replaced it must write the same bytes.  (The Go builder itself cannot run here: these pin the restatement against itself over
time, the rule-level tests in tests/test_builder_normalizer.py pin it against the reference's rules.)
    python tests/golden/make_builder_golden.py > tests/golden/builder_images.json"""
import hashlib
import json
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def cases():
    rng = random.Random(7)
    alphabet = [b"a", b"b", b"c", b" ", b"'", b"s", b"D", b"C", b"W", b"1", b"2", b"_", b".", b"\xc3\xa9", b"\xe2\x80\x99", b"\x7f", b"A", b"\n"]
    out = []
    for cc in (0, 1, 2):
        toks = set()
        while len(toks) < 2500:
            toks.add(b"".join(rng.choice(alphabet) for _ in range(rng.randint(1, 9))))
        toks = sorted(toks)
        sp = [1 if rng.random() < 0.01 else 0 for _ in toks]
        out.append(("fuzz-capcode%d-special" % cc, toks, cc, 1, sp))
        out.append(("fuzz-capcode%d" % cc, toks, cc, 1, None))
    toks = set()
    while len(toks) < 1500:
        s = "".join(rng.choice("ab c's1_.é’A") for _ in range(rng.randint(1, 8)))
        toks.add(s.encode("utf-16-le"))
    out.append(("utf16-capcode0", sorted(toks), 0, 2, None))
    out.append(("utf16-capcode2", sorted(toks), 2, 2, None))
    # long tokens: "D "-duplicates that would exceed 40 bytes, multi-word tokens, suffixes
    words = [b"the", b"of", b"and", b"cat's", b"dog\xe2\x80\x99s", b"x_y", b"42", b"A1", b"\xc3\xa9t\xc3\xa9"]
    toks = set()
    while len(toks) < 1500:
        t = b" ".join(rng.choice(words) for _ in range(rng.randint(1, 7)))
        if rng.random() < 0.3:
            t = b" " + t
        toks.add(t[:40])
    out.append(("phrases-capcode2", sorted(toks), 2, 1, None))
    out.append(("phrases-capcode0", sorted(toks), 0, 1, None))
    return out


def digests():
    from tokenmonster_amd import synth
    res = {}
    for name, toks, cc, cs, sp in cases():
        for unk in (False, True):
            img = synth.build_vocab(toks, capcode=cc, charset=cs, norm_flag=1, level=3, with_unk=unk, special=sp)
            res["%s%s" % (name, "-unk" if unk else "")] = hashlib.md5(bytes(img)).hexdigest()
    return res


if __name__ == "__main__":
    json.dump(digests(), sys.stdout, indent=1, sort_keys=True)
    print()
