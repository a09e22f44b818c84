#!/usr/bin/env python3
"""Generates tests/golden/gpt2_vocab.json.gz: a vocabulary built from the ONLY real token list in the reference tree,
/root/reference/yaml_guide/gpt2.json (50 257 GPT-2 tokens), tokenized by the REFERENCE's own C++ runtime
(oracle/_ref/libtmref.so).  Runs in the build container only; the gzipped JSON is committed.

    python tests/golden/make_gpt2_golden.py

Token strings are converted the way yaml_guide/convert_gpt2tokenizer.py:46-49 does (the byte-level stand-ins for space,
newline, carriage return and tab are replaced, everything else is UTF-8 encoded as it stands); header as that script writes it:
charset utf-8, capcode 0, normalization none.  IDs are assigned by this repository's vocabulary builder (tm_build_vocab), not
GPT-2's: the fixture pins the walk on a real token set, not GPT-2's numbering."""
import base64
import gzip
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_bind import Reference  # noqa: E402
from tokenmonster_amd import synth  # noqa: E402

PROSE = [
    "The quick brown fox jumps over the lazy dog. It was the best of times, it was the worst of times.",
    "In the beginning the Universe was created. This has made a lot of people very angry and been widely regarded as a bad move.",
    "Tokenization is the process of splitting text into smaller units called tokens, which may be words, subwords or characters.",
    "She said, \"I don't know what you're talking about,\" and walked away without another word. He couldn't believe it.",
    "On 12 March 2021 the company reported revenue of $4,382,100.57, up 13.2% year-over-year; shares rose 4.5% after hours.",
    "def fibonacci(n):\n    if n < 2:\n        return n\n    return fibonacci(n - 1) + fibonacci(n - 2)\n\nprint(fibonacci(10))\n",
    "for (int i = 0; i < n; ++i) {\n\tsum += values[i] * weights[i];\n}\nreturn sum / static_cast<double>(n);\n",
    "{\"name\": \"example\", \"version\": \"1.0.3\", \"dependencies\": {\"left-pad\": \"^1.3.0\"}, \"private\": true}",
    "2023-09-24T21:58:03Z INFO  [worker-7] request_id=af31c9 status=200 latency_ms=12.7 path=/api/v1/users/42",
    "Antidisestablishmentarianism and pneumonoultramicroscopicsilicovolcanoconiosis are unusually long English words.",
    "WASHINGTON (Reuters) - The U.S. Senate voted 52-48 on Tuesday to confirm the nominee, ending weeks of debate.",
    "https://www.example.com/path/to/resource?query=string&other=value#fragment user@example.org +1 (555) 010-9999",
    "   leading spaces,\ttabs\tand  double  spaces.  Trailing whitespace   \n\n\nMultiple newlines above.",
    "e = mc^2; a^2 + b^2 = c^2; f(x) = \\sum_{i=0}^{n} x_i; 3.14159265358979 2.718281828 1e-9 0xDEADBEEF",
    "Ünïcödé tëxt — “curly quotes”, ellipsis… and emoji 🙂 are not in GPT-2's printable stand-ins as raw bytes.",
    "", " ", "a", "The", " the", "\n",
]


def main():
    j = json.load(open("/root/reference/yaml_guide/gpt2.json"))
    space_char, newline_char, carriage_char, tab_char = "Ġ", "Ċ", "č", "ĉ"
    toks, seen = [], set()
    for s, _id in sorted(j.items(), key=lambda kv: kv[1]):
        t = s.replace(space_char, " ").replace(newline_char, "\n").replace(carriage_char, "\r").replace(tab_char, "\t").encode()
        if 0 < len(t) <= 40 and t not in seen:
            seen.add(t)
            toks.append(t)
    img = synth.build_vocab(toks, capcode=0, charset=1, norm_flag=0, level=5)
    ref = Reference(img)
    rng = np.random.default_rng(0x47505432)
    docs = [p.encode() for p in PROSE]
    # documents assembled from the token list itself, rank-weighted (low GPT-2 ids = frequent merges): long matches, every length
    ranks = np.arange(1, len(toks) + 1, dtype=np.float64)
    w = 1.0 / ranks
    w /= w.sum()
    for n in rng.integers(5, 400, size=60):
        docs.append(b"".join(toks[i] for i in rng.choice(len(toks), size=int(n), p=w)))
    docs.append(b"".join(toks[i] for i in rng.choice(len(toks), size=6000, p=w)))      # spans many 256-byte segments
    ids, missing, count = [], [], []
    for d in docs:
        t, m = ref.tokenize_normalized(d)
        c, _ = ref.count_normalized(d)
        ids.append([int(x) for x in t])
        missing.append(int(m))
        count.append(int(c))
    out = {"note": "vocabulary built from /root/reference/yaml_guide/gpt2.json (%d tokens), capcode 0, charset utf-8, no normalization; "
                   "ids/missing/count from the reference C++ runtime" % len(toks),
           "generator": "tests/golden/make_gpt2_golden.py via oracle/_ref/libtmref.so",
           "vocab_b64": base64.b64encode(img).decode(), "docs_b64": [base64.b64encode(d).decode() for d in docs],
           "ids": ids, "missing": missing, "count": count}
    with gzip.GzipFile(os.path.join(HERE, "gpt2_vocab.json.gz"), "wb", compresslevel=9, mtime=0) as f:
        f.write(json.dumps(out).encode())
    print("gpt2 vocabulary: %d tokens, image %d bytes; %d docs, %d tokens, %d missing" % (
        len(toks), len(img), len(docs), sum(len(x) for x in ids), sum(missing)))


if __name__ == "__main__":
    main()
