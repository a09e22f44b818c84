#!/usr/bin/env python3
"""Generates tests/golden/realtext.json.gz (committed) and tests/golden/_realtext_docs.bin.gz (git-ignored; travels to the GPU box with the
snapshot like oracle/_ref): REAL prose and code — every *.md / *.go / *.js / *.cpp / *.hpp / *.py / *.yaml under /root/reference (about
0.8 MB: the reference's READMEs, its Go / JS / C++ / Python sources), one document per file and one per 4 KiB slice of it (cut on UTF-8
character boundaries) — tokenized by the REFERENCE's own C++ runtime (oracle/_ref/libtmref.so) with two vocabularies built from the only
real token list in the tree, yaml_guide/gpt2.json:
  (a) "gpt2"        capcode 0, charset UTF-8, no normalization: the image of tests/golden/gpt2_vocab.json.gz
  (b) "gpt2-capcode2-nfd"  the same token list as the reference's normalizer (capcode 2 + NFD) writes each token - lower case, capital
                    markers - built with capcode 2, norm_flag 1 (the builder adds the "D " duplicates itself)
The committed fixture holds, per document, the sha1 of its raw bytes and the ids / missing / count the reference returns for it under
both vocabularies, plus image (b); the documents' TEXT is the reference's own files and stays out of the repository's history: it is
written beside the fixture into a git-ignored file (the -m gpu test skips its text-dependent half where that file is missing and the
reference tree is not there to make it again).

    python tests/golden/make_realtext_golden.py

Runs in the build container only (needs /root/reference and oracle/_ref)."""
import base64
import gzip
import hashlib
import json
import os
import struct
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle_bind import Reference  # noqa: E402
from tokenmonster_amd import synth  # noqa: E402

REF = "/root/reference"
EXTS = (".md", ".go", ".js", ".cpp", ".hpp", ".py", ".yaml")
SLICE = 4096


def collect():
    files = []
    for d, _, names in os.walk(REF):
        for n in names:
            if n.endswith(EXTS):
                files.append(os.path.join(d, n))
    files.sort()
    docs, names = [], []
    for p in files:
        b = open(p, "rb").read()
        rel = os.path.relpath(p, REF)
        docs.append(b); names.append(rel)
        if len(b) > SLICE:
            at, k = 0, 0
            while at < len(b):
                end = min(at + SLICE, len(b))
                while end < len(b) and (b[end] & 0xC0) == 0x80:      # not inside a character
                    end -= 1
                docs.append(b[at:end]); names.append("%s[%d]" % (rel, k))
                at, k = end, k + 1
    return docs, names


def write_docs(docs, path):
    with gzip.GzipFile(path, "wb", compresslevel=9, mtime=0) as f:
        f.write(struct.pack("<I", len(docs)))
        for d in docs:
            f.write(struct.pack("<I", len(d)))
            f.write(d)


def read_docs(path):
    b = gzip.open(path, "rb").read()
    n = struct.unpack_from("<I", b, 0)[0]
    at, docs = 4, []
    for _ in range(n):
        l = struct.unpack_from("<I", b, at)[0]
        docs.append(b[at + 4:at + 4 + l]); at += 4 + l
    return docs


def capcode_vocab(gpt2_tokens):
    """token list (b): every GPT-2 token as the reference's normalizer writes it under capcode 2 + NFD"""
    probe = Reference(synth.build_vocab([b"a", b"D", b"C", b"W", b" "], capcode=2, charset=1, norm_flag=1, level=5))
    toks, seen = [], set()
    for t in [b"D", b"C", b"W"] + gpt2_tokens:
        e = probe.normalize(t) if t not in (b"D", b"C", b"W") else t
        if e.startswith(b"D ") and len(e) > 2:        # a mid-word fragment: the builder adds this duplicate of the plain token itself (go :3450-3462)
            e = e[2:]
        if 0 < len(e) <= 40 and e not in seen:
            seen.add(e); toks.append(e)
    return synth.build_vocab(toks, capcode=2, charset=1, norm_flag=1, level=5), len(toks)


def main():
    docs, names = collect()
    g = json.loads(gzip.open(os.path.join(HERE, "gpt2_vocab.json.gz")).read())
    img_a = base64.b64decode(g["vocab_b64"])
    j = json.load(open(os.path.join(REF, "yaml_guide", "gpt2.json")))
    gpt2_tokens = []
    for s, _ in sorted(j.items(), key=lambda kv: kv[1]):
        t = s.replace("Ġ", " ").replace("Ċ", "\n").replace("č", "\r").replace("ĉ", "\t").encode()
        if 0 < len(t) <= 40:
            gpt2_tokens.append(t)
    img_b, n_b = capcode_vocab(gpt2_tokens)
    out = {"note": "real text: %d documents (%d files of /root/reference + their 4 KiB slices), %d bytes; ids / missing / count from the reference C++ "
                   "runtime on the RAW bytes (Vocab::tokenize = normalize + walk)" % (len(docs), sum(1 for n in names if "[" not in n), sum(map(len, docs))),
           "generator": "tests/golden/make_realtext_golden.py via oracle/_ref/libtmref.so",
           "names": names, "sha1": [hashlib.sha1(d).hexdigest() for d in docs], "bytes": [len(d) for d in docs],
           "vocab_b_b64": base64.b64encode(img_b).decode(), "vocabs": {}}
    for key, img in (("gpt2", img_a), ("gpt2-capcode2-nfd", img_b)):
        ref = Reference(img)
        ids, missing, count, nbytes = [], [], [], []
        for d in docs:
            t, m = ref.tokenize(d)
            nd = ref.normalize(d)
            c, _ = ref.count_normalized(nd)
            assert t.size == 0 or int(t.max()) < 65536
            ids.append(base64.b64encode(t.astype("<u2").tobytes()).decode())
            missing.append(int(m)); count.append(int(c)); nbytes.append(len(nd))
        out["vocabs"][key] = {"ids_u16_b64": ids, "missing": missing, "count": count, "normalized_bytes": nbytes}
        print("%-20s %d documents, %d tokens, %d missing, %d normalized bytes" % (key, len(docs), sum(len(base64.b64decode(x)) // 2 for x in ids), sum(missing), sum(nbytes)))
    with gzip.GzipFile(os.path.join(HERE, "realtext.json.gz"), "wb", compresslevel=9, mtime=0) as f:
        f.write(json.dumps(out).encode())
    write_docs(docs, os.path.join(HERE, "_realtext_docs.bin.gz"))
    print("vocabulary (b): %d tokens, image %d bytes; fixture %d bytes, text %d bytes (git-ignored)" % (
        n_b, len(img_b), os.path.getsize(os.path.join(HERE, "realtext.json.gz")), os.path.getsize(os.path.join(HERE, "_realtext_docs.bin.gz"))))


if __name__ == "__main__":
    main()
