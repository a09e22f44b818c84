"""CPU: fixtures written by the reference's Go implementation (go_oracle/dump.go -> tests/golden/go_*.json), consumed WHEN PRESENT.
No Go toolchain exists in the image this library was developed in, so the files are not in the tree; whoever has Go and network
access runs the three commands in go_oracle/dump.go's header and this test pins
  * the vocabulary builder: tm_build_vocab's image of the fixture's token list == the bytes the reference's NewVocab + Save wrote
    (go/tokenmonster.go:3423-3793, :2602-2653), byte for byte;
  * the walk: the oracle's ids / missing / count on the host-normalized documents == the reference's Tokenize / Count on the raw ones
    (tests/test_gpu_golden.py runs the same fixtures through the HIP path under -m gpu).
Until then the builder is pinned by hand-worked vectors (tests/test_builder_handworked.py) and against itself over time
(tests/golden/builder_images.json): DESIGN.md's parity table says so."""
import base64
import glob
import json
import os

import numpy as np
import pytest

from oracle_bind import Oracle
from tokenmonster_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "go_*.json")))


def test_go_fixture_generator_is_in_the_tree():
    root = os.path.dirname(GOLDEN.rstrip("/").rsplit("/tests", 1)[0] + "/x")
    assert os.path.exists(os.path.join(root, "go_oracle", "dump.go")) and os.path.exists(os.path.join(root, "go_oracle", "make_cases.py"))


@pytest.mark.skipif(not FIXTURES, reason="tests/golden/go_*.json absent: written by go_oracle/dump.go, which needs a Go toolchain and network access "
                                         "(neither exists in this image); the builder stays pinned by hand-worked vectors only")
@pytest.mark.parametrize("path", FIXTURES or ["-"])
def test_builder_and_walk_against_the_go_implementation(path):
    g = json.load(open(path))
    toks = [base64.b64decode(t) for t in g["tokens_b64"]]
    special = g.get("special") or None
    img = synth.build_vocab(toks, capcode=g["capcode"], charset=g["charset"], norm_flag=0, level=5, special=special)
    ref_img = base64.b64decode(g["vocab_b64"])
    assert len(img) == len(ref_img), "%s: image of %d bytes, the Go implementation wrote %d" % (g["name"], len(img), len(ref_img))
    if img != ref_img:
        at = next(i for i in range(len(img)) if img[i] != ref_img[i])
        raise AssertionError("%s: images differ first at byte %d" % (g["name"], at))
    if g["capcode"] == 1:
        return        # capcode level 1 normalization has no statement in the reference tree: the image is all that can be compared
    orc = Oracle(ref_img)
    for doc_b64, ids, missing, count in zip(g["docs_b64"], g["ids"], g["missing"], g["count"]):
        norm = synth.normalize(base64.b64decode(doc_b64), g["capcode"], 0)
        got, miss = orc.tokenize(np.frombuffer(norm, dtype=np.uint8))
        assert got.tolist() == ids and miss == missing
        assert orc.count(np.frombuffer(norm, dtype=np.uint8))[0] == count


@pytest.mark.gpu
@pytest.mark.skipif(not FIXTURES, reason="tests/golden/go_*.json absent (go_oracle/dump.go needs a Go toolchain and network access)")
@pytest.mark.parametrize("path", FIXTURES or ["-"])
def test_go_fixture_through_hip(path):
    """the same fixtures through the HIP path: raw documents -> tm_tokenize (normalize + walk) == the reference's Tokenize"""
    import tokenmonster_amd as tm
    g = json.load(open(path))
    if g["capcode"] == 1:
        pytest.skip("capcode level 1 normalization is not implemented (no statement in the reference tree)")
    v = tm.Vocab(base64.b64decode(g["vocab_b64"]))
    got = v.tokenize([base64.b64decode(d) for d in g["docs_b64"]])
    for k, ids in enumerate(g["ids"]):
        assert got[k].tolist() == ids, "document %d" % k
