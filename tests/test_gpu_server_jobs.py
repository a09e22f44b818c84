"""-m gpu: the tokenmonsterserver wire protocol (training/tokenmonsterserver.go:184-209, :339-394, :753-800) spoken by
examples/server_jobs.c through the C ABI.  Requests are framed exactly as the reference's Python client frames them
(python/tokenmonster.py:1036-1089: struct.pack('<BIQ', job, id, length)[0:12] + payload); responses are parsed the way the client
parses them and compared, byte for byte, with what the reference's server would send given the same ids."""
import os
import struct
import subprocess

import numpy as np
import pytest

from conftest import example_env
import tokenmonster_amd as tm
from oracle_bind import Oracle
from tokenmonster_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "server_jobs")


def frame(job, vid, payload=b""):
    return struct.pack("<BIQ", job, vid, len(payload))[0:12] + payload


def batches(docs):
    return struct.pack("<I", len(docs)) + b"".join(struct.pack("<Q", len(d)) + d for d in docs)


class Client:
    def __init__(self):
        self.p = subprocess.Popen([EXE], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=example_env())

    def call(self, job, vid, payload=b""):
        self.p.stdin.write(frame(job, vid, payload))
        self.p.stdin.flush()
        head = self.p.stdout.read(9)
        status = head[0]
        if status == 0:                                   # HEADER_IS_LENGTH
            return status, self.p.stdout.read(struct.unpack("<Q", head[1:9])[0])
        return status, struct.unpack("<I", head[1:5])[0]

    def close(self):
        self.p.stdin.close()
        assert self.p.wait(timeout=30) == 0


def parse_batches(body):
    n = struct.unpack("<I", body[:4])[0]
    pos, out = 4, []
    for _ in range(n):
        l = struct.unpack("<Q", body[pos:pos + 8])[0]
        out.append(body[pos + 8:pos + 8 + l])
        pos += 8 + l
    assert pos == len(body)
    return out


def test_jobs_1_20_decode_load_unload(tmp_path):
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.PIPE)
    img = synth.synth_vocab(synth.ENGLISHCODE, 5000, capcode=2, norm_flag=1, level=3, seed=0x53525652)
    path = tmp_path / "v.vocab"
    path.write_bytes(img)
    v, orc = tm.Vocab(img), Oracle(img)
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 400_000, seed=9)
    docs = [raw[int(roffs[d]):int(roffs[d + 1])].tobytes() for d in range(roffs.size - 1)]
    docs += [b"", b"Hello World, this is A TEST of HTTPServer2Go!", "It’s “quoted” — naïve café".encode()]
    norm = [synth.normalize(d, 2, 1) for d in docs]
    exp_ids = [orc.tokenize(x)[0] for x in norm]
    c = Client()
    name = str(path).encode()
    assert c.call(0, 0) == (1, 5)                                                  # version
    assert c.call(10, 0, bytes([len(name)]) + name) == (1, 0)                      # load -> id 0
    assert c.call(10, 0, bytes([3]) + b"/no") == (12, 0)                           # ERROR_FILE_CANNOT_OPEN
    # job 1: one document, then all of them in one request (the goroutine fan-out of :363-378)
    st, body = c.call(1, 0, batches(docs[:1]))
    assert st == 0 and parse_batches(body) == [exp_ids[0].astype("<u2").tobytes()]
    st, body = c.call(1, 0, batches(docs))
    got = parse_batches(body)
    assert st == 0 and len(got) == len(docs)
    for g, e in zip(got, exp_ids):
        assert g == e.astype("<u2").tobytes()
    # job 20: Count on raw text (b-branches count once, go :1281)
    st, body = c.call(20, 0, batches(docs))
    n = struct.unpack("<I", body[:4])[0]
    counts = np.frombuffer(body[4:], dtype="<u8")
    assert st == 0 and n == len(docs) and counts.size == n
    assert counts.tolist() == [orc.count(x)[0] for x in norm]
    # job 2: decode of 2-byte ids (DecodeSerialized, :399-446)
    st, body = c.call(2, 0, batches([e.astype("<u2").tobytes() for e in exp_ids[:40]]))
    assert st == 0 and parse_batches(body) == [v.decode(e) for e in exp_ids[:40]]
    # errors: unknown job, unknown id, unloaded id
    assert c.call(99, 0) == (15, 0)
    assert c.call(1, 7, batches(docs[:1])) == (10, 0)
    assert c.call(11, 0) == (2, 0)
    assert c.call(1, 0, batches(docs[:1])) == (11, 0)
    assert c.call(11, 5) == (10, 0)
    c.close()


def test_job_1_packs_four_bytes_above_65536_ids(tmp_path):
    """training/tokenmonsterserver.go:350-353: 2 bytes per id unless vocab.Len() > 65536, then FOUR (TokenizeToSerialized's own
    automatic choice would be three, go/tokenmonster.go:990-996)"""
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.PIPE)
    img = synth.config_vocab("englishcode-100256-clean")
    path = tmp_path / "big.vocab"
    path.write_bytes(img)
    orc = Oracle(img)
    raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 200_000, seed=10)
    docs = [raw[int(roffs[d]):int(roffs[d + 1])].tobytes() for d in range(roffs.size - 1)]
    c = Client()
    name = str(path).encode()
    assert c.call(10, 0, bytes([len(name)]) + name) == (1, 0)
    st, body = c.call(1, 0, batches(docs))
    got = parse_batches(body)
    assert st == 0
    for g, d in zip(got, docs):
        assert g == orc.tokenize(synth.normalize(d, 2, 1))[0].astype("<u4").tobytes()
    c.close()


def test_jobs_5_to_9_streaming_decoder_and_12_save(tmp_path):
    """the streaming Decoder over the wire (training/tokenmonsterserver.go:449-504: jobs 5 new, 6 unload, 7/8/9 decode with 2/3/4-byte ids)
    and Save (job 12, :537-554).  The ids arrive a few at a time, also cut inside a multi-byte character and inside a capcode marker
    sequence; the pieces joined must be the text of one Decode of all ids, which the reference's own Decoder gives too
    (tests/test_gpu_golden.py pins tm_decoder_* against it)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "examples")], check=True, stdout=subprocess.PIPE)
    img = synth.synth_vocab(synth.ENGLISHCODE, 3000, capcode=2, norm_flag=1, level=3, seed=0x53525653)
    path = tmp_path / "v.vocab"
    path.write_bytes(img)
    v, orc = tm.Vocab(img), Oracle(img)
    text = "Hello World, this is A TEST of HTTPServer2Go! It’s “quoted” — naïve café, ÉCOLE 12AB34cd x".encode()
    ids = orc.tokenize(synth.normalize(text, 2, 1))[0]
    whole = v.decode(ids)
    c = Client()
    name = str(path).encode()
    assert c.call(10, 0, bytes([len(name)]) + name) == (1, 0)
    assert c.call(5, 3) == (10, 0)                                                  # no such vocabulary
    assert c.call(5, 0) == (1, 0) and c.call(5, 0) == (1, 1)                        # two decoders
    rng = np.random.default_rng(3)
    for dec, enc, dt in ((0, 2, "<u2"), (1, 4, "<u4")):
        pieces, pos = [], 0
        while pos < ids.size:
            k = int(rng.integers(1, 4))
            st, body = c.call(5 + enc, dec, ids[pos:pos + k].astype(dt).tobytes())
            assert st == 0
            pieces.append(body)
            pos += k
        got = b"".join(pieces)
        assert whole.startswith(got) and len(whole) - len(got) < 8                  # (what a cut character still holds back stays in the decoder)
    st, body = c.call(8, 0, b"".join(int(i).to_bytes(3, "little") for i in ids[:5]))  # 3-byte ids on a decoder that has seen others
    assert st == 0
    assert c.call(6, 1) == (2, 0) and c.call(9, 1, b"") == (11, 0)                  # unloaded decoder
    assert c.call(5, 0) == (1, 1)                                                   # its slot is handed out again
    assert c.call(7, 9, b"") == (10, 0) and c.call(6, 77) == (4, 0)
    # job 12: Save writes the file back, byte for byte
    out = str(tmp_path / "saved.vocab").encode()
    assert c.call(12, 0, bytes([len(out)]) + out) == (2, 0)
    assert (tmp_path / "saved.vocab").read_bytes() == bytes(img)
    bad = b"/nonexistent-dir/x.vocab"
    assert c.call(12, 0, bytes([len(bad)]) + bad) == (12, 0)
    assert c.call(12, 4, bytes([len(out)]) + out) == (10, 0)
    c.close()
