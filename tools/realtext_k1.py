#!/usr/bin/env python3
"""tools/realtext_k1.py [--mbytes 256] — ON THE GPU BOX: k_match_branch on REAL text (the documents of tests/golden/realtext.json.gz: the
reference tree's own prose and code) with the two vocabularies made from yaml_guide/gpt2.json, before and after tm_vocab_tune.

The files are split in two halves (every other file); tm_vocab_tune lays the tables out by use on half A, and the kernels are timed on half B
repeated to --mbytes of normalized text (documents of one file each).  The first non-synthetic datum for "real text is more concentrated than
the Zipf word soup of tm_synth.cpp": per-kernel milliseconds (HIP events, tm_batch_run_timed), ids md5 before == after the tune.  torch-free."""
import argparse
import base64
import ctypes as C
import hashlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def timed(N, v, text, offs, reps):
    import numpy as np
    nd = offs.size - 1
    batch = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, int(text.size) + (1 << 20), nd, C.byref(batch)))
    N.check(N.lib.tm_batch_upload(batch, N.ptr(text), N.ptr(offs), nd))
    ms = (C.c_float * N.TM_NUM_KERNELS)()
    acc = np.zeros(N.TM_NUM_KERNELS)
    N.check(N.lib.tm_batch_run(batch, None))
    for _ in range(reps):
        N.check(N.lib.tm_batch_run_timed(batch, None, ms))
        acc += np.array(list(ms))
    acc /= reps
    ntok, nmiss = C.c_uint64(), C.c_uint64()
    N.check(N.lib.tm_batch_totals(batch, C.byref(ntok), C.byref(nmiss)))
    ids = np.empty(max(int(ntok.value), 1), dtype=np.uint32)
    toff = np.empty(nd + 1, dtype=np.uint64)
    N.check(N.lib.tm_batch_download(batch, N.ptr(ids), int(ntok.value), N.ptr(toff), None))
    N.lib.tm_batch_free(batch)
    names = [N.lib.tm_kernel_name(k).decode() for k in range(N.TM_NUM_KERNELS)]
    return dict(zip(names, acc)), int(ntok.value), hashlib.md5(ids[: int(ntok.value)].tobytes() + toff.tobytes()).hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mbytes", type=int, default=256)
    ap.add_argument("--reps", type=int, default=4)
    a = ap.parse_args()
    import numpy as np
    import tokenmonster_amd as tm
    from tokenmonster_amd import _native as N
    from conftest import GOLDEN_DIR, load_golden, realtext_fixture
    N.check(N.lib.tm_set_device(0))
    g, docs = realtext_fixture()
    if docs is None:
        raise SystemExit("tests/golden/_realtext_docs.bin.gz is not here (python tests/golden/make_realtext_golden.py where /root/reference is)")
    files = [d for d, n in zip(docs, g["names"]) if "[" not in n]
    half_a, half_b = files[0::2], files[1::2]
    print("real text: %d files, half A (tune) %d bytes, half B (timed) %d bytes" % (len(files), sum(map(len, half_a)), sum(map(len, half_b))))
    for key in ("gpt2", "gpt2-capcode2-nfd"):
        img = base64.b64decode(load_golden(os.path.join(GOLDEN_DIR, "gpt2_vocab.json.gz"))["vocab_b64"] if key == "gpt2" else g["vocab_b_b64"])
        v = tm.Vocab(img)
        nb = [v.normalize(d) for d in half_b]
        na = b"".join(v.normalize(d) for d in half_a)
        per = sum(map(len, nb))
        reps = max(1, (a.mbytes << 20) // per)
        text, offs = tm.pack_documents(nb * reps)
        t0 = time.time()
        before, ntok, md5a = timed(N, v, text, offs, a.reps)
        v.tune(np.frombuffer(na, dtype=np.uint8))
        after, _, md5b = timed(N, v, text, offs, a.reps)
        mib = text.size / (1 << 20)
        print("%-18s %d ids, %.0f MiB normalized (%d x half B, %d documents), %.2f bytes per token" % (key, v.n_ids(), mib, reps, offs.size - 1, text.size / max(ntok, 1)))
        print("   as loaded        : " + " ".join("%s %.3f" % kv for kv in before.items()) + "   -> match_branch %.2f ms per GiB" % (before["match_branch"] * 1024 / mib))
        print("   tuned on half A  : " + " ".join("%s %.3f" % kv for kv in after.items()) + "   -> match_branch %.2f ms per GiB (%+.1f %%)  ids %s  (%.0f s)" % (
            after["match_branch"] * 1024 / mib, 100 * (after["match_branch"] / before["match_branch"] - 1), "identical" if md5a == md5b else "DIFFER", time.time() - t0))


if __name__ == "__main__":
    main()
