"""tools/pipeline_stress_fresh.py [iterations] — the four variants of tests/test_gpu_host_api.py::test_pipeline_equals_single_batch with a FRESH vocabulary
(new lanes, new workspaces) every time, over and over (development aid: races in the first calls of a pipeline)."""
import os, sys, time, gc
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import synth
n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 100
img = synth.synth_vocab(synth.ENGLISHCODE, 6000, capcode=2, norm_flag=1, level=3, seed=0x484F5354)
raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 3_000_000, seed=71)
text, offs = synth.normalize_batch(raw, roffs, 2, 1)
bad = 0
t0 = time.time()
for it in range(n_iter):
    for raw_mode in (False, True):
        for pinned in (False, True):
            v = tm.Vocab(img)
            ids, toff, miss = v.tokenize_packed(text, offs)
            src, soff = (raw, roffs) if raw_mode else (text, offs)
            out = None; keep = []
            if pinned:
                pin = tm.PinnedBuffer(src.size); pin.array[:] = src
                pout = tm.PinnedBuffer(2 * ids.size + 64)
                keep += [pin, pout]; src, out = pin.array, pout.array
            for chunk, lanes in ((200_000, 3), (40_000, 4), (1 << 30, 1)):
                blob, boff, bmiss, enc, st = v.tokenize_pipeline(src, soff, raw=raw_mode, chunk_bytes=chunk, lanes=lanes, out=out)
                b = np.asarray(blob).reshape(-1, 2).astype(np.uint32)
                got = b[:, 0] | (b[:, 1] << 8)
                ok = enc == 2 and (bmiss == miss).all() and (boff == toff * np.uint64(2)).all() and got.size == ids.size and (got == ids).all()
                if not ok:
                    bad += 1
                    k = int(np.argmax(got[: min(got.size, ids.size)] != ids[: min(got.size, ids.size)])) if got.size else -1
                    d = int(np.searchsorted(toff, k, side="right") - 1)
                    print("iteration %d raw %s pinned %s chunk %d lanes %d: DIFFERS: %d ids (expected %d), first difference at id %d = document %d of %d (doc bytes %d..%d), offsets equal %s, missing equal %s, chunks %d"
                          % (it, raw_mode, pinned, chunk, lanes, got.size, ids.size, k, d, toff.size - 1, int(soff[d]), int(soff[d + 1]), bool((boff == toff * np.uint64(2)).all()), bool((bmiss == miss).all()), st["chunks"]), flush=True)
            blob, boff, _, enc, _ = v.tokenize_pipeline(src, soff, raw=raw_mode, encoding_length=4, chunk_bytes=300_000, out=np.empty(16, np.uint8))
            del keep, v
print("stress2: %d iterations, %d differences, %.0f s" % (n_iter, bad, time.time() - t0))
