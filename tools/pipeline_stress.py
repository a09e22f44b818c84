"""tools/pipeline_stress.py [iterations] — tm_tokenize_pipeline (raw text, pinned buffers, small chunks, several lanes) over and over against one
tm_tokenize_batch of the same corpus: a development aid for races between the lanes of the host-to-host pipeline."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import synth

n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 200
img = synth.synth_vocab(synth.ENGLISHCODE, 6000, capcode=2, norm_flag=1, level=3, seed=0x484F5354)
raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 3_000_000, seed=71)
text, offs = synth.normalize_batch(raw, roffs, 2, 1)
v = tm.Vocab(img)
ids, toff, miss = v.tokenize_packed(text, offs)
pin = tm.PinnedBuffer(raw.size); pin.array[:] = raw
pout = tm.PinnedBuffer(2 * ids.size + 64)
bad = 0
t0 = time.time()
for it in range(n_iter):
    for chunk, lanes in ((200_000, 3), (40_000, 4), (25_000, 6), (1 << 30, 1)):
        pout.array[:] = 0xEE
        blob, boff, bmiss, enc, st = v.tokenize_pipeline(pin.array, roffs, raw=True, chunk_bytes=chunk, lanes=lanes, out=pout.array)
        b = np.asarray(blob).reshape(-1, 2).astype(np.uint32)
        got = b[:, 0] | (b[:, 1] << 8)
        ok = enc == 2 and (bmiss == miss).all() and (boff == toff * np.uint64(2)).all() and got.size == ids.size and (got == ids).all()
        if not ok:
            bad += 1
            k = int(np.argmax(got[: ids.size] != ids[: got.size])) if got.size else -1
            d = int(np.searchsorted(toff, k, side="right") - 1)
            print("iteration %d chunk %d lanes %d: DIFFERS: enc %d, %d ids (expected %d), first difference at id %d = document %d (of %d), offsets equal %s, missing equal %s, chunks %d"
                  % (it, chunk, lanes, enc, got.size, ids.size, k, d, toff.size - 1, bool((boff == toff * np.uint64(2)).all()), bool((bmiss == miss).all()), st["chunks"]), flush=True)
print("pipeline stress: %d iterations x 4 configurations, %d differences, %.0f s" % (n_iter, bad, time.time() - t0))
