"""tools/h2h_sweep.py — the host-to-host pipeline (tm_tokenize_pipeline) over lanes x chunk size on one GPU (development aid)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import synth
cfg = "englishcode-32000-consistent"
kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[cfg]
v = tm.Vocab(synth.config_vocab(cfg))
raw, roffs = synth.synth_corpus(kind, 1024 << 20, seed=0x434F5250 + 2)
pin_in = tm.PinnedBuffer(raw.size); pin_in.array[:] = raw
pin_out = tm.PinnedBuffer(raw.size + 4096)
for lanes, chunk in ((4, 32), (2, 64), (2, 48), (2, 32), (3, 32), (2, 96), (1, 64), (4, 32), (2, 64)):
    v.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=chunk << 20, lanes=lanes, out=pin_out.array)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); v.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=chunk << 20, lanes=lanes, out=pin_out.array); best = min(best, time.perf_counter() - t0)
    print("lanes %2d chunk %3d MiB: %.2f ms  %.2f GB/s" % (lanes, chunk, best * 1e3, raw.size / best / 1e9), flush=True)
