"""tools/h2h_lane_trace.py [passes] — development aid: TM_TRACE=1 python tools/h2h_lane_trace.py 2> trace.txt prints the pass times; the [pipe] lines of
every pass (one per chunk: worker, size, start, upload, compute, order wait, download, end) go to stderr between '=== pass' markers"""
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
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    print("=== pass %d" % i, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    v.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=32 << 20, lanes=4, out=pin_out.array)
    dt = (time.perf_counter() - t0) * 1e3
    print("=== pass %d took %.2f ms" % (i, dt), file=sys.stderr, flush=True)
    print("pass %d: %.2f ms" % (i, dt), flush=True)
