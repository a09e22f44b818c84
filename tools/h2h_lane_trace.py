import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import synth
cfg = "englishcode-32000-consistent"
kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[cfg]
v = tm.Vocab(synth.config_vocab(cfg))
raw, roffs = synth.synth_corpus(kind, 1024 << 20, seed=0x434F5250 + 2)
pin_in = tm.PinnedBuffer(raw.size); pin_in.array[:] = raw
pin_out = tm.PinnedBuffer(raw.size + 4096)
for i in range(3):
    v.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=32 << 20, lanes=4, out=pin_out.array)
print("=== traced pass", file=sys.stderr, flush=True)
os.environ["TM_TRACE_ON"] = "1"
t0 = time.perf_counter(); v.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=32 << 20, lanes=4, out=pin_out.array); print("pass %.2f ms" % ((time.perf_counter() - t0) * 1e3))
