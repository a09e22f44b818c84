"""tools/h2h_sweep.py [lanes:chunkMiB ...] — the host-to-host pipeline (tm_tokenize_pipeline) over lanes x chunk size on one GPU (development aid).
Warm-up first (a fresh process needs ~6 calls), then best / median of 8 calls per setting.  The ring's knobs come from the environment
(TM_RING=0: the lanes' form; TM_RING_SLOTS, TM_RING_STREAMS, TM_RING_SLACK; GPU_MAX_HW_QUEUES)."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tokenmonster_amd as tm
from tokenmonster_amd import synth
cfg = os.environ.get("TM_SWEEP_CFG", "englishcode-32000-consistent")
kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[cfg]
img = synth.config_vocab(cfg)
if "TM_SWEEP_FLAG" in os.environ:       # the same vocabulary with another normalization flag byte (e.g. 186: lowercase collapse trim quotemarks unixlines - the filter pass, the lanes' form)
    img = bytes(img[:2]) + bytes([int(os.environ["TM_SWEEP_FLAG"])]) + bytes(img[3:])
v = tm.Vocab(img)
raw, roffs = synth.synth_corpus(kind, 1024 << 20, seed=0x434F5250 + 2)
pin_in = tm.PinnedBuffer(raw.size); pin_in.array[:] = raw
pin_out = tm.PinnedBuffer(raw.size + 4096)
settings = [tuple(int(x) for x in a.split(":")) for a in sys.argv[1:]] or [(4, 32)]
env = " ".join("%s=%s" % (k, os.environ[k]) for k in ("TM_SWEEP_FLAG", "TM_RING", "TM_RING_SLOTS", "TM_RING_STREAMS", "TM_RING_SLACK", "TM_RING_FIRST_KIB", "TM_RING_RAMP", "GPU_MAX_HW_QUEUES") if k in os.environ)
for lanes, chunk in settings:
    for _ in range(6):
        v.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=chunk << 20, lanes=lanes, out=pin_out.array)
    ts = []
    for _ in range(8):
        t0 = time.perf_counter(); r = v.tokenize_pipeline(pin_in.array, roffs, raw=True, chunk_bytes=chunk << 20, lanes=lanes, out=pin_out.array); ts.append((time.perf_counter() - t0) * 1e3)
    st = r[4]
    ts.sort()
    print("[%s] lanes %2d chunk %3d MiB: best %.2f  median %.2f  worst %.2f ms  %.2f GB/s (median)  ring %d exact %d chunks %d" % (
        env, lanes, chunk, ts[0], ts[4], ts[-1], raw.size / ts[4] / 1e6, st["ring"], st["ring_exact_chunks"], st["chunks"]), flush=True)
