cd /root/repo
cat > /tmp/h2h.py <<'PY'
import time, numpy as np, sys, os
sys.path.insert(0, '.')
if os.environ.get("WITH_TORCH"):
    import torch
    torch.cuda.set_device(0); torch.zeros(4, device="cuda")
import tokenmonster_amd as tm
from tokenmonster_amd import synth
img = synth.config_vocab("englishcode-32000-consistent")
v = tm.Vocab(img)
raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 1024 << 20, seed=0x434F5250 + 2)
pin = tm.PinnedBuffer(raw.size); pin.array[:] = raw
pout = tm.PinnedBuffer(int(os.environ.get("OUTSZ", raw.size)))
for lanes, chunk in ((3, 64 << 20), (4, 32 << 20)):
    v.tokenize_pipeline(pin.array, roffs, raw=True, chunk_bytes=chunk, lanes=lanes, out=pout.array)
    t0 = time.perf_counter()
    for _ in range(3): v.tokenize_pipeline(pin.array, roffs, raw=True, chunk_bytes=chunk, lanes=lanes, out=pout.array)
    dt = (time.perf_counter() - t0) / 3
    print("TORCH=%s OUTSZ=%s lanes %d chunk %d MiB: %.2f ms = %.2f GB/s" % (os.environ.get("WITH_TORCH"), os.environ.get("OUTSZ"), lanes, chunk >> 20, dt * 1e3, raw.size / dt / 1e9))
PY
python /tmp/h2h.py 2>&1 | grep TORCH
WITH_TORCH=1 python /tmp/h2h.py 2>&1 | grep TORCH
OUTSZ=1100000000 python /tmp/h2h.py 2>&1 | grep TORCH
