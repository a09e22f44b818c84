cd /root/repo
cat > /tmp/h2h.py <<'PY'
import time, numpy as np, sys, os, ctypes as C
sys.path.insert(0, '.')
import tokenmonster_amd as tm
from tokenmonster_amd import synth, _native as N
img = synth.config_vocab("englishcode-32000-consistent")
v = tm.Vocab(img)
raw, roffs = synth.synth_corpus(synth.ENGLISHCODE, 1024 << 20, seed=0x434F5250 + 2)
mode = os.environ.get("MODE", "plain")
if mode in ("batch", "batchfree"):
    text, offs = synth.normalize_batch(raw, roffs, 2, 1)
    batch = C.c_void_p()
    N.check(N.lib.tm_batch_create(v.handle, int(text.size) + (1 << 20), roffs.size - 1, C.byref(batch)))
    N.check(N.lib.tm_batch_upload_raw(batch, N.ptr(raw), N.ptr(roffs), roffs.size - 1))
    for _ in range(3):
        N.check(N.lib.tm_batch_normalize(batch, None)); N.check(N.lib.tm_batch_run(batch, None))
    nt = C.c_uint64(); N.check(N.lib.tm_batch_totals(batch, C.byref(nt), None))
    if mode == "batchfree": N.lib.tm_batch_free(batch)
pin = tm.PinnedBuffer(raw.size); pin.array[:] = raw
pout = tm.PinnedBuffer(raw.size)
for lanes, chunk in ((4, 32 << 20),):
    v.tokenize_pipeline(pin.array, roffs, raw=True, chunk_bytes=chunk, lanes=lanes, out=pout.array)
    t0 = time.perf_counter()
    for _ in range(3): v.tokenize_pipeline(pin.array, roffs, raw=True, chunk_bytes=chunk, lanes=lanes, out=pout.array)
    dt = (time.perf_counter() - t0) / 3
    print("MODE=%s lanes %d chunk %d MiB: %.2f ms = %.2f GB/s" % (mode, lanes, chunk >> 20, dt * 1e3, raw.size / dt / 1e9))
PY
for m in plain batch batchfree; do MODE=$m python /tmp/h2h.py 2>&1 | grep MODE; done
