"""Development aid: per-phase wall cycles of k_match_branch.

Needs a library built with the phase timers:  TM_EXTRA_FLAGS=-DTM_PHASE_TIMERS python tokenmonster_amd/build.py --force
(never the product build).  Prints, per segment, the cycles one wavefront spends in each phase and the loop counts."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.environ.get("TM_PKG_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # TM_PKG_ROOT: a copy of the package with the phase-timer library
from tokenmonster_amd import _native as N, synth, vocab as V   # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "englishcode-32000-consistent"
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
kind, vsize, capcode, norm_flag, level, vseed = synth.CONFIGS[cfg]
img = synth.config_vocab(cfg)
voc = V.Vocab(img)
raw, roffs = synth.synth_corpus(kind, mb << 20, seed=0x434F5250 + 2)
text, offs = synth.normalize_batch(raw, roffs, capcode, norm_flag)
lib = N.lib
lib.tm_debug_phases.argtypes = [C.c_void_p, C.c_int]
batch = C.c_void_p()
N.check(lib.tm_batch_create(voc.handle, text.size, offs.size - 1, C.byref(batch)))
N.check(lib.tm_batch_upload(batch, N.ptr(text), N.ptr(offs), offs.size - 1))
N.check(lib.tm_batch_run(batch, None))
out = (C.c_ulonglong * 32)()
lib.tm_debug_phases(None, 1)
reps = 3
ms = (C.c_float * N.TM_NUM_KERNELS)()
for _ in range(reps):
    N.check(lib.tm_batch_run_timed(batch, None, ms))
lib.tm_debug_phases(out, 0)
v = np.array(list(out), dtype=np.float64)
nseg = v[12]
names = ["stage+zero", "B: rows + T(p,0)", "A1 main loop", "barrier wait", "A2", "A3", "B: T(p,1) + stores", "C"]
tot = v[:8].sum()
print("segments %d (x%d passes), k_match_branch %.3f ms" % (nseg / reps, reps, ms[1]))
for i, n in enumerate(names):
    print("%-20s %9.0f cycles/segment  %5.1f %%" % (n, v[i] / nseg, 100 * v[i] / tot))
print("total %.0f cycles/segment" % (tot / nseg))
print("per segment: A1 loop rounds %.1f, walks handed to the task list %.1f, their rounds behind the loop %.1f, A3 + task probe rounds %.1f, C rounds %.1f, A3 tasks %.1f, (p,1) states %.1f"
      % (v[8] / nseg, v[9] / nseg, v[10] / nseg, v[11] / nseg, v[13] / nseg, v[14] / nseg, v[15] / nseg))
