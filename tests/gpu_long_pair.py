"""Scale check on the GPU box (not a test): one long mono pair through the whole path.

  python tests/gpu_long_pair.py [seconds]      # default 14400 (4 h)
"""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from describealign_amd import _native, synth, align as A
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 14400.0
t0 = time.perf_counter()
pair = synth.make_pair(13, secs, n_jumps=20, first_gap=300.0, channels=1)
print("generated in %.1f s" % (time.perf_counter() - t0), pair.video.shape, pair.audio.shape, flush=True)
c = _native.Context(0, _native.PREC_BF16)
t0 = time.perf_counter()
vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
tm = {}
x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=c, timings=tm)
el = time.perf_counter() - t0
offs = x - y
truth = [pair.true_offset_at(float(t)) for t in y[::2] + 0.5]
print("\naligned %.0f s pair in %.1f s (%.0fx real time); nodes %d; sim %.1f" % (secs, el, secs / el, len(x), sim))
print("timings", {k: (round(v, 2) if isinstance(v, float) else v) for k, v in tm.items() if not isinstance(v, dict)})
print("device", {k: round(v, 1) for k, v in tm.get("device", {}).items()})
err = max(abs(o - t) for o, t in zip(offs[::2], truth))
print("max |offset error| vs injected: %.2f ms" % (err * 1e3))
