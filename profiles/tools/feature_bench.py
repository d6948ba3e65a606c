import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from describealign_amd import _native, synth
ctx = _native.Context(0, _native.PREC_BF16)
pair = synth.make_pair(5, 7200.0, n_jumps=10, first_gap=200.0, channels=2)
for side, pcm in ((0, pair.video), (1, pair.audio)):
  ms = []
  for r in range(12):
    f = ctx.features(pcm, side)
    ms.append(ctx.stats()["features_ms"])
  ms = sorted(ms[2:])
  nbytes = ctx.stats()["features_bytes"]
  print(side, pcm.shape, "feat_ms median %.4f min %.4f -> %.0f GB/s (median), %.0f (best)" % (ms[len(ms)//2], ms[0], nbytes / ms[len(ms)//2] / 1e6, nbytes / ms[0] / 1e6))
