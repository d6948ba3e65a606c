"""GPU-box script (not a pytest): AlignPipeline over a long batch of random pairs (lengths 40 s .. 20 min, mono / stereo, 0-3
offset jumps) against align() on the same pairs, pair by pair: node times, similarity, slope and pass-2 path identical.

  python tests/gpu_stress_pipeline.py [pairs] [lp_workers]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from describealign_amd import _native, synth  # noqa: E402
from describealign_amd import align as A  # noqa: E402


def main():
  n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
  workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
  rng = np.random.default_rng(77)
  ctx = _native.Context(0, _native.PREC_BF16)
  t0 = time.time()
  feats = []
  for k in range(n_pairs):
    sec = float(10 ** rng.uniform(1.6, 3.08))
    pair = synth.make_pair(int(rng.integers(1 << 30)), sec, n_jumps=int(rng.integers(0, 4)), first_gap=float(rng.uniform(5, max(6.0, sec / 5))),
                           channels=int(rng.integers(1, 3)))
    feats.append((ctx.features(pair.video, 0), ctx.features(pair.audio, 1)))
  want, keep = [], []
  for vf, af in feats:
    try:
      want.append(A.align(vf, af, vf[0], af[0], ctx=ctx)); keep.append((vf, af))
    except RuntimeError:                              # a pair align() refuses (too few matches): not part of the batch
      pass
  feats = keep
  t1 = time.time()
  bad = 0
  with A.AlignPipeline(ctx, lp_workers=workers) as pipe:
    for k, g in enumerate(pipe.run(feats, expected=len(feats))):
      w = want[k]
      ok = np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]) and g[2] == w[2] and g[4] == w[4] and np.array_equal(g[3], w[3])
      if not ok:
        bad += 1; print("MISMATCH", k, len(feats[k][0][0]), flush=True)
  print(f"pairs {len(feats)}, mismatches {bad}; sequential {t1 - t0:.0f} s, pipeline {time.time() - t1:.0f} s", flush=True)
  ctx.close()
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
