"""GPU-box script (not a pytest): the resident path of the chain DP -- ranks from the video frames that have a match, audio
rows counted on the device, hand-over buffers from the context's pool -- against the host utility on the fetched match
list, over random synthetic pairs of random lengths, both GEMM precisions, several DPs in flight.

  python tests/gpu_stress_resident.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from describealign_amd import _native, synth  # noqa: E402


def main():
  budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
  rng = np.random.default_rng(404)
  ctxs = {p: _native.Context(0, p) for p in (_native.PREC_F32, _native.PREC_BF16)}
  t0 = time.time(); done = 0; bad = 0; tot = 0
  pending = []                                            # (ctx, ticket, expected path)
  try:
    while time.time() - t0 < budget:
      sec = float(10 ** rng.uniform(1.5, 3.0)); prec = [_native.PREC_F32, _native.PREC_BF16][int(rng.integers(2))]
      pair = synth.make_pair(int(rng.integers(1 << 30)), sec, n_jumps=int(rng.integers(0, 4)), first_gap=float(rng.uniform(0, sec / 4)),
                             channels=int(rng.integers(1, 3)))
      ctx = ctxs[prec]
      vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
      mi, mv, mq = ctx.match(vf, af)
      if len(mi) == 0:
        continue
      want = _native.chain_host(mi, mv, mq)
      pending.append((ctx, ctx.chain_begin(), want, sec, prec))
      if len(pending) >= 3 or rng.integers(3) == 0:
        for c, t, w, s, p in pending:
          gi, gv = c.chain_finish(t)
          ok = len(gi) == len(w[0]) and np.array_equal(gi, w[0]) and np.array_equal(gv, w[1])
          done += 1; tot += len(w[0])
          if not ok:
            bad += 1
            print("MISMATCH", s, p, len(gi), len(w[0]), flush=True)
        pending = []
  finally:
    for c in ctxs.values():
      c.close()
  print(f"pairs {done}, path points {tot}, mismatches {bad}, {time.time() - t0:.0f} s", flush=True)
  return 1 if bad else 0


if __name__ == "__main__":
  sys.exit(main())
