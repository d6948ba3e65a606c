"""GPU-box helper (not a pytest): run the device stages of the alignment path on synthetic pairs and save the
LP's input -- the pass-1 fit points (describealign.py:743-767) -- so that the host LP work (align.solve_trend_lp,
its windowed decomposition and the certificates) can be developed and measured on any CPU.

  python tests/gpu_dump_fit_points.py OUTDIR case [case ...]

case = name:seed:seconds:n_jumps:first_gap:channels:precision[:rate_change]
Writes OUTDIR/<name>.npz with fx, fy (float64 fit points), the pass-1 path kept by the continuity filter (int32 x, y),
and the frame counts.
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  out_dir = sys.argv[1]
  os.makedirs(out_dir, exist_ok=True)
  import torch  # noqa: F401  (before libdalign.so)
  from describealign_amd import _native, synth
  from describealign_amd import align as A
  ctxs = {}
  for spec in sys.argv[2:]:
    parts = spec.split(":")
    name, seed, secs, nj, gap, ch, prec = parts[0], int(parts[1]), float(parts[2]), int(parts[3]), float(parts[4]), int(parts[5]), parts[6]
    rate = float(parts[7]) if len(parts) > 7 else 0.0
    p = _native.PREC_F32 if prec == "f32" else _native.PREC_BF16
    if p not in ctxs:
      ctxs[p] = _native.Context(0, p)
    ctx = ctxs[p]
    t0 = time.perf_counter()
    pair = synth.make_pair(seed, secs, n_jumps=nj, first_gap=gap, channels=ch, rate_change=rate)
    t1 = time.perf_counter()
    vf = ctx.features(pair.video, _native.SIDE_VIDEO); af = ctx.features(pair.audio, _native.SIDE_AUDIO)
    tm = {}
    n_ve, n_ae = len(vf[0]), len(af[0])
    A._stage_gpu_match(ctx, vf, af, _native.MATCH_HASHED, tm)
    px, py = ctx.chain_resident(min_len=A.min_path_length(n_ve, n_ae))
    x = np.asarray(px).astype(np.int64); y = np.asarray(py).astype(np.int64)
    keep = A.continuity_error(x, y) < 3
    x, y = x[keep], y[keep]
    fx, fy = A.compress_path(x, y)
    t2 = time.perf_counter()
    np.savez_compressed(os.path.join(out_dir, name + ".npz"), fx=fx, fy=fy, x=x.astype(np.int32), y=y.astype(np.int32),
                        n_ve=n_ve, n_ae=n_ae, jump_video_times=np.array(pair.jump_video_times), jump_lengths=np.array(pair.jump_lengths))
    print(json.dumps(dict(name=name, fit_points=len(fx), path=len(px), kept=len(x), gen_s=round(t1 - t0, 1), gpu_s=round(t2 - t1, 1))), flush=True)
    del pair, vf, af


if __name__ == "__main__":
  main()
