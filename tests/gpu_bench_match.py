"""Kernel-level timing on the GPU box (not a test): feature + match kernels on the cfg1 pair."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from describealign_amd import _native, synth
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1320.0
ch = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = sys.argv[3] if len(sys.argv) > 3 else None          # "f32" / "bf16": just that precision
pair = synth.make_pair(5, secs, n_jumps=10, first_gap=200.0, channels=ch)
ref = None
for prec, name, peak in ((_native.PREC_F32, "f32", 157.3), (_native.PREC_BF16, "bf16", 2500.0)):
  if only and name != only:
    continue
  c = _native.Context(0, prec)
  c.pcm_upload(0, pair.video); c.pcm_upload(1, pair.audio)
  fts = []
  for rep in range(3):
    vf = c.features_resident(0); s0 = c.stats()
    af = c.features_resident(1); s1 = c.stats()
    fts.append(((s0["features_bytes"] + s1["features_bytes"]) / ((s0["features_ms"] + s1["features_ms"]) * 1e-3) / 1e9,
                s0["features_ms"] + s1["features_ms"]))
  best = None
  for rep in range(3):
    mi, mv, mq = c.match(vf, af)
    st = c.stats()
    tf = st["gemm_flops"] / (st["gemm_ms"] * 1e-3) / 1e12
    if best is None or st["gemm_ms"] < best["gemm_ms"]:
      best = dict(st, tflops=tf)
  key = set(zip(mi.tolist(), mv.tolist()))
  if ref is None: ref = key
  print(json.dumps(dict(prec=name, feat_GBs=round(max(f[0] for f in fts), 1), feat_ms=round(min(f[1] for f in fts), 4),
                        gemm_ms=round(best["gemm_ms"], 3), tflops=round(best["tflops"], 2), frac=round(best["tflops"] / peak, 4),
                        pairs=best["gemm_pairs"], survivors=best["survivors"], matches=best["matches"],
                        verify_ms=round(best["verify_ms"], 3), prep_ms=round(best["prep_ms"], 3), same_as_f32=(key == ref))))
  c.close()
