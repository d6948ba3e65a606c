"""Diagnostic run on the GPU box: prints stage-by-stage numbers (not a test)."""
import os, sys, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from oracle import dalign_oracle as O
from describealign_amd import _native, synth
from describealign_amd import align as A

ctx = _native.Context(0, _native.PREC_F32)
g = np.load(os.path.join(ROOT, "tests/golden/features.npz"))
for name in cases.FEATURE_CLIPS:
  rows = ctx.features(cases.feature_clip(name))
  for k, f in enumerate(rows):
    r = g[f"{name}.f{k}"].astype(np.float64)
    if f.shape != r.shape:
      print("SHAPE", name, k, f.shape, r.shape); continue
    d = np.abs(f - r)
    print(f"feat {name:12s} row{k} maxabs {d.max() if d.size else 0:.3g} maxrel {(d/np.maximum(np.abs(r),1e-2)).max() if d.size else 0:.3g}")
ga = np.load(os.path.join(ROOT, "tests/golden/align_a40.npz"))
vf = [ga[f"vf{k}"] for k in range(5)]; af = [ga[f"af{k}"] for k in range(5)]
for prec in (_native.PREC_F32, _native.PREC_BF16):
  c = _native.Context(0, prec)
  t = time.time(); mi, mv, mq = c.match(vf, af); dt = time.time() - t
  got = set(zip(mi.tolist(), mv.tolist())); want = set(zip(ga["m_i"].tolist(), ga["m_v"].tolist()))
  print("match prec", prec, "n", len(mi), "want", len(want), "missing", len(want - got), "extra", len(got - want), "t", dt, c.stats())
  c.close()
tm = {}
x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx, timings=tm)
print("a40 align", x, y, sim, med, {k: v for k, v in tm.items() if k != "device"})
for name in ("e180", "e1320"):
  pair = cases.align_case(name)
  gg = np.load(os.path.join(ROOT, f"tests/golden/align_{name}.npz"))
  t = time.time(); vf2 = ctx.features(pair.video, 0); af2 = ctx.features(pair.audio, 1); tf = time.time() - t
  tm = {}
  x, y, sim, path, med = A.align(vf2, af2, vf2[0], af2[0], ctx=ctx, timings=tm)
  print(name, "feat_s", tf, "max|dx|", np.abs(x - gg["x"]).max() if len(x) == len(gg["x"]) else ("nodes", len(x), len(gg["x"])),
        "sim", sim, float(gg["sim"]), "med", med)
  print("   ", json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in tm.items() if k != "device"}))
  print("   ", json.dumps({k: round(v, 3) for k, v in tm["device"].items()}))
