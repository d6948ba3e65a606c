"""GPU-box probe: the j1800 golden (stereo half hour, 25 jumps) -- where does the pass-2 path differ from the reference's, with the
tree LP and with the reference's linprog call?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import cases
from describealign_amd import _native, align as A
name = sys.argv[1] if len(sys.argv) > 1 else "j1800"
g = np.load(os.path.join(ROOT, "tests", "golden", f"align_{name}.npz"))
pair = cases.align_case(name)
for prec in (_native.PREC_F32, _native.PREC_BF16):
  c = _native.Context(0, prec)
  vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
  for tree in ("1", "0"):
    os.environ["DALIGN_LP_TREE"] = tree
    tm = {}
    x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=c, timings=tm)
    path = np.asarray(path); want20 = g["path20"]
    frame = np.rint(path[:, 1] * 210.0).astype(np.int64); want_frame = np.rint(want20[:, 1] * 210.0).astype(np.int64)
    at = np.minimum(np.searchsorted(frame, want_frame), len(path) - 1)
    got = path[at]
    same = (frame[at] == want_frame) & (np.abs(got[:, 0] - want20[:, 0]) < 2e-4)
    bad = np.flatnonzero(~same)
    print("prec", prec, "tree", tree, tm.get("lp_method"), "nodes", len(x), len(g["x"]), "max node diff %.2e" % max(np.max(np.abs(x - g["x"])), np.max(np.abs(y - g["y"]))),
          "rows", len(path), int(g["path_rows"]), "same %.5f" % same.mean(), "bad rows at audio s:", np.round(want20[bad[:6], 1], 2), "...", np.round(want20[bad[-3:], 1], 2) if len(bad) else "",
          "dv(ms)", np.round(1e3 * (got[bad[:6], 0] - want20[bad[:6], 0]), 3), "clusters", len(np.unique(path[:, 2])), len(np.unique(want20[:, 2])), flush=True)
  c.close()
