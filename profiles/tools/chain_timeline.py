"""GPU-box script: timeline of the column-pipelined chain DP (diagnostic library, `make -C describealign_amd/csrc dbg`).

  DALIGN_LIB=describealign_amd/libdalign_dbg.so python profiles/tools/chain_timeline.py [seconds] [out.txt]

Every column stamps the 100 MHz wall clock at its start and at the end of every k-th batch; this prints when the
columns start, how far a column trails its left neighbour at the same batch (the hand-over skew) and how fast the
last column advances (the steady-state batch time)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sec = float(sys.argv[1]) if len(sys.argv) > 1 else 7200.0
out = os.path.abspath(sys.argv[2]) if len(sys.argv) > 2 else "/tmp/chain_timeline.txt"
os.environ["DALIGN_DEBUG_STAMPS"] = out
if os.path.exists(out):
  os.remove(out)
from describealign_amd import _native, synth  # noqa: E402

ctx = _native.Context(0, _native.PREC_BF16 if sec >= 3000 else _native.PREC_F32)
pair = synth.make_pair(5, sec, n_jumps=10, first_gap=200.0)
vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
for rep in range(2):
  ctx.match_begin(vf, af); n = ctx.match_finish()
  gi, gv = ctx.chain_resident()
st = ctx.stats()
print(f"matches {n}, path {len(gi)}, chain_ms {st['chain_ms']:.2f}, columns {int(st['chain_columns'])}, widest column allowed {int(st['chain_column_width'])}")
if not os.path.exists(out):          # the production library writes no stamps
  sys.exit(0)
t = np.loadtxt(out)
nc = t.shape[0]
start = t[:, 1]; tl = t[:, 2:33]; cnt = t[:, 33:41]
valid = (tl > 0).all(axis=0)
k_last = int(np.nonzero(valid)[0][-1])
print(f"matches {n}, path {len(gi)}, chain_ms {st['chain_ms']:.2f}, columns {nc}, samples 0..{k_last}")
print("column starts (us): first", start[0], "median", np.median(start), "last", start.max())
for k in (0, k_last // 2, k_last):
  skew = np.diff(tl[:, k])
  print(f"sample {k}: column 0 at {tl[0, k]:.0f} us, last column at {tl[-1, k]:.0f} us; per-column skew mean {skew.mean():.2f} us, median {np.median(skew):.2f}, p90 {np.percentile(skew, 90):.2f}, max {skew.max():.2f}")
for c in (0, nc // 2, nc - 1):
  d = np.diff(tl[c, :k_last + 1])
  print(f"column {c}: per-sample interval mean {d.mean():.1f} us (min {d.min():.1f}, max {d.max():.1f}); active counters (us): " + " ".join(f"{x:.0f}" for x in cnt[c]))
