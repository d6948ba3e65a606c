"""GPU-box script: how the matches of a pair fall into the (column, batch) cells of the column-pipelined chain DP, and
what the dependency structure (a cell waits for its left neighbour and its predecessor in the column) makes of it.

  python profiles/tools/chain_cells.py [seconds] [n_cols]

Cost model of a cell with m matches: 4 us + 0.06 us x m + 2.5 us x ceil(m / 64) (from the diagnostic stamps); prints the
longest dependent path (last-passage time) beside the simple bounds, for equal-width and for weight-balanced columns."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from describealign_amd import _native, synth  # noqa: E402

sec = float(sys.argv[1]) if len(sys.argv) > 1 else 7200.0
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 384
ctx = _native.Context(0, _native.PREC_BF16 if sec >= 3000 else _native.PREC_F32)
pair = synth.make_pair(5, sec, n_jumps=10, first_gap=200.0)
vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
mi, mv, mq = ctx.match(vf, af)
n = len(mi)
rows_u, row = np.unique(mi, return_inverse=True)
ranks_u, rank = np.unique(mv, return_inverse=True)       # the device ranks the listed video rows; close enough here
nr = len(ranks_u); nb = (len(rows_u) + 255) // 256
print(f"matches {n}, rows {len(rows_u)}, ranks {nr}, batches {nb}")
batch = row // 256


def lpp(t):
  """last-passage time over cells t[c, b]: T[c, b] = max(T[c-1, b], T[c, b-1]) + t[c, b]"""
  T = np.zeros(t.shape[1])
  for c in range(t.shape[0]):
    x = 0.0
    tc = t[c]
    for b in range(t.shape[1]):
      x = max(x, T[b]) + tc[b]
      T[b] = x
  return T[-1]


def report(name, col):
  cells = np.bincount(col.astype(np.int64) * nb + batch, minlength=nc * nb).reshape(nc, nb)
  t = 4.0 + 0.06 * cells + 2.5 * np.ceil(cells / 64.0)
  colsum = t.sum(axis=1)
  print(f"{name}: cell matches mean {cells.mean():.0f}, p99 {np.percentile(cells, 99):.0f}, max {cells.max()}; column time mean {colsum.mean()/1e3:.1f} ms, max {colsum.max()/1e3:.1f} ms; "
        f"sum over batches of the slowest cell {t.max(axis=0).sum()/1e3:.1f} ms; uniform-cell bound {(nc + nb) * t.mean()/1e3:.1f} ms; longest dependent path {lpp(t)/1e3:.1f} ms")
  big = np.argwhere(cells > 8 * cells.mean())
  if len(big):
    print(f"   {len(big)} cells above 8x the mean; e.g. (column, batch, matches):", [(int(c), int(b), int(cells[c, b])) for c, b in big[:: max(1, len(big) // 8)][:8]])
  return cells




def balanced(count_per_unit, unit_of_match, n_parts, alpha):
  """parts of equal weight; a unit (rank or row) weighs its matches + alpha average shares: at most (1 + alpha) / alpha x the average units per part"""
  m = len(count_per_unit)
  front = (np.cumsum(count_per_unit) - count_per_unit).astype(np.float64) * m + alpha * np.arange(m) * float(n)
  part = np.minimum(n_parts - 1, (front / ((1.0 + alpha) * n * m) * n_parts).astype(np.int64))
  return part[unit_of_match], part


def model(cells, fixed):
  t = fixed + 0.06 * cells + 2.5 * np.ceil(cells / 64.0)
  return t


hist = np.bincount(rank, minlength=nr)
per_row = np.bincount(row)
print("matches per row: mean %.0f, p50 %.0f, p99 %.0f, max %d" % (per_row.mean(), np.median(per_row), np.percentile(per_row, 99), per_row.max()))
print("matches per rank: mean %.0f, p50 %.0f, p99 %.0f, max %d" % (hist.mean(), np.median(hist), np.percentile(hist, 99), hist.max()))
nrows = len(rows_u)
for ncols in (384, 512, 640):
  col, _ = balanced(hist, rank, ncols, 1.0)
  for label, avg_rows, alpha in (("256 rows fixed", 256, None), ("avg 128 rows, <= 256", 128, 1.0), ("avg 192 rows, <= 256", 192, 3.0),
                                 ("avg 256 rows, <= 512", 256, 1.0), ("avg 64 rows, <= 128", 64, 1.0)):
    nbb = (nrows + avg_rows - 1) // avg_rows
    if alpha is None:
      bat = row // 256
    else:
      bat, part = balanced(per_row, row, nbb, alpha)
      assert np.bincount(part).max() <= round(avg_rows * (1 + alpha) / alpha) + 1, np.bincount(part).max()
    cells = np.bincount(col.astype(np.int64) * nbb + bat, minlength=ncols * nbb).reshape(ncols, nbb)
    for fixed in (4.0, 2.0):
      t = model(cells, fixed)
      print(f"cols {ncols}, batches {label} ({nbb}), fixed {fixed} us: cell max {cells.max()}, column mean {t.sum(axis=1).mean()/1e3:.1f} ms, heaviest cell {t.max():.0f} us, "
            f"uniform bound {(ncols + nbb) * t.mean()/1e3:.1f} ms, longest path {lpp(t)/1e3:.1f} ms")
