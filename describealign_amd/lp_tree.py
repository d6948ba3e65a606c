"""The reference's L1 trend-fit LP (describealign.py:769-858) solved by HiGHS through a tree of warm starts.

What is solved is the reference's LP, whole, by the HiGHS dual simplex that `scipy.optimize.linprog(method='highs-ds')`
drives (scipy's own build, reached through the binding scipy itself uses, `scipy.optimize._highspy._core`); what changes is
where the simplex STARTS.  From the slack basis HiGHS needs ~4.6 pivots per fit point on this LP and every pivot costs O(n)
(the rows are first differences, so the basis inverse is a cumulative sum: dense): a 2 h pair's 10 282 points take 47 000
pivots of ~130 us.  Here the fit points are cut into leaves of a few hundred points whose sub-LPs (the same rows and columns,
restricted) are solved cold -- the same pivots per point, each ~30 x cheaper -- and neighbouring sub-LPs are merged, four at a
time and the two halves at the root: a merged LP starts from its children's optimal bases with the rows of the cuts between
them left to their logicals (a dual feasible start, see _merge_basis) and needs a few hundred to a few thousand pivots; the
last merge IS the full LP.  Every level is an exact HiGHS solve; the root's model status, primal and dual solution are HiGHS'
own for the reference's LP.

Two details:
  * median_slope is shared by all rows.  Below the root it is held at m_c (sub-LPs with different slopes cannot be stitched);
    the rows are written for m' = median_slope - m_c, so that the root can free m' from its non-basic value 0.  The optimum
    does not depend on m_c, only the number of pivots at the root does: m_c starts from a data estimate and is re-centred level
    by level on the weighted median of the level's own plateau slopes (recentre_level in solve()).
  * devex pricing at every level: steepest-edge weights of a user basis have to be computed from scratch, and with them the
    same merges take 3-4 x the pivots (profiles/r06_lp_decomposition.txt).

Exactness: windows are NOT independent (the optimum inside a window depends on the rest of the file through the slope, the rate
chain's duals and shot levels that persist for hundreds of points), which is why nothing is taken from a sub-LP but a
starting basis.  The root's result is checked against the ORIGINAL LP's optimality conditions (kkt_certificate) by the
caller; any failure, any status other than optimal, any exception -> None, and the caller makes the reference's own call.
"""
from __future__ import annotations

import os
import threading
import time

import numpy as np
import scipy.sparse

LEAF_POINTS = int(os.environ.get("DALIGN_LP_LEAF", "280"))           # 240-320 measured best (profiles/r06_lp_decomposition.txt)
PARALLEL_MIN_POINTS = 6000                                            # fit points from which one LP is worth spreading over helper processes
MIN_POINTS = int(os.environ.get("DALIGN_LP_TREE_MIN", "1000"))       # below this a solve takes 0.2 s either way (1 588 points: 0.59 -> 0.38 s)

_core = None


def available() -> bool:
  """scipy's HiGHS binding with the calls this module needs (scipy >= 1.15)."""
  global _core
  if _core is None:
    try:
      import scipy.optimize._highspy._core as core
      need = ("HighsLp", "HighsBasis", "HighsOptions", "HighsBasisStatus", "HighsModelStatus", "MatrixFormat", "simplex_constants")
      ok = all(hasattr(core, k) for k in need) and all(hasattr(core._Highs, k) for k in ("passModel", "passOptions", "setBasis", "run", "getSolution", "getBasis", "getInfo"))
      _core = core if ok else False
    except Exception:
      _core = False
  return bool(_core)


# ---- the LP of a run of fit points ---------------------------------------------------------------------------------------
def block_sizes(n):
  """Column blocks of the LP on n fit points, in order, and its three row blocks."""
  return (n, n, n - 1, n - 1, n, n, n - 1, n - 1, n - 1, n - 1, n - 2, n - 2, 1), (n - 1, n - 1, n - 2)


def assemble(x, y, jump_cost, slope_shift=0.0):
  """(c, A csc, b, lb, ub) of the trend LP on fit points (x, y) with the given per-segment jump costs -- for the whole
  path this is align.build_trend_lp's LP (same triplets, same floats); for a slice of the path it is that LP's rows and
  columns restricted to the slice.  slope_shift: the rows are written for median_slope - slope_shift."""
  n = len(x)
  dx = np.diff(x); dy = np.diff(y)
  inv = 1.0 / dx
  c = np.concatenate([np.ones(2 * n), jump_cost, jump_cost, np.full(2 * n, 0.01), np.full(2 * n - 2, 3.0),
                      np.full(2 * n - 2, 0.001), np.full(2 * n - 4, 10.0 * 4000), [0.0]])
  o_fe_p, o_fe_m = 0, n
  o_j_p, o_j_m = 2 * n, 3 * n - 1
  o_s_p, o_s_m = 4 * n - 2, 5 * n - 2
  o_sj_p, o_sj_m = 6 * n - 2, 7 * n - 3
  o_rj_p, o_rj_m = 8 * n - 4, 9 * n - 5
  o_rc_p, o_rc_m = 10 * n - 6, 11 * n - 8
  o_med = 12 * n - 10
  r1 = np.arange(n - 1)
  r2 = np.arange(n - 2)
  rows, cols, vals = [], [], []

  def put(r, col, v):
    rows.append(r); cols.append(col); vals.append(np.broadcast_to(v, r.shape))

  put(r1, o_fe_p + r1, -inv); put(r1, o_fe_p + r1 + 1, inv)
  put(r1, o_fe_m + r1, inv); put(r1, o_fe_m + r1 + 1, -inv)
  for op, om in ((o_j_p, o_j_m), (o_sj_p, o_sj_m), (o_rj_p, o_rj_m)):
    put(r1, op + r1, inv); put(r1, om + r1, -inv)
  put(r1, np.full(n - 1, o_med), 1.0)
  b2 = (n - 1) + r1
  put(b2, o_s_p + r1, -1.0); put(b2, o_s_p + r1 + 1, 1.0)
  put(b2, o_s_m + r1, 1.0); put(b2, o_s_m + r1 + 1, -1.0)
  put(b2, o_sj_p + r1, -1.0); put(b2, o_sj_m + r1, 1.0)
  b3 = (2 * n - 2) + r2
  put(b3, o_rj_p + r2, -inv[:-1]); put(b3, o_rj_p + r2 + 1, inv[1:])
  put(b3, o_rj_m + r2, inv[:-1]); put(b3, o_rj_m + r2 + 1, -inv[1:])
  put(b3, o_rc_p + r2, -1.0); put(b3, o_rc_m + r2, 1.0)
  A = scipy.sparse.csc_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                              shape=(3 * n - 4, 12 * n - 9))
  A.sort_indices()
  b = np.concatenate([dy / dx - slope_shift, np.zeros(2 * n - 3)])
  lb = np.zeros(12 * n - 9); ub = np.full(12 * n - 9, np.inf)
  ub[4 * n - 2:6 * n - 2] = 2.0
  lb[-1] = -np.inf
  return c, A, b, lb, ub


def estimate_slope(x, y):
  """A rough median_slope for the sub-LPs: the median of the slopes between fit points ~64 apart whose span holds no jump
  (|dy - dx| of every segment small).  Only the pivot count of the root depends on how good it is."""
  n = len(x)
  g = max(8, min(64, n // 8))
  dx, dy = np.diff(x), np.diff(y)
  s0 = float(np.median(dy / dx))                       # crude: most segments are a few frames long; good enough to tell a jump from a rate
  if not (0.1 < s0 < 10.0):
    s0 = 1.0
  dev = np.abs(dy - s0 * dx)
  bad = np.concatenate([[0], np.cumsum(dev > 8.0)])
  i = np.arange(0, n - g)
  ok = (bad[i + g] - bad[i]) == 0
  if ok.sum() < 8:
    return 1.0
  s = (y[i + g] - y[i])[ok] / (x[i + g] - x[i])[ok]
  m = float(np.median(s))
  return m if 0.1 < m < 10.0 else 1.0


# ---- HiGHS through scipy's binding -----------------------------------------------------------------------------------------
class _Node:
  """One solved sub-LP: the run of fit points [a, e), its solution and its basis (lists of HighsBasisStatus, per block)."""
  __slots__ = ("a", "e", "x", "col_status", "row_status", "row_dual", "col_dual", "pivots", "solver")


def _split(seq, sizes):
  out, at = [], 0
  for s in sizes:
    out.append(seq[at:at + s]); at += s
  return out


def _run(c, A, b, lb, ub, basis, free_slope):
  """One HiGHS dual-simplex solve with scipy.optimize.linprog(method='highs-ds')'s settings.  Returns (x, col_status,
  row_status, row_dual, col_dual, pivots, the solver object) or None unless HiGHS reports the model optimal."""
  H = _core
  n_col, n_row = len(c), len(b)
  inf = H.kHighsInf
  if not free_slope:
    lb = lb.copy(); ub = ub.copy()
    lb[-1] = ub[-1] = 0.0
  h = H._Highs()
  o = H.HighsOptions()
  o.output_flag = False; o.log_to_console = False
  o.solver = "simplex"
  o.simplex_strategy = H.simplex_constants.SimplexStrategy.kSimplexStrategyDual
  o.highs_debug_level = 0
  # devex pricing at every level: steepest-edge weights of a user basis have to be computed from scratch (the same merges take
  # 3-4 x the pivots with them), and in the cold leaves devex saves a fifth of the pivots too
  o.simplex_dual_edge_weight_strategy = int(H.simplex_constants.SimplexEdgeWeightStrategy.kSimplexEdgeWeightStrategyDevex)
  if h.passOptions(o) == H.HighsStatus.kError:
    return None
  # the model goes in as arrays (the overload that takes numpy buffers: a HighsLp's vector members are filled element by element)
  st = h.passModel(n_col, n_row, int(A.nnz), int(H.MatrixFormat.kColwise), int(H.ObjSense.kMinimize), 0.0,
                   np.ascontiguousarray(c, dtype=np.float64), np.where(np.isinf(lb), -inf, lb), np.where(np.isinf(ub), inf, ub),
                   np.ascontiguousarray(b, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64),
                   np.ascontiguousarray(A.indptr, dtype=np.int32), np.ascontiguousarray(A.indices, dtype=np.int32),
                   np.ascontiguousarray(A.data, dtype=np.float64), np.zeros(n_col, dtype=np.int32))          # integrality: all continuous
  if st == H.HighsStatus.kError:
    return None
  if basis is not None:
    hb = H.HighsBasis()
    hb.col_status = basis[0]; hb.row_status = basis[1]
    if h.setBasis(hb) != H.HighsStatus.kOk:
      return None
  return _finish(h)


def _finish(h, pivots_before=0):
  """Run the solver `h` as it stands and read the result: (x, col_status, row_status, row_dual, col_dual, pivots, h) or None."""
  H = _core
  if h.run() == H.HighsStatus.kError or h.getModelStatus() != H.HighsModelStatus.kOptimal:
    return None
  sol = h.getSolution(); bas = h.getBasis(); info = h.getInfo()
  return (np.asarray(sol.col_value, dtype=np.float64), list(bas.col_status), list(bas.row_status),
          np.asarray(sol.row_dual, dtype=np.float64), np.asarray(sol.col_dual, dtype=np.float64),
          int(info.simplex_iteration_count) - pivots_before, h)


def _shift_slope_rows(h, slope_rows_rhs):
  """New right-hand sides for the slope rows (the first len(slope_rows_rhs) rows) of a solver that holds an optimal basis:
  the basis and its factorisation stay, the next run() is a hot start in the dual simplex's second phase."""
  for i, v in enumerate(slope_rows_rhs.tolist()):
    h.changeRowBounds(i, v, v)


def _merge_basis(children, root):
  """The start of the merged LP on [children[0].a, children[-1].e): every child's optimal basis; of each cut (the segment
  between one child's last and the next one's first point: one row of the slope block, one of the shot block, two of the
  rate block, and the jump, shot jump, rate jump and two rate changes that only these rows hold) every column non-basic at
  zero and the four rows' logicals basic.  That start is DUAL feasible -- the cut rows price at zero, so no reduced cost of
  any child changes, and the cuts' own columns price at their (positive) costs -- and primal infeasible in the cut rows only:
  the dual simplex goes straight to its second phase and repairs those rows.  (Making the cuts' columns basic instead, signed
  for primal feasibility, leaves thousands of dual infeasibilities -- a rate change prices its row at +-40 000 and the rate
  chain carries that through the children -- and costs 1.5-2.5 x the pivots.)  At the root the slope variable is free, from 0."""
  H = _core
  B, L, Z = H.HighsBasisStatus.kBasic, H.HighsBasisStatus.kLower, H.HighsBasisStatus.kZero
  cut_cols = ([], [], [L], [L], [], [], [L], [L], [L], [L], [L, L], [L, L])
  cut_rows = ([B], [B], [B, B])
  cols = [[] for _ in range(12)]
  rows = [[] for _ in range(3)]
  for k, ch in enumerate(children):
    sizes_c, sizes_r = block_sizes(ch.e - ch.a)
    for blk, part, cut in zip(cols, _split(ch.col_status, sizes_c), cut_cols):
      if k:
        blk += cut
      blk += part
    for blk, part, cut in zip(rows, _split(ch.row_status, sizes_r), cut_rows):
      if k:
        blk += cut
      blk += part
  return [s for blk in cols for s in blk] + [Z if root else L], [s for blk in rows for s in blk]


def _level_slope(level, x, m_c):
  """The weighted median (weights: segment lengths) of the slopes m_c + rate_jump / dx over every segment of the level's
  solved sub-LPs: what median_slope will be if the slopes stay as they are (it is the minimiser of the 0.001 |rate_jump| terms)."""
  sl, w = [], []
  for nd in level:
    nw = nd.e - nd.a
    dx = np.diff(x[nd.a:nd.e])
    sl.append((nd.x[8 * nw - 4:9 * nw - 5] - nd.x[9 * nw - 5:10 * nw - 6]) / dx)
    w.append(dx)
  sl = np.concatenate(sl); w = np.concatenate(w)
  o = np.argsort(sl, kind="stable")
  cw = np.cumsum(w[o])
  return m_c + float(sl[o][np.searchsorted(cw, 0.5 * cw[-1])])


def solve(x, y, jump_cost, leaf_points=None, stats=None):
  """The trend LP on fit points (x, y) through the tree of warm starts.  Returns (solution in the reference's variable
  order, row duals, column duals) with HiGHS' model status optimal at the root, or None."""
  if not available():
    return None
  n = len(x)
  x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
  jump_cost = np.asarray(jump_cost, dtype=np.float64)
  leaf = max(16, int(leaf_points or LEAF_POINTS))
  fan = max(2, int(os.environ.get("DALIGN_LP_FAN", "4")))
  # the tree is laid out from the top: two halves under the root (where the held slope is re-centred most reliably), `fan`
  # children under every other node, as many levels as bring the leaves closest to `leaf` points
  k = 2
  while (n / (k * fan)) * np.sqrt(fan) >= leaf and n // (k * fan) >= 16:
    k *= fan
  if n // k < 8:
    return None
  m_c = estimate_slope(x, y)
  cuts = [int(round(i * n / k)) for i in range(k)] + [n]
  pivots = []
  t_begin = time.perf_counter()

  recentre = os.environ.get("DALIGN_LP_RECENTRE", "1") != "0"
  recentred = []

  def solve_range(a, e, basis, root):
    c, A, b, lb, ub = assemble(x[a:e], y[a:e], jump_cost[a:e - 1], m_c)
    return _run(c, A, b, lb, ub, basis, free_slope=root)

  def recentre_level(level):
    """Move the held slope to where this level's solutions put the median (see _level_slope) and re-solve the level's
    sub-LPs for it from their own bases: only the right-hand side changes, the bases stay dual feasible, and the pivots it
    takes (the rate jumps of whole plateaus entering or leaving the basis) cost far less here than at the root."""
    nonlocal m_c
    m_new = _level_slope(level, x, m_c)
    if m_new == m_c or not (0.1 < m_new < 10.0):
      recentred.append(0)
      return True
    m_c = m_new
    spent = 0
    for nd in level:
      # same solver object, same basis and factorisation: only the slope rows' right-hand sides move
      dxs = np.diff(x[nd.a:nd.e])
      _shift_slope_rows(nd.solver, np.diff(y[nd.a:nd.e]) / dxs - m_c)
      got = _finish(nd.solver, pivots_before=int(nd.solver.getInfo().simplex_iteration_count))
      if got is None:
        return False
      nd.x, nd.col_status, nd.row_status, nd.row_dual, nd.col_dual, piv, nd.solver = got
      spent += piv
    recentred.append(spent)
    return True

  level = []
  for a, e in zip(cuts[:-1], cuts[1:]):
    got = solve_range(a, e, None, False)
    if got is None:
      return None
    nd = _Node()
    nd.a, nd.e = a, e
    nd.x, nd.col_status, nd.row_status, nd.row_dual, nd.col_dual, nd.pivots, nd.solver = got
    level.append(nd)
  pivots.append(sum(nd.pivots for nd in level))
  seconds = [time.perf_counter() - t_begin]
  while len(level) > 1:
    t_level = time.perf_counter()
    if recentre and not recentre_level(level):
      return None
    nxt, spent = [], 0
    root = len(level) == 2
    for i in range(0, len(level), fan):
      group = level[i:i + fan]
      if len(group) == 1:
        nxt.append(group[0])                         # an odd one out joins the next level as it is
        continue
      got = solve_range(group[0].a, group[-1].e, _merge_basis(group, root), root)
      if got is None:
        return None
      nd = _Node()
      nd.a, nd.e = group[0].a, group[-1].e
      nd.x, nd.col_status, nd.row_status, nd.row_dual, nd.col_dual, nd.pivots, nd.solver = got
      spent += nd.pivots
      nxt.append(nd)
    pivots.append(spent)
    seconds.append(time.perf_counter() - t_level)
    level = nxt
  top = level[0]
  sol = top.x.copy()
  sol[-1] += m_c
  if stats is not None:
    stats["_basis"] = (top.col_status, top.row_status, m_c)      # for refactor()
    stats.update(leaves=k, slope_held=m_c, pivots_per_level=pivots, recentre_pivots=recentred, seconds_per_level=[round(t, 3) for t in seconds])
  return sol, top.row_dual, top.col_dual


def kkt_certificate(c, A, b, lb, ub, sol, row_dual, tol=1e-6):
  """Is `sol` optimal for  min c'x, Ax = b, lb <= x <= ub  -- the LP exactly as the reference poses it -- with `row_dual` as
  witness?  Checked directly on that LP, whatever produced the two vectors: primal feasibility, dual feasibility (the sign
  of every reduced cost c - A'pi agrees with the bound the variable sits at; zero for a variable strictly between its bounds
  is implied by the gap), and equal primal and dual objectives.  Tolerances are relative to the cost / right-hand-side scale
  and an order looser than HiGHS' own 1e-7 (they are applied to the unscaled LP).  Returns (ok, worst violations)."""
  sol = np.asarray(sol, dtype=np.float64); pi = np.asarray(row_dual, dtype=np.float64)
  res = A @ sol - b
  p_inf = float(np.max(np.abs(res))) if len(res) else 0.0
  below = float(np.max(np.maximum(lb - sol, 0.0))); above = float(np.max(np.maximum(sol - ub, 0.0)))
  d = c - A.T @ pi
  scale = np.maximum(1.0, np.abs(c))
  at_lb = sol <= lb + tol
  at_ub = sol >= ub - tol
  free_low = np.isinf(lb)
  # a reduced cost may be negative only at an upper bound, positive only at a lower bound
  neg_bad = np.where(~at_ub, np.maximum(-d, 0.0) / scale, 0.0)
  pos_bad = np.where(~at_lb | free_low, np.maximum(d, 0.0) / scale, 0.0)
  d_inf = float(max(np.max(neg_bad), np.max(pos_bad)))
  primal = float(c @ sol)
  finite_ub = np.isfinite(ub)
  dual = float(b @ pi + np.sum(np.where(finite_ub, ub, 0.0) * np.minimum(d, 0.0)) + np.sum(np.where(np.isfinite(lb), lb, 0.0) * np.maximum(d, 0.0)))
  gap = abs(primal - dual) / max(1.0, abs(primal))
  worst = dict(primal_infeasibility=max(p_inf, below, above), dual_infeasibility=d_inf, relative_gap=gap)
  ok = p_inf <= tol * max(1.0, float(np.max(np.abs(b)))) and below <= tol and above <= tol and d_inf <= tol and gap <= tol
  return bool(ok), worst


def refactor(x, y, jump_cost, basis):
  """The full LP once more from the root's optimal basis (`stats["_basis"]` of solve()): no pivots, but a fresh factorisation.
  The duals that come back after thousands of basis updates are good to HiGHS' tolerance on ITS scaled LP, not always to 1e-6
  on the LP as posed (seen once: a rate jump's reduced cost off by 1.7e-6 of its 0.001 on the 1 h golden pair); the reference's
  call ends with such a clean-up too, after its postsolve.  Costs 0.1-0.3 s at 2 h (mostly moving 150 000 basis statuses
  through the binding), so the caller asks for it only when the certificate objects to the dual side.  Returns
  (solution, row duals, column duals) or None."""
  col_status, row_status, m_c = basis
  x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
  c, A, b, lb, ub = assemble(x, y, np.asarray(jump_cost, dtype=np.float64), m_c)
  got = _run(c, A, b, lb, ub, (col_status, row_status), free_slope=True)
  if got is None:
    return None
  sol = got[0].copy()
  sol[-1] += m_c
  return sol, got[3], got[4]


# ---- one long pair: the tree's independent sub-LPs on helper processes ---------------------------------------------------------
# A directory batch keeps every CPU busy with other pairs' LPs (align.AlignPipeline's worker processes call solve()).  ONE long
# pair on its own -- align() on a 4 h film, rank 0 of a tiled 8 h pair -- leaves them idle while its LP runs for seconds: the
# leaves (128 at 8 h) and the merges of a level are independent, so solve_parallel() hands them to a small pool of helper
# processes and only the root runs alone.  Same sub-LPs, same starts, same certificate; nothing else changes.
_pool = None
_STATUS = None


def _status_table():
  global _STATUS
  if _STATUS is None:
    H = _core
    _STATUS = [H.HighsBasisStatus(i) for i in range(5)]
  return _STATUS


class HelperPool:
  """`procs` helper processes (python -m describealign_amd.lp_helper) fed through pipes; run() hands them a list of tasks and returns
  the results in order.  Started once per process, lazily; ended when the interpreter exits."""

  def __init__(self, procs):
    import subprocess
    import sys
    self.procs = int(procs)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
    env["PYTHONPATH"] = root + (os.pathsep + env["PYTHONPATH"] if env.get("PYTHONPATH") else "")
    self.children = [subprocess.Popen([sys.executable, "-m", "describealign_amd.lp_helper"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
                     for _ in range(self.procs)]
    self.ready = None
    self.lock = threading.Lock()

  def wait_ready(self):
    with self.lock:
      if self.ready is None:
        self.ready = all(ch.stdout.read(1) == b"R" for ch in self.children)
      return self.ready

  def run(self, tasks):
    """tasks: argument tuples of _pool_solve.  Results in task order; a helper that died or raised makes the whole call fail
    (RuntimeError): the caller falls back to the serial tree."""
    import pickle
    import struct
    if not self.wait_ready():
      raise RuntimeError("lp helper processes did not start")
    results = [None] * len(tasks)
    nxt = [0]
    errors = []
    take = threading.Lock()

    def feed(ch):
      while True:
        with take:
          k = nxt[0]
          nxt[0] += 1
        if k >= len(tasks) or errors:
          return
        try:
          blob = pickle.dumps(tasks[k], protocol=pickle.HIGHEST_PROTOCOL)
          ch.stdin.write(struct.pack("<Q", len(blob))); ch.stdin.write(blob); ch.stdin.flush()
          head = ch.stdout.read(8)
          if len(head) < 8:
            raise RuntimeError("lp helper process ended")
          (size,) = struct.unpack("<Q", head)
          res = pickle.loads(ch.stdout.read(size))
          if isinstance(res, tuple) and len(res) == 2 and res[0] == "error":
            raise RuntimeError(res[1])
          results[k] = res
        except Exception as e:                   # noqa: BLE001
          errors.append(e)
          return

    with self.lock:                              # one run() at a time per pool
      threads = [threading.Thread(target=feed, args=(ch,), daemon=True) for ch in self.children[:max(1, min(self.procs, len(tasks)))]]
      for t in threads:
        t.start()
      for t in threads:
        t.join()
    if errors:
      self.close()
      raise RuntimeError(f"lp helper failed: {errors[0]}")
    return results

  def close(self):
    global _pool
    for ch in self.children:
      try:
        ch.stdin.close()
      except Exception:
        pass
    for ch in self.children:
      try:
        ch.wait(timeout=2)
      except Exception:
        ch.kill()
    self.children = []
    if _pool is self:
      _pool = None


_pool_guard = threading.Lock()


def helper_pool(procs):
  """This process's pool of helper processes (at least `procs` of them), started on first use."""
  global _pool
  import atexit
  with _pool_guard:
    if _pool is not None and _pool.procs >= procs and _pool.children:
      return _pool
    if _pool is not None:
      _pool.close()
    _pool = HelperPool(procs)
    atexit.register(_pool.close)
    return _pool


def _codes(statuses):
  return np.fromiter((int(s) for s in statuses), dtype=np.int8, count=len(statuses))


def _pool_solve(xs, ys, jcs, m_c, col_codes, row_codes, free_slope):
  """One sub-LP in a helper process: the bases travel as int8 codes.  Returns (x, col codes, row codes, row duals, col duals,
  pivots) or None."""
  if not available():
    return None
  basis = None
  if col_codes is not None:
    tab = _status_table()
    basis = ([tab[i] for i in col_codes.tolist()], [tab[i] for i in row_codes.tolist()])
  c, A, b, lb, ub = assemble(xs, ys, jcs, m_c)
  got = _run(c, A, b, lb, ub, basis, free_slope=free_slope)
  if got is None:
    return None
  return got[0], _codes(got[1]), _codes(got[2]), got[3], got[4], got[5]


class _PNode:
  __slots__ = ("a", "e", "x", "col", "row", "row_dual", "col_dual")


def _merge_codes(children, root):
  """_merge_basis on int8 codes: children's bases, cut columns at their lower bound, cut rows' logicals basic."""
  H = _core
  B, L, Z = int(H.HighsBasisStatus.kBasic), int(H.HighsBasisStatus.kLower), int(H.HighsBasisStatus.kZero)
  cut_cols = (0, 0, 1, 1, 0, 0, 1, 1, 1, 1, 2, 2)
  cut_rows = (1, 1, 2)
  cols = [[] for _ in range(12)]
  rows = [[] for _ in range(3)]
  for k, ch in enumerate(children):
    sizes_c, sizes_r = block_sizes(ch.e - ch.a)
    at = 0
    for j in range(12):
      if k and cut_cols[j]:
        cols[j].append(np.full(cut_cols[j], L, dtype=np.int8))
      cols[j].append(ch.col[at:at + sizes_c[j]]); at += sizes_c[j]
    at = 0
    for j in range(3):
      if k:
        rows[j].append(np.full(cut_rows[j], B, dtype=np.int8))
      rows[j].append(ch.row[at:at + sizes_r[j]]); at += sizes_r[j]
  col = np.concatenate([a for blk in cols for a in blk] + [np.array([Z if root else L], dtype=np.int8)])
  row = np.concatenate([a for blk in rows for a in blk])
  return col, row


def solve_parallel(x, y, jump_cost, procs, leaf_points=None, stats=None):
  """solve() with the sub-LPs of every level run on `procs` helper processes (the root, and the re-centring of the two halves
  beside each other, are what is left of the critical path).  Returns what solve() returns, or None."""
  if not available():
    return None
  n = len(x)
  x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)
  jump_cost = np.asarray(jump_cost, dtype=np.float64)
  leaf = max(16, int(leaf_points or LEAF_POINTS))
  fan = max(2, int(os.environ.get("DALIGN_LP_FAN", "4")))
  k = 2
  while (n / (k * fan)) * np.sqrt(fan) >= leaf and n // (k * fan) >= 16:
    k *= fan
  if n // k < 8:
    return None
  pool = helper_pool(procs)
  m_c = estimate_slope(x, y)
  cuts = [int(round(i * n / k)) for i in range(k)] + [n]
  t_begin = time.perf_counter()
  pivots, seconds, recentred = [], [], []

  def run_all(jobs):
    """jobs: (a, e, col codes | None, row codes | None, free_slope) -> _PNodes in order, pivots; None if any sub-LP failed."""
    res = pool.run([(x[a:e], y[a:e], jump_cost[a:e - 1], m_c, cc, rc, fs) for a, e, cc, rc, fs in jobs])
    out, spent = [], 0
    for (a, e, _, _, _), got in zip(jobs, res):
      if got is None:
        return None, 0
      nd = _PNode()
      nd.a, nd.e = a, e
      nd.x, nd.col, nd.row, nd.row_dual, nd.col_dual, piv = got
      spent += piv
      out.append(nd)
    return out, spent

  level, spent = run_all([(a, e, None, None, False) for a, e in zip(cuts[:-1], cuts[1:])])
  if level is None:
    return None
  pivots.append(spent); seconds.append(time.perf_counter() - t_begin)
  recentre = os.environ.get("DALIGN_LP_RECENTRE", "1") != "0"
  while len(level) > 1:
    t_level = time.perf_counter()
    if recentre:
      m_new = _level_slope(level, x, m_c)
      if m_new != m_c and 0.1 < m_new < 10.0:
        m_c = m_new
        level, spent = run_all([(nd.a, nd.e, nd.col, nd.row, False) for nd in level])
        if level is None:
          return None
        recentred.append(spent)
      else:
        recentred.append(0)
    root = len(level) == 2
    jobs, carried = [], []
    for i in range(0, len(level), fan):
      group = level[i:i + fan]
      if len(group) == 1:
        carried.append((len(jobs), group[0]))
        continue
      col, row = _merge_codes(group, root)
      jobs.append((group[0].a, group[-1].e, col, row, root))
    merged, spent = run_all(jobs)
    if merged is None:
      return None
    for at, nd in carried:
      merged.insert(at, nd)
    pivots.append(spent); seconds.append(time.perf_counter() - t_level)
    level = merged
  top = level[0]
  sol = top.x.copy()
  sol[-1] += m_c
  if stats is not None:
    tab = _status_table()
    stats["_basis"] = ([tab[i] for i in top.col.tolist()], [tab[i] for i in top.row.tolist()], m_c)
    stats.update(leaves=k, slope_held=m_c, pivots_per_level=pivots, recentre_pivots=recentred, seconds_per_level=[round(t, 3) for t in seconds],
                 helper_processes=procs)
  return sol, top.row_dual, top.col_dual
