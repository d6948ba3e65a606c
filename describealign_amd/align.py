"""Host side of the MI355X alignment path: the drop-in for describealign's `align()`.

    align(video_features, audio_desc_features, video_energy, audio_desc_energy)
        -> (audio_desc_times, video_times, similarity_percent, path, median_slope)

is the seam `combine()` calls (reference describealign.py:1121-1122).  The quadratic and
data-parallel stages run in HIP behind the C ABI (include/dalign.h):

  stage 1+2  mean-sub / norms / hash digits, similarity GEMM on MFMA, exact verification  da_match
  stage 2    heaviest-chain DP over the verified matches (device-resident)                  da_chain_begin/finish
  stage 4    banded line extension + second DP                                              da_refine

and this module does what the reference keeps on the host: the pass-1 continuity filter,
per-feature scaling and path compression (:701-767), the L1 trend-fit LP through
scipy.optimize.linprog exactly as the reference poses it (:769-858), the line clustering
(:861-893) and the node / similarity extraction (:993-1027).
"""
from __future__ import annotations

import time

import os

import numpy as np
import scipy.optimize
import scipy.sparse
from scipy.signal.windows import hann as _hann

from . import _native

FRAMES_PER_SECOND = 210
NODE_FRAMES = 21                      # 210 // TIMESTEPS_PER_SECOND (:29, :596)
MISMATCH_MSG = "Alignment failed, are the input files mismatched?"
LP_FAIL_MSG = "Smooth Alignment L1-Min Optimization Failed!"

_W41 = _hann(2 * NODE_FRAMES + 1)[1:-1]
_W41N = _W41 / np.sum(_W41)


def min_path_length(n_video: int, n_audio: int) -> float:
  """Shorter paths mean mismatched inputs (:698, :991)."""
  return max(min(n_video, n_audio) / 500.0, 5 * FRAMES_PER_SECOND)


# ------------------------------------------------------------------------------------ stage 3
def _smooth_same(a):
  return np.convolve(_W41N, a, mode="same")[:len(a)]


def continuity_error(x, y, deriv=False):
  """How far each path point is from lines through smoothed neighbours ahead/behind (:702-724)."""
  taps = _W41N[:NODE_FRAMES - 1]
  taps = taps / taps.sum()
  lag = NODE_FRAMES // 2
  delay = NODE_FRAMES + lag - 2
  x = np.asarray(x, dtype=np.float64); y = np.asarray(y, dtype=np.float64)

  def line_through(kernel):
    xs = np.convolve(x, kernel, mode="valid"); ys = np.convolve(y, kernel, mode="valid")
    slope = (ys[lag:] - ys[:-lag]) / (xs[lag:] - xs[:-lag])
    return xs, ys, slope

  xs, ys, sf = line_through(taps)
  ahead = np.abs(sf * x[:-delay] + (ys[:-lag] - xs[:-lag] * sf) - y[:-delay])
  xs, ys, sp = line_through(taps[::-1])
  behind = np.abs(sp * x[delay:] + (ys[lag:] - xs[lag:] * sp) - y[delay:])
  shift = delay - (1 if deriv else 0)
  err = np.full(len(x) - (1 if deriv else 0), np.inf)
  err[:len(err) - shift] = ahead
  err[shift:] = np.minimum(err[shift:], behind)
  return err


def scale_feature_stacks(video_features, audio_features, x, y):
  """(L,3) stacks of the first three features in units of the audio feature's std, the video
  side additionally scaled by its least-squares gain onto the audio at the path (:733-741)."""
  a_cols, v_cols = [], []
  for vf, af in list(zip(video_features, audio_features))[:3]:
    vf = np.asarray(vf); af = np.asarray(af)
    sd = np.std(af)
    gain = np.linalg.lstsq(vf[y][:, None], af[x], rcond=None)[0]
    a_cols.append(af / sd)
    v_cols.append(vf * gain / sd)
  na = min(map(len, a_cols)); nv = min(map(len, v_cols))
  a = np.empty((na, 3)); v = np.empty((nv, 3))
  for k in range(3):
    a[:, k] = a_cols[k][:na]; v[:, k] = v_cols[k][:nv]
  return a, v


def compress_path(x, y, run=70, tol=3):
  """Replace straight runs of `run` path points by their mean; merge equal-x points (:743-767)."""
  sx, sy = _smooth_same(x), _smooth_same(y)
  with np.errstate(divide="ignore", invalid="ignore"):
    slope = np.diff(sy) / np.diff(sx)
    dev = np.abs(slope * x[:-1] + (sy[:-1] - sx[:-1] * slope) - y[:-1])
  starts = np.arange(10, len(x) - 80, run)
  if len(starts) == 0:
    raise RuntimeError(MISMATCH_MSG)
  ok = dev < tol
  # all(ok[s:s+run]) per run via a prefix count (NaN deviations compare False, as in numpy)
  csum = np.concatenate([[0], np.cumsum(ok)])
  ends = np.minimum(starts + run, len(dev))
  straight = (csum[ends] - csum[starts]) == (ends - starts)
  px, py = [np.asarray(x[:10], dtype=np.float64)], [np.asarray(y[:10], dtype=np.float64)]
  for s, flat in zip(starts.tolist(), straight.tolist()):
    if flat:
      px.append(np.array([np.mean(x[s:s + run])])); py.append(np.array([np.mean(y[s:s + run])]))
    else:
      px.append(np.asarray(x[s:s + run], dtype=np.float64)); py.append(np.asarray(y[s:s + run], dtype=np.float64))
  tail = int(starts[-1]) + run
  px.append(np.asarray(x[tail:tail + run], dtype=np.float64)); py.append(np.asarray(y[tail:tail + run], dtype=np.float64))
  fx, fy = np.concatenate(px), np.concatenate(py)
  # consecutive duplicates of x collapse to the mean of their y's (x is non-decreasing)
  first = np.concatenate([[True], fx[1:] != fx[:-1]])
  idx = np.flatnonzero(first)
  counts = np.diff(np.concatenate([idx, [len(fx)]]))
  ux = fx[idx]
  uy = np.array([np.mean(fy[i:i + c]) for i, c in zip(idx.tolist(), counts.tolist())])
  return ux, uy


def trend_jump_cost(x, y):
  """Cost per unit of `jump` on every segment (:778-779): 10, less where the path is already discontinuous."""
  return np.full(len(x) - 1, 10.0) / np.maximum(1, np.sqrt(continuity_error(x, y, deriv=True) / 3.0))


def build_trend_lp(x, y):
  """The reference's L1 trend-filter LP (:773-840; variable layout SURVEY appendix A.6), assembled directly as COO triplets
  (lp_tree.assemble).  Returns (c, A_eq csc, b_eq, bounds) as the reference hands them to linprog."""
  from . import lp_tree
  n = len(x)
  c, A, b, _, _ = lp_tree.assemble(np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), trend_jump_cost(x, y))
  bounds = [[0, None]] * (4 * n - 2) + [[0, 2.0]] * (2 * n) + [[0, None]] * (6 * n - 8) + [[None, None]]
  return c, A, b, bounds


def _rate_terms_certificate(pi1, dx, tol=1e-7):
  """Can the block-3 duals z (one per interior fit point) be chosen so that every rate_jump and
  rate_change column of the full LP prices out non-negative?  With the block-1 duals pi1 of the LP
  solved WITHOUT those columns the conditions are (reduced cost = c - A'pi >= 0 at a lower bound):
      rate_jump+-[k]   : |pi1[k] - z[k] + z[k-1]| <= 0.001 * dx[k]        (z[-1] = z[n-2] = 0)
      rate_change+-[k] : |z[k]| <= 40 000
  which is a chain of intervals: O(n) propagation.  True = the reduced optimum, padded with zeros,
  is optimal for the full LP (to the solver's own dual feasibility tolerance)."""
  lo = hi = 0.0
  m = len(pi1)
  cap = 40000.0 + tol
  for k in range(m - 1):
    e = 0.001 * dx[k] + tol
    lo = max(lo + pi1[k] - e, -cap)
    hi = min(hi + pi1[k] + e, cap)
    if lo > hi:
      return False
  e = 0.001 * dx[m - 1] + tol
  return lo + pi1[m - 1] - e <= 0.0 <= hi + pi1[m - 1] + e


def _solve_without_rate_terms(c, A, b, bounds, n, dx):
  """The LP minus its rate_jump / rate_change columns and the block-3 rows that only they touch:
  a third of the columns and rows, 3-3.5x fewer simplex iterations.  Returns the full-length
  solution if the certificate above proves it optimal for the full LP, else None."""
  o_rj, o_med = 8 * n - 4, 12 * n - 10
  cols = np.concatenate([np.arange(o_rj), [o_med]])
  rows = 2 * n - 2
  Ar = A[:rows][:, cols]
  bd = [bounds[k] for k in cols]
  fit = scipy.optimize.linprog(c[cols], A_eq=Ar, b_eq=b[:rows], bounds=bd, method="highs-ds")
  if not fit.success:
    return None
  pi1 = np.asarray(fit.eqlin.marginals)[:n - 1]
  if not _rate_terms_certificate(pi1, dx):
    return None
  sol = np.zeros(len(c))
  sol[cols] = fit.x
  return sol


def solve_trend_lp(x, y, pricing=None, reduce=None, tree=None, procs=0):
  """The reference's scipy.optimize.linprog call (:841-858): HiGHS dual simplex, IPM retry on
  status 4.

  reduce (opt-in: reduce=True or DALIGN_LP_REDUCE=1; off by default, so the default IS the
  reference's call): for short inputs the optimum of this LP usually has every rate_jump and
  rate_change variable at zero -- a rate difference between the files is absorbed by median_slope;
  they only come into play when the rate changes inside the file, or when the running sum of the
  slope-row duals reaches the rate_change cost of 40 000, which long files do.  The LP is then first
  solved without those 4n-6 columns and the n-2 rows they alone appear in (same solver, same call),
  and a dual certificate (_rate_terms_certificate) decides whether that optimum, padded with zeros,
  is optimal for the full LP; if not, the full LP is solved exactly as the reference does.  The
  certificate proves OPTIMALITY, not that HiGHS would return the same vertex when the optimum is
  not unique (median_slope has zero cost; L1 fits can tie on exactly colinear data), which is why
  this shortcut is not the default (10-minute golden pair: 0.46 -> 0.16 s, solution equal to 1e-11).

  pricing="reference" (the default) keeps HiGHS' default steepest-edge pricing; "dantzig" / "devex"
  (or DALIGN_LP_PRICING) select another rule for the full solve: same optimum, 1.3-2x faster on an
  idle core, but slower with 24 solves side by side on the GPU box, so not the default."""
  c, A, b, bounds = build_trend_lp(x, y)
  n = len(x)
  if reduce is None:
    env = os.environ.get("DALIGN_LP_REDUCE", "")
    reduce = (env == "1")
  if tree is None:
    tree = os.environ.get("DALIGN_LP_TREE", "1") != "0"
  s, how, tree_stats = None, "reference", {}
  if tree and pricing in (None, "reference") and not reduce:
    s = _solve_by_tree(x, y, c, A, b, tree_stats, procs)
    how = "tree" if s is not None else ("reference (tree: " + tree_stats.get("declined", "unavailable") + ")")
  if s is None and reduce and n >= 4:
    s = _solve_without_rate_terms(c, A, b, bounds, n, np.diff(x))
  if s is None:
    s = _solve_full_lp(c, A, b, bounds, pricing)
  fit_err = s[:n] - s[n:2 * n]
  rate_jump = s[8 * n - 4:9 * n - 5] - s[9 * n - 5:10 * n - 6]
  median_slope = s[-1]
  slopes = median_slope + rate_jump / np.diff(x)
  return dict(solution=s, fit_err=fit_err, slopes=slopes, median_slope=median_slope,
              smooth_x=np.asarray(x, dtype=np.float64), smooth_y=np.asarray(y) - fit_err, method=how, tree=tree_stats)


def _solve_by_tree(x, y, c, A, b, stats, procs=0):
  """The same LP -- the reference's, whole -- solved by the same HiGHS dual simplex, started from the bases of sub-LPs instead
  of from the slack basis (lp_tree: leaves of ~300 fit points solved cold, merged four at a time, the last merge is the full LP).
  A 2 h pair's solve takes a third of the CPU time, an 8 h pair's a tenth.  What HiGHS returns at the root is accepted only if
  it passes the optimality conditions of the LP as the reference poses it (lp_tree.kkt_certificate: primal and dual
  feasibility and a closed gap, checked on the original c, A, b and bounds); otherwise -- scipy without the binding, fewer
  than lp_tree.MIN_POINTS fit points, any status but optimal, any exception -- None: the caller makes the reference's call."""
  from . import lp_tree
  n = len(x)
  if n < lp_tree.MIN_POINTS:
    stats["declined"] = "fewer than %d fit points" % lp_tree.MIN_POINTS
    return None
  if not lp_tree.available():
    stats["declined"] = "scipy.optimize._highspy._core not usable"
    return None
  try:
    got = None
    if procs >= 2 and n >= lp_tree.PARALLEL_MIN_POINTS:
      # one long pair on its own: the independent sub-LPs of every level on helper processes (lp_tree.solve_parallel)
      try:
        got = lp_tree.solve_parallel(x, y, c[2 * n:3 * n - 1], procs, stats=stats)
      except Exception as e:                                # noqa: BLE001 -- helpers gone: the same tree in this process
        stats["helpers_failed"] = f"{type(e).__name__}: {e}"
    if got is None:
      got = lp_tree.solve(x, y, c[2 * n:3 * n - 1], stats=stats)
    if got is None:
      stats["declined"] = "a sub-LP did not end optimal"
      return None
    sol, row_dual, _ = got
    lb = np.zeros(len(c)); ub = np.full(len(c), np.inf)
    ub[4 * n - 2:6 * n - 2] = 2.0
    lb[-1] = -np.inf
    basis = stats.pop("_basis", None)
    ok, worst = lp_tree.kkt_certificate(c, A, b, lb, ub, sol, row_dual)
    if not ok and basis is not None and worst["primal_infeasibility"] <= 1e-6 and worst["relative_gap"] <= 1e-6:
      # only the multipliers are off (update error of a long warm-started run): the same basis, freshly factorised
      again = lp_tree.refactor(x, y, c[2 * n:3 * n - 1], basis)
      if again is not None:
        sol, row_dual, _ = again
        ok, worst = lp_tree.kkt_certificate(c, A, b, lb, ub, sol, row_dual)
        stats["refactored"] = True
    stats["certificate"] = worst
    if not ok:
      stats["declined"] = "optimality certificate failed"
      return None
    return sol
  except Exception as e:                                    # noqa: BLE001 -- whatever it was, the reference's call is the answer
    stats["declined"] = f"{type(e).__name__}: {e}"
    return None


def _solve_full_lp(c, A, b, bounds, pricing=None):
  """linprog exactly as the reference calls it (:841-848), optionally with another pricing rule."""
  pricing = pricing or os.environ.get("DALIGN_LP_PRICING", "reference")
  fit = None
  if pricing != "reference":
    fit = scipy.optimize.linprog(c, A_eq=A, b_eq=b, bounds=bounds, method="highs-ds",
                                 options=dict(simplex_dual_edge_weight_strategy=pricing))
  if fit is None or not fit.success:           # the reference's own sequence (also its error behaviour)
    fit = scipy.optimize.linprog(c, A_eq=A, b_eq=b, bounds=bounds, method="highs-ds")
  if not fit.success and fit.status == 4:
    fit = scipy.optimize.linprog(c, A_eq=A, b_eq=b, bounds=bounds, method="highs-ipm")
  if not fit.success:
    print(fit)
    raise RuntimeError(LP_FAIL_MSG)
  return fit.x


# ------------------------------------------------------------------------------------ stage 4
def cluster_lines(sx, sy, slopes):
  """Colinear clustering of the smooth path (:861-893).  Returns arrays (x_first, x_last,
  offset, slope), one entry per kept line cluster, in the reference's order."""
  s_ext = np.concatenate([slopes[:1], slopes, slopes[-1:]])
  s_round = np.round(s_ext, 6)
  buckets = {}
  for k in range(len(sx)):
    px, py = sx[k], sy[k]
    for t in (k, k + 1):
      s = s_ext[t]
      if s < 0.1 or s > 10:
        continue
      key = (s_round[t], int(np.round(py - s * px, 0)))
      buckets.setdefault(key, []).append((px, py))
  order = sorted(buckets.items(), key=lambda kv: -len(kv[1]))      # stable: ties keep insertion order
  taken = set()
  merged = []
  for key, pts in order:
    if key in taken:
      continue
    s, o = key
    taken.add(key)
    del buckets[key]
    for key2 in list(buckets.keys()):
      p2 = buckets[key2]
      if abs(p2[0][1] - (p2[0][0] * s + o)) < 3 and abs(p2[-1][1] - (p2[-1][0] * s + o)) < 3:
        pts.extend(p2)
        taken.add(key2)
        del buckets[key2]
    merged.append(pts)
  x0, x1, off, slo = [], [], [], []
  for pts in merged:
    pts = sorted(pts)
    if not (abs(pts[0][0] - pts[-1][0]) > 10 and len(pts) > 5):
      continue
    cx, cy = np.array(pts).T
    sol = np.linalg.lstsq(np.stack([np.ones(len(cx)), cx], axis=1), cy, rcond=None)[0]
    x0.append(cx[0]); x1.append(cx[-1]); off.append(sol[0]); slo.append(sol[1])
  return (np.array(x0, dtype=np.float64), np.array(x1, dtype=np.float64),
          np.array(off, dtype=np.float64), np.array(slo, dtype=np.float64))


def nodes_and_similarity(path, n_audio_scaled, n_video_scaled, n_audio_energy, n_video_energy):
  """Similarity percentage and piecewise-linear nodes in seconds (:993-1027)."""
  vj, ai, cl, q = path[:, 0], path[:, 1], path[:, 2], path[:, 3]
  solid = (q == 0) | (q > 0.3)
  sim = 100 * max(len(np.unique(ai[solid])) / n_audio_scaled, len(np.unique(vj[solid])) / n_video_scaled)
  change = np.flatnonzero(cl[:-1] != cl[1:])
  nx, ny = [], []
  if cl[0] == cl[1]:
    nx.append(ai[0]); ny.append(vj[0])
  for k in change.tolist():
    nx += [ai[k] - 0.1, ai[k + 1] + 0.1]
    ny += [vj[k] - 0.1, vj[k + 1] + 0.1]
  if cl[-2] == cl[-1]:
    nx.append(ai[-1]); ny.append(vj[-1])
  nx = np.array(nx) / float(FRAMES_PER_SECOND); ny = np.array(ny) / float(FRAMES_PER_SECOND)
  if nx[1] - nx[0] > 2:
    s = (ny[1] - ny[0]) / (nx[1] - nx[0])
    nx[0] = 0
    ny[0] = ny[1] - nx[1] * s
    if ny[0] < 0:
      nx[0] = nx[1] - ny[1] / s
      ny[0] = 0
  if nx[-1] - nx[-2] > 2:
    s = (ny[-1] - ny[-2]) / (nx[-1] - nx[-2])
    nx[-1] = (n_audio_energy - 1) / float(FRAMES_PER_SECOND)
    ny[-1] = ny[-2] + (nx[-1] - nx[-2]) * s
    v_end = (n_video_energy - 1) / float(FRAMES_PER_SECOND)
    if ny[-1] > v_end:
      ny[-1] = v_end
      nx[-1] = nx[-2] + (ny[-1] - ny[-2]) / s
  return nx, ny, sim


# ------------------------------------------------------------------------------------ align()
_default_ctx = None


def default_context(device: int = 0, precision: int = _native.PREC_F32) -> "_native.Context":
  global _default_ctx
  if _default_ctx is None or _default_ctx.precision != precision or _default_ctx.device != device:
    _default_ctx = _native.Context(device, precision)
  return _default_ctx


def lp_helper_procs(n_video_frames):
  """Helper processes for the LP of ONE pair aligned on its own (align(), rank 0 of align_tiled()): none for pairs under
  80 minutes (their LP takes a second) and none inside a batch pipeline (its worker processes keep every CPU busy with other
  pairs' LPs); otherwise half the CPUs this process may use, at most 8.  DALIGN_LP_PROCS overrides (0: never)."""
  env = os.environ.get("DALIGN_LP_PROCS", "")
  if env != "":
    return max(0, int(env))
  if n_video_frames < FRAMES_PER_SECOND * 4800:
    return 0
  budget = cpu_quota()
  if budget is None:
    budget = len(_cpu_topology()[0])
  return int(max(2, min(8, budget // 2)))


def _warm_lp_helpers(procs):
  """Start the helper processes (imports: ~1 s) beside the GPU stages, not in front of the LP."""
  if procs < 2:
    return
  import threading
  from . import lp_tree

  def start():
    try:
      if lp_tree.available():
        lp_tree.helper_pool(procs).wait_ready()
    except Exception:
      pass
  threading.Thread(target=start, daemon=True).start()


def _stage_gpu_match(ctx, video_features, audio_desc_features, mode, tm, rows=None):
  """Stages 1+2 on the GPU: prep, similarity GEMM, exact verification, sort.  The verified
  matches stay on the device; returns their number."""
  t0 = time.perf_counter()
  ctx.match_begin(video_features, audio_desc_features, mode, rows)
  n = ctx.match_finish()
  tm["device"] = ctx.stats()
  tm["match_s"] = time.perf_counter() - t0
  tm["n_matches"] = n
  return n


def _stage_pass1(px, py, video_features, audio_desc_features, tm):
  """Pass-1 host work on the stage-2 path (:701-767): continuity filter, per-feature scaling,
  path compression.  px = audio frames, py = video frames of the chain."""
  t2 = time.perf_counter()
  x = np.asarray(px).astype(np.int64); y = np.asarray(py).astype(np.int64)
  keep = continuity_error(x, y) < 3
  x, y = x[keep], y[keep]
  a_scaled, v_scaled = scale_feature_stacks(video_features, audio_desc_features, x, y)
  fx, fy = compress_path(x, y)
  tm.update(pass1_host_s=time.perf_counter() - t2, n_path1=len(px), n_fit_points=len(fx))
  return fx, fy, a_scaled, v_scaled


def _stage_match(ctx, video_features, audio_desc_features, n_ve, n_ae, mode, tm):
  """Matching + chain DP, both on the device; only the path comes back."""
  _stage_gpu_match(ctx, video_features, audio_desc_features, mode, tm)
  t1 = time.perf_counter()
  px, py = ctx.chain_resident(min_len=min_path_length(n_ve, n_ae))          # raises the mismatch error (:698)
  tm["device"]["chain_ms"] = ctx.stats()["chain_ms"]
  tm["chain_s"] = time.perf_counter() - t1
  return _stage_pass1(px, py, video_features, audio_desc_features, tm)


def _stage_refine(ctx, lp, a_scaled, v_scaled, n_ve, n_ae, tm, clusters=None):
  """Stage 4: clustering, banded extension + second DP (GPU + host), nodes."""
  t0 = time.perf_counter()
  x0, x1, off, slo = clusters if clusters is not None else cluster_lines(lp["smooth_x"], lp["smooth_y"], lp["slopes"])
  t1 = time.perf_counter()
  path, n_points = ctx.refine(a_scaled, v_scaled, x0, x1, off, slo, min_len=min_path_length(n_ve, n_ae))
  st = ctx.stats()
  t2 = time.perf_counter()
  nx, ny, sim = nodes_and_similarity(path, len(a_scaled), len(v_scaled), n_ae, n_ve)
  path[:, :2] /= float(FRAMES_PER_SECOND)
  t3 = time.perf_counter()
  tm.update(cluster_s=t1 - t0, refine_s=t2 - t1, nodes_s=t3 - t2, n_clusters=len(x0), n_points=n_points,
            n_path2=len(path))
  tm.setdefault("device", {}).update(refine_kernel_ms=st["refine_kernel_ms"], refine_dp_ms=st["refine_dp_ms"],
                                     refine_points=st["refine_points"])
  return nx, ny, sim, path, lp["median_slope"]


def align(video_features, audio_desc_features, video_energy, audio_desc_energy, ctx=None, timings=None,
          mode=_native.MATCH_HASHED):
  """Drop-in for describealign.align (:595-1027); same arguments, same return tuple."""
  ctx = ctx or default_context()
  tm = timings if timings is not None else {}
  t0 = time.perf_counter()
  n_ve, n_ae = len(video_energy), len(audio_desc_energy)
  procs = lp_helper_procs(n_ve)
  _warm_lp_helpers(procs)
  print("  memorizing video...        \r", end='')
  print("  matching audio...  \r", end='')
  fx, fy, a_scaled, v_scaled = _stage_match(ctx, video_features, audio_desc_features, n_ve, n_ae, mode, tm)
  print("  refining match: pass 1 of 2...\r", end='')
  t1 = time.perf_counter()
  lp = solve_trend_lp(fx, fy, procs=procs)
  tm["lp_s"] = time.perf_counter() - t1
  tm["lp_method"] = lp["method"]
  tm["lp_helper_processes"] = lp["tree"].get("helper_processes", 0)
  print("  refining match: pass 2 of 2...\r", end='')
  out = _stage_refine(ctx, lp, a_scaled, v_scaled, n_ve, n_ae, tm)
  tm["total_s"] = time.perf_counter() - t0
  return out


def align_tiled(video_features, audio_desc_features, video_energy, audio_desc_energy, group, ctx=None,
                timings=None, mode=_native.MATCH_HASHED, match_lock=None):
  """One long pair across all ranks of `group` (BASELINE config 5): the quadratic matching stage
  is split into contiguous audio-row blocks, one per GPU (each rank holds all video features);
  the verified match lists are gathered on rank 0 in ONE exchange (RCCL over xGMI, device to device:
  Group.gather_matches_to_root), rank 0 alone runs what is sequential -- chain DP on its GPU, the
  host LP, pass 2 -- and broadcasts the result.  Returns the same tuple as align() on every rank.
  After matching every rank gives the stage's scratch memory back (da_trim: an 8 h pair's survivor buffer
  is 23 GB per rank).  match_lock: optional context manager held during this rank's matching stage --
  for ranks that SHARE a device (tests, one-GPU emulation of an 8-GPU run) and would not fit side by side."""
  from .distrib import row_blocks_by_load
  ctx = ctx or default_context()
  tm = timings if timings is not None else {}
  n_ve, n_ae = len(video_energy), len(audio_desc_energy)
  procs = lp_helper_procs(n_ve) if group.rank == 0 else 0
  _warm_lp_helpers(procs)
  t0 = time.perf_counter()
  # blocks of about equal numbers of non-quiet audio rows (:657-658), the rows that cost anything
  n_rows = max(0, n_ae - (2 * NODE_FRAMES - 1))
  blocks = row_blocks_by_load(np.asarray(audio_desc_energy[:n_rows]) > 0.5, group.world)
  # every rank derives the cuts from ITS copy of the energy row: rank 0's are used everywhere and a rank that had
  # derived others (one energy value either side of 0.5) fails loudly instead of dropping or doubling rows
  cuts = group.agree_on([b for b, _ in blocks] + [blocks[-1][1]])
  rb, re = cuts[group.rank], cuts[group.rank + 1]
  import contextlib
  n_local, match_err = 0, None
  try:
    with (match_lock if match_lock is not None else contextlib.nullcontext()):
      n_local = _stage_gpu_match(ctx, video_features, audio_desc_features, mode, tm, rows=(rb, re))
      ctx.trim()
  except BaseException as e:                                 # e.g. out of memory on this rank's survivor buffer
    match_err = e
  # a rank that failed must not leave the others blocked in the gather
  if not group.all_ok(match_err is None):
    if match_err is not None:
      raise match_err
    raise RuntimeError("align_tiled: the matching stage failed on another rank")
  t1 = time.perf_counter()
  total = group.gather_matches_to_root(ctx, n_local)
  t2 = time.perf_counter()
  tm.update(match_s=t1 - t0, gather_s=t2 - t1, n_matches=total if total is not None else n_local, rows=(rb, re))
  out, err, root_exc = None, None, None
  if group.rank == 0:
    try:
      gathered = getattr(ctx, "_gathered", None)
      if gathered is not None:                              # host-staged lists (gloo): upload + device DP
        ctx._gathered = None
        px, py = ctx.chain(*gathered, min_len=min_path_length(n_ve, n_ae))
      else:
        px, py = ctx.chain_resident(min_len=min_path_length(n_ve, n_ae))
      tm["device"]["chain_ms"] = ctx.stats()["chain_ms"]
      tm["chain_s"] = time.perf_counter() - t2
      fx, fy, a_scaled, v_scaled = _stage_pass1(px, py, video_features, audio_desc_features, tm)
      t3 = time.perf_counter()
      lp = solve_trend_lp(fx, fy, procs=procs)
      tm["lp_s"] = time.perf_counter() - t3
      tm["lp_method"] = lp["method"]
      tm["lp_helper_processes"] = lp["tree"].get("helper_processes", 0)
      out = _stage_refine(ctx, lp, a_scaled, v_scaled, n_ve, n_ae, tm)
    except BaseException as e:                               # whatever it is, the other ranks are waiting in the broadcast
      err = f"{type(e).__name__}: {e}" if not isinstance(e, RuntimeError) else str(e)
      err = err or type(e).__name__
      root_exc = e
  try:
    out = group.broadcast_result(out, err)
  except RuntimeError:
    if root_exc is not None and not isinstance(root_exc, RuntimeError):
      raise root_exc                                         # rank 0: the original exception, now that everyone has been told
    raise
  tm["total_s"] = time.perf_counter() - t0
  return out


RESIDENT_PCM = object()    # what a pipeline job returns when the pair's PCM is resident on the context it was given (see AlignPipeline._gpu_stage)


def _lp_worker(args):
  fx, fy = args
  t0 = time.perf_counter()
  lp = solve_trend_lp(fx, fy)
  lp.pop("solution", None)
  lp.pop("tree", None)
  return lp, time.perf_counter() - t0


def _cpu_topology(cpus=None):
  """(primary, secondary, domain): one hardware thread per physical core, their SMT siblings, and the L3
  domain (CCD) of every CPU this process may use.  Read from sysfs; without it, the usual Linux numbering
  (second half = siblings, eight cores to an L3) is assumed."""
  import os
  cpus = sorted(os.sched_getaffinity(0)) if cpus is None else sorted(cpus)
  allowed = set(cpus)

  def read_list(path):
    out = []
    for part in open(path).read().strip().split(","):
      if "-" in part:
        a, b = part.split("-"); out.extend(range(int(a), int(b) + 1))
      elif part:
        out.append(int(part))
    return out

  try:
    primary, secondary, domain = [], [], {}
    for c in cpus:
      sib = [x for x in read_list(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list") if x in allowed]
      (primary if c == min(sib) else secondary).append(c)
      try:
        domain[c] = min(read_list(f"/sys/devices/system/cpu/cpu{c}/cache/index3/shared_cpu_list"))
      except Exception:
        domain[c] = c // 8
  except Exception:
    half = max(1, len(cpus) // 2)
    primary, secondary = cpus[:half], cpus[half:]
    domain = {c: (c % half) // 8 for c in cpus}
  return primary, secondary, domain


def _smt_siblings(cpus):
  """cpu -> the set of hardware threads of its physical core (itself included), from sysfs; without sysfs the usual
  Linux numbering (second half of the list = siblings of the first half)."""
  out = {}
  try:
    for c in cpus:
      txt = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
      sibs = set()
      for part in txt.split(","):
        if "-" in part:
          a, b = part.split("-"); sibs.update(range(int(a), int(b) + 1))
        elif part:
          sibs.add(int(part))
      out[c] = sibs
  except Exception:
    cpus = sorted(cpus)
    half = max(1, len(cpus) // 2)
    out = {c: {c, cpus[(k + half) % len(cpus)]} for k, c in enumerate(cpus)}
  return out


def aux_core_order(cpus, worker_cores, topology=None, siblings=None):
  """(rest, dealt): the CPUs left for a pipeline's own threads once the LP workers' cores and their SMT siblings are set
  aside, and the physical cores among them in the order the threads take them -- L3 domains with the fewest LP workers
  first, one core per domain before a second one of any."""
  primary, _, domain = topology if topology is not None else _cpu_topology(cpus)
  sib = siblings if siblings is not None else _smt_siblings(cpus)
  taken = set()
  for c in worker_cores:
    taken.update(sib.get(c, {c}))
  rest = set(cpus) - taken
  load = {}
  for c in worker_cores:
    load[domain[c]] = load.get(domain[c], 0) + 1
  by_dom = {}
  for c in sorted(c for c in primary if c in rest):
    by_dom.setdefault(domain[c], []).append(c)
  doms = sorted(by_dom, key=lambda d: (load.get(d, 0), d))
  dealt, k, n = [], 0, sum(len(v) for v in by_dom.values())
  while len(dealt) < n:
    for d in doms:
      if k < len(by_dom[d]):
        dealt.append(by_dom[d][k])
    k += 1
  return rest, dealt


def cpu_order(cpus=None):
  """The CPUs this process may use, ordered so that the first K are the best K places for K
  single-threaded solver processes: one hardware thread per physical core first (a HiGHS solve runs
  2x slower next to a busy SMT sibling), dealt round-robin over the L3 domains (CCDs) so that
  neighbours share as little cache as possible; the second hardware threads come last."""
  primary, secondary, domain = _cpu_topology(cpus)

  def deal(group):
    buckets = {}
    for c in group:
      buckets.setdefault(domain[c], []).append(c)
    keys = sorted(buckets)
    out, k = [], 0
    while any(buckets[d] for d in keys):
      for d in keys:
        if k < len(buckets[d]):
          out.append(buckets[d][k])
      k += 1
      if k > len(group):
        break
    return out

  return deal(primary) + deal(secondary)


def _pin_worker(slot_counter, lock, first_slot, order):
  """Give every worker process its own physical core (see cpu_order), offset by the rank's
  `first_slot` so that ranks sharing a host do not pile onto the same cores."""
  import os
  try:
    with lock:
      k = slot_counter.value
      slot_counter.value += 1
    os.sched_setaffinity(0, {order[(first_slot + k) % len(order)]})
  except Exception:
    pass


def cpu_quota(fs_root="/"):
  """CPUs' worth of run time per second that this process's cgroup may use (cgroup v2 `cpu.max`, v1 `cpu.cfs_quota_us` /
  `cpu.cfs_period_us`; the tightest limit on the path from the process's cgroup to the root), or None without a limit
  (`fs_root`: where /proc and /sys are found -- tests point it at a directory of their own).
  The GPU box is a 2 x 64-core host, but its container runs under `cpu.max = 1600000 100000`: 16 CPUs' worth of time however
  many of the 256 hardware threads it spreads over (profiles/r06_host_cpu_quota_probe.txt: a pure register loop on 32 pinned
  processes already takes 1.8 x as long as on 16, on 128 10 x).  That quota -- not the L3 domains rounds 2-5 blamed -- is
  what flattened every worker sweep: the host stage delivers  quota / (CPU-seconds per pair)  pairs per second."""
  best = None

  def take(q):
    nonlocal best
    if q is not None and q > 0:
      best = q if best is None else min(best, q)

  def v2(d):
    try:
      a, b = open(os.path.join(d, "cpu.max")).read().split()[:2]
      return None if a == "max" else float(a) / float(b)
    except Exception:
      return None

  def v1(d):
    try:
      q = float(open(os.path.join(d, "cpu.cfs_quota_us")).read())
      per = float(open(os.path.join(d, "cpu.cfs_period_us")).read())
      return None if q <= 0 else q / per
    except Exception:
      return None

  rel2, rel1 = "", ""
  try:
    for line in open(os.path.join(fs_root, "proc/self/cgroup")):
      _, ctrl, path = line.strip().split(":", 2)
      if ctrl == "":
        rel2 = path
      elif "cpu" in ctrl.split(","):
        rel1 = path
  except Exception:
    pass
  for root, rel, read in ((os.path.join(fs_root, "sys/fs/cgroup"), rel2, v2), (os.path.join(fs_root, "sys/fs/cgroup/cpu"), rel1, v1),
                          (os.path.join(fs_root, "sys/fs/cgroup/cpu,cpuacct"), rel1, v1)):
    parts = [p for p in rel.split("/") if p]
    for k in range(len(parts), -1, -1):
      take(read(os.path.join(root, *parts[:k])))
  return best


def default_worker_count(local_world: int = 1) -> int:
  """LP worker processes per rank.  The host stage (pass 1 + HiGHS LP + clustering, one thread per pair) is bound by CPU TIME:
  `budget` = the cgroup's CPU quota (cpu_quota) or, without one, the physical cores this process may use; a rank's share is
  budget / ranks on the host.  Under a quota the share is oversubscribed by a quarter (the workers are the only consumers
  that matter -- GPU feeder, refine and hand-off threads take ~0.3 CPU-seconds of a 2 h pair's 6.4 -- and a worker that waits
  for its next pair leaves quota unused), on the GPU box 16 -> 20.  Without a quota: three workers per four cores of the
  share, at most 64 per rank (every worker in flight pins a /dev/shm block of 10-70 MB and the pipeline window is two pairs
  per worker).  Never fewer than 2, never more than one per physical core of the share."""
  primary, _, _ = _cpu_topology()
  local_world = max(1, int(local_world))
  cores = max(1, len(primary))
  quota = cpu_quota()
  if quota is not None and quota < cores:
    want = int(np.ceil(1.25 * quota / local_world))
  else:
    want = min(64, (3 * cores) // (4 * local_world))
  return int(max(2, min(want, max(1, cores // local_world))))


# ---- worker-process side of the batch pipeline ---------------------------------------------------
# Worker processes never touch the GPU (a second process with a device context makes the GPU
# time-slice between processes, which costs far more than it gains): they run the host-only
# stages -- pass-1 numpy, the HiGHS LP, clustering -- each under its own GIL.

def _block_layout(n, le_v, lo_v, le_a, lo_a):
  """Byte offsets of one pair's shared block: the stage-2 path (n points), feature rows, scaled stacks."""
  off, lay = 0, {}
  la, lv = min(le_a, lo_a), min(le_v, lo_v)
  for name, count, size in (("a_scaled", 3 * la, 8), ("v_scaled", 3 * lv, 8), ("px", n, 4), ("py", n, 4),
                            ("vf0", le_v, 4), ("vf", 4 * lo_v, 4), ("af0", le_a, 4), ("af", 4 * lo_a, 4)):
    lay[name] = (off, count)
    off += ((count * size + 63) // 64) * 64
  return lay, max(off, 64)


def _block_views(buf, lay, le_v, lo_v, le_a, lo_a):
  def arr(name, dtype, shape=None):
    o, c = lay[name]
    a = np.frombuffer(buf, dtype=dtype, count=c, offset=o)
    return a if shape is None else a.reshape(shape)
  vf = [arr("vf0", np.float32)] + list(arr("vf", np.float32, (4, lo_v)))
  af = [arr("af0", np.float32)] + list(arr("af", np.float32, (4, lo_a)))
  return (arr("px", np.int32), arr("py", np.int32), vf, af,
          arr("a_scaled", np.float64, (-1, 3)), arr("v_scaled", np.float64, (-1, 3)))


def _proc_mid(fname, fsize, n, le_v, lo_v, le_a, lo_a):
  """Host-only middle of the pipeline for one pair: pass 1, LP, clustering."""
  lay, size = _block_layout(n, le_v, lo_v, le_a, lo_a)
  mm = np.memmap(fname, dtype=np.uint8, mode="r+", shape=(fsize,))
  px, py, vf, af, a_out, v_out = _block_views(mm, lay, le_v, lo_v, le_a, lo_a)
  tm = {}
  t_in = time.perf_counter()
  c_in = time.process_time()
  fx, fy, a_s, v_s = _stage_pass1(px, py, vf, af, tm)
  t0 = time.perf_counter()
  lp = solve_trend_lp(fx, fy)
  tm["lp_s"] = time.perf_counter() - t0
  tm["lp_method"] = lp["method"]
  t1 = time.perf_counter()
  clusters = cluster_lines(lp["smooth_x"], lp["smooth_y"], lp["slopes"])
  tm["cluster_s"] = time.perf_counter() - t1
  a_out[:] = a_s
  v_out[:] = v_s
  mm.flush()
  del mm
  tm["worker_s"] = time.perf_counter() - t_in
  tm["worker_cpu_s"] = time.process_time() - c_in          # CPU time, not wall: what the pair costs of the host's CPU-time budget
  return clusters, float(lp["median_slope"]), tm


class AlignPipeline:
  """Directory-batch throughput.  One pair's latency is dominated by host work (the
  single-threaded HiGHS LP), so pairs are pipelined:

    GPU threads       (one da_ctx / HIP stream each) features + matching of successive pairs:
                      prep, similarity GEMM, verification, sort; then the pair's chain DP is
                      enqueued on a stream of its own (da_chain_begin: one persistent workgroup,
                      several pairs' DPs run beside the GEMMs of later pairs) and collected when it
                      has finished -- the match list never leaves the device, only the path does
    worker processes  (own GIL, no device) per pair: pass-1 host -> LP (HiGHS) -> clustering; they
                      map the pair's /dev/shm block (path + feature rows), nothing big is pickled
    refine threads    (one da_ctx each, this process) banded extension kernels + second DP + nodes

  A job is either a tuple (video_features, audio_features) or a callable job(ctx) returning that
  tuple (e.g. running the feature kernel on that context's resident PCM).  Results come back in
  submission order and are identical to align()'s.

      with AlignPipeline([ctx0, ctx1], lp_workers=16) as pipe:
        for result in pipe.run(jobs):
          ...
  """

  def __init__(self, ctx=None, lp_workers=4, mode=_native.MATCH_HASHED, refine_threads=4):
    import concurrent.futures as cf
    import multiprocessing as mp
    import os
    import tempfile
    import threading
    if ctx is None:
      ctx = default_context()
    self.gpu_ctxs = list(ctx) if isinstance(ctx, (list, tuple)) else [ctx]
    self.ctx = self.gpu_ctxs[0]
    self.mode = mode
    self.depth = max(1, int(lp_workers))
    # worker processes inherit the environment: one BLAS/OpenMP thread each, or dozens of workers
    # oversubscribe the host with their numpy thread pools and every LP slows down
    saved = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS")}
    for k in saved:
      os.environ[k] = "1"
    mpc = mp.get_context("spawn")
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    pin = int(os.environ.get("DALIGN_PIN_WORKERS", "1")) and ncpu >= 2 * self.depth * local_world
    init, initargs = (None, ())
    order = cpu_order() if pin else []
    if pin:
      # rank r of a host takes slots r, r + local_world, r + 2 local_world, ... of the common order:
      # every rank's workers are spread over all L3 domains and no two ranks share a core
      mine = order[local_rank::local_world] if local_world > 1 else order
      init, initargs = _pin_worker, (mpc.Value("i", 0), mpc.Lock(), 0, mine)
    self.pool = cf.ProcessPoolExecutor(max_workers=self.depth, mp_context=mpc, initializer=init, initargs=initargs)
    list(self.pool.map(int, range(self.depth)))          # spawn them now, under that environment
    for k, v in saved.items():
      if v is None:
        os.environ.pop(k, None)
      else:
        os.environ[k] = v
    # keep this process's own threads (GPU feeder, refine pool, hand-off, the caller) off the cores the LP workers of
    # EVERY rank of this host are pinned to and off their SMT siblings; threads created below inherit the mask.  The
    # pools' threads are then pinned one per core (`_aux_cores`: physical cores in the L3 domains with the fewest LP
    # workers first), so that the second DP -- a pointer chase through ~270 MB per 2 h pair -- and the hand-off copies do
    # not migrate between domains from pair to pair (round 4: the same binary ran the second DP 65 % slower on one host).
    self._old_affinity = None
    self._aux_cores, self._aux_next, self._pin_lock = [], 0, threading.Lock()
    self._floating, self._pinned_tids, self._moved = None, set(), {}
    if pin and int(os.environ.get("DALIGN_PIN_MAIN", "1")):
      try:
        cpus = sorted(os.sched_getaffinity(0))
        rest, dealt = aux_core_order(cpus, order[:self.depth * local_world])
        if len(rest) >= 8 * local_world:
          mine_aux = dealt[local_rank::local_world] if local_world > 1 else dealt
          self._old_affinity = set(cpus)
          if local_world == 1:
            os.sched_setaffinity(0, rest)
            self._rest = set(rest)
          if int(os.environ.get("DALIGN_PIN_THREADS", "1")):
            self._aux_cores = mine_aux
            # everything else of this process -- the caller's thread, the HIP runtime's helper threads -- floats over the cores
            # that are NEITHER an LP worker's NOR one of the pinned threads' (_float_others, called again once the contexts exist:
            # a helper thread that the runtime starts from a pinned thread inherits that thread's single core)
            n_pinned = len(self.gpu_ctxs) + max(1, int(refine_threads)) + 2
            floating = set(rest) - set(mine_aux[:n_pinned])
            if local_world == 1 and len(floating) >= 8:
              self._floating = floating
      except Exception:
        self._old_affinity = None
    self.gpu_threads = [cf.ThreadPoolExecutor(max_workers=1, initializer=self._pin_thread) for _ in self.gpu_ctxs]
    self.refine_pool = cf.ThreadPoolExecutor(max_workers=max(1, int(refine_threads)), initializer=self._pin_thread)
    self.handoff_pool = cf.ThreadPoolExecutor(max_workers=2, initializer=self._pin_thread)      # copies path + feature rows into the workers' /dev/shm blocks
    self._local = threading.local()
    self._ctxs = []
    self._lock = threading.Lock()
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    self._dir = tempfile.mkdtemp(prefix=f"dalign_{os.getpid()}_", dir=shm)
    self._seq = 0
    self._free = []        # reusable (file name, bytes) blocks: fresh tmpfs pages cost ~0.2 ms per MB
    self.pace = int(os.environ.get("DALIGN_PIPELINE_PACE", "1")) != 0
    self._work_ema = None  # worker seconds per pair (pass 1 + LP + clustering), exponential average
    self._last_admit = 0.0
    self._last_lp_start = 0.0
    self._lp_running = 0   # pairs handed to the worker pool (or about to be) and not yet back
    self._stagger = False  # set by run(expected=...): long batches only
    self._n_admitted = 0   # pairs whose GPU stage has started / pairs whose worker stage has finished
    self._n_done = 0
    self._queued = {}      # per GPU context: pairs submitted to its thread that have not started yet
    self._chains = {}      # per GPU context: pairs whose chain DP is enqueued (ticket, ...), oldest first
    self.max_chains = int(os.environ.get("DALIGN_MAX_CHAINS", "6"))       # DPs in flight per context (the library allows 16; a DP takes 10-70 ms)
    import sys
    self._old_switch = sys.getswitchinterval()
    sys.setswitchinterval(2e-4)

  def _pin_thread(self):
    """Initializer of the pipeline's own threads: each takes the next core of `_aux_cores` for itself."""
    import threading
    with self._pin_lock:
      if not self._aux_cores or self._aux_next >= len(self._aux_cores):
        return
      core = self._aux_cores[self._aux_next]
      self._aux_next += 1
    tid = threading.get_native_id()
    with self._pin_lock:
      self._pinned_tids.add(tid)          # BEFORE the mask changes: _float_others must never move a pinned thread back
    try:
      os.sched_setaffinity(tid, {core})
    except Exception:
      with self._pin_lock:
        self._pinned_tids.discard(tid)

  def _float_others(self):
    """Every thread of this process that is not one of the pinned pool threads goes onto the floating cores."""
    if not self._floating:
      return
    try:
      tids = [int(t) for t in os.listdir("/proc/self/task")]
    except Exception:
      return
    for tid in tids:
      with self._pin_lock:
        if tid in self._pinned_tids:
          continue
      try:
        if tid not in self._moved:          # remembered once, restored by __exit__: the embedding application's threads are not ours to keep confined
          self._moved[tid] = os.sched_getaffinity(tid)
        os.sched_setaffinity(tid, self._floating)
      except Exception:
        pass

  def _thread_ctx(self):
    c = getattr(self._local, "ctx", None)
    if c is None:
      c = _native.Context(self.ctx.device, self.ctx.precision)
      self._local.ctx = c
      with self._lock:
        self._ctxs.append(c)
    return c

  def warm(self):
    """Start the worker processes (imports) and refine threads before anything is timed."""
    x = np.arange(40.0)
    list(self.pool.map(_lp_worker, [(x, x + 0.25 * np.sin(x))] * self.depth))
    list(self.refine_pool.map(lambda _: self._thread_ctx(), range(self.refine_pool._max_workers)))
    for g in self.gpu_threads:                      # start (and pin) the GPU threads, then move every other thread off their cores
      g.submit(lambda: None).result()
    list(self.handoff_pool.map(lambda _: time.sleep(0.01), range(2)))
    self._float_others()

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    import shutil
    import sys
    for tid, mask in list(getattr(self, "_moved", {}).items()):       # threads _float_others moved (runtime helpers, the caller's own)
      if getattr(self, "_old_affinity", None) and mask == getattr(self, "_rest", None):
        mask = self._old_affinity                                   # it had inherited this pipeline's own narrowing of the caller's mask
      try:
        os.sched_setaffinity(tid, mask)
      except Exception:
        pass                                                        # the thread has ended meanwhile
    self._moved = {}
    if getattr(self, "_old_affinity", None):
      try:
        os.sched_setaffinity(0, self._old_affinity)
      except Exception:
        pass
      self._old_affinity = None
    for g in self.gpu_threads:
      g.shutdown(wait=True, cancel_futures=True)
    # the hand-off threads submit to self.pool: let them finish (or fail) before the workers go away
    self.handoff_pool.shutdown(wait=True, cancel_futures=True)
    self.pool.shutdown(wait=True, cancel_futures=True)
    self.refine_pool.shutdown(wait=True, cancel_futures=True)
    for c in self._ctxs:
      c.close()
    self._ctxs = []
    shutil.rmtree(self._dir, ignore_errors=True)
    sys.setswitchinterval(self._old_switch)

  def _gpu_stage(self, ctx, job, tm, fname, done):
    """Features + matching of one pair on this GPU thread, then its chain DP is enqueued on its own
    stream.  Finished DPs of earlier pairs are collected and handed to the worker processes between
    match_begin and match_finish, i.e. under the GEMM."""
    try:
      with self._lock:
        self._queued[id(ctx)] = self._queued.get(id(ctx), 0) - 1
      tp = time.perf_counter()
      c_gpu = time.thread_time()
      self._pace(ctx)
      t0 = time.perf_counter()
      tm["pace_s"] = t0 - tp
      got = job(ctx) if callable(job) else job
      if got is RESIDENT_PCM:
        # the pair's PCM is resident on this context: everything up to the enqueued chain DP in one native call.  Finished DPs
        # of earlier pairs are handed on first (they used to be collected under the GEMM; one GEMM later costs nothing: the
        # LP stage behind them takes 20-200 x a GEMM)
        self._collect_chains(ctx, block_above=self.max_chains - 1)
        tm["features_s"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        vf, af, n, ticket = ctx.pair_stage(self.mode)
        tm["device"] = ctx.stats()
        tm["match_s"] = time.perf_counter() - t1
        tm["match_begin_s"], tm["collect_under_gemm_s"], tm["match_finish_s"], tm["chain_begin_s"] = 0.0, 0.0, tm["match_s"], 0.0
        tm["n_matches"] = n
      else:
        vf, af = got
        tm["features_s"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        ctx.match_begin(vf, af, self.mode)
        ta = time.perf_counter()
        self._collect_chains(ctx, block_above=self.max_chains - 1)
        tb = time.perf_counter()
        n = ctx.match_finish()
        tc = time.perf_counter()
        tm["device"] = ctx.stats()
        tm["match_s"] = time.perf_counter() - t1
        tm["match_begin_s"], tm["collect_under_gemm_s"], tm["match_finish_s"] = ta - t1, tb - ta, tc - tb
        tm["n_matches"] = n
        t2 = time.perf_counter()
        ticket = ctx.chain_begin()
        tm["chain_begin_s"] = time.perf_counter() - t2
      tm["t_gpu_stage_end"] = time.perf_counter()
      tm["gpu_thread_cpu_s"] = time.thread_time() - c_gpu     # the feeding thread's own CPU time for this pair (polling waits included)
      self._stages_done = getattr(self, "_stages_done", 0) + 1
      if self._stages_done in (1, 4, 16):             # helper threads the runtime started from this (pinned) thread meanwhile
        self._float_others()
      self._chains.setdefault(id(ctx), []).append((ticket, vf, af, tm, fname, done, time.perf_counter()))
    except BaseException as e:
      with self._lock:
        self._n_done += 1
      done.set_exception(e)

  def _pace(self, ctx):
    """Admission pacing.  When the host stage is the slower one (a 2 h pair: ~10 s of LP per pair on one
    of `depth` workers against 0.3 s on the GPU) pairs are admitted `worker seconds per pair / depth`
    apart instead of as fast as the GPU can take them: the workers then start -- and finish -- evenly
    spread in time, the queue in front of them (and the /dev/shm blocks it pins) stays short, and the
    pipeline's output rate is the sustainable one from the first generation of solves on, not a burst
    at the GPU's rate followed by a stall.  No effect while the GPU stage is the slower one."""
    if not self.pace:
      return
    waited = False
    while True:
      with self._lock:
        ema, in_flight = self._work_ema, self._n_admitted - self._n_done
      if ema is None:
        # nothing has come back yet: no queue in front of the workers until their speed is known
        wait = 0.004 if in_flight >= self.depth * max(1, len(self.gpu_ctxs)) else 0.0
      else:
        wait = self._last_admit + 0.97 * ema / (self.depth * max(1, len(self.gpu_ctxs))) - time.perf_counter()
      if wait <= 0:
        break
      waited = True
      self._collect_chains(ctx)                          # finished DPs are still handed on while waiting
      time.sleep(min(wait, 0.004))
    if waited and not ctx.chain_masked():
      # (only when the chain DPs' streams carry no CU mask: DALIGN_CHAIN_CUS = 0 / off, or a runtime that refused it)  host-bound: the GPU has time to spare, so the chain DPs
      # still in flight are let finish before the next similarity GEMM starts -- spread over the chip, a DP's column waves keep
      # whole CUs from taking GEMM workgroups (+25 % GEMM time).  Confined to 8 CUs per XCD (the default) a DP costs the GEMM
      # nothing, and waiting for it here only serialised the two: in a batch near the balance of GPU and LP stage (configs[1])
      # the pacing waits a millisecond now and then, and each time the next pair's kernels started 6 ms late -- behind the DP.
      self._collect_chains(ctx, block_above=0)
    with self._lock:
      self._n_admitted += 1
    self._last_admit = time.perf_counter()

  def _collect_chains(self, ctx, block_above=None):
    """Hand every pair whose chain DP has finished to the worker processes, in order.  With
    block_above = k, wait for the oldest ones until at most k DPs are outstanding (k = 0: drain)."""
    queue = self._chains.get(id(ctx), [])
    while queue:
      must = block_above is not None and len(queue) > block_above
      if not must and not ctx.chain_done(queue[0][0]):
        break
      self._hand_off(ctx, *queue.pop(0))

  def _collect_idle(self, ctx):
    """Runs on the GPU thread after every pair: while no further pair is waiting for this context,
    wait for its outstanding chain DPs one at a time (otherwise they are collected under the next
    pair's GEMM) -- so a caller that stops submitting until results arrive cannot starve them."""
    # (polling, not da_chain_finish's blocking wait: the next pair may be submitted a millisecond from now -- the caller
    # submits one whenever a result is delivered -- and a thread blocked on a 6-50 ms DP would start that pair's kernels only
    # behind it: a kernel trace of the configs[1] batch showed every pair's feature kernels 6 ms late, right behind the previous
    # pair's DP, profiles/r05_trace_cfg1_gaps.json)
    while self._chains.get(id(ctx)) and self._queued.get(id(ctx), 0) <= 0:
      queue = self._chains[id(ctx)]
      if ctx.chain_done(queue[0][0]):
        self._hand_off(ctx, *queue.pop(0))
      else:
        time.sleep(0.0003)

  def _hand_off(self, ctx, ticket, vf, af, tm, fname, done, t_begin):
    """Collect a finished chain DP (on the thread that owns the context) and pass the pair on; the copy into
    the workers' shared-memory block (10-70 MB: path + ten feature rows) and the submission run on a helper
    thread, so the GPU-feeding thread goes straight back to the next pair's kernels."""
    try:
      dims = (len(vf[0]), len(vf[1]), len(af[0]), len(af[1]))
      px, py = ctx.chain_finish(ticket, min_len=min_path_length(dims[0], dims[2]))      # raises the mismatch error (:698)
      tm["device"]["chain_ms"] = ctx.stats()["chain_ms"]
      tm["chain_s"] = time.perf_counter() - t_begin          # enqueue -> collected (includes waiting in the queue)
    except BaseException as e:
      with self._lock:
        self._n_done += 1
      done.set_exception(e)
      return
    self.handoff_pool.submit(self._hand_off_copy, px, py, vf, af, dims, tm, fname, done)

  def _hand_off_copy(self, px, py, vf, af, dims, tm, fname, done):
    try:
      tm["t_handoff"] = time.perf_counter()           # path collected, copy about to start (interval stamps: where a pair waits)
      c_hand = time.thread_time()
      n = len(px)
      state = {}
      lay, size = _block_layout(n, *dims)
      with self._lock:
        pick = next((b for b in self._free if b[1] >= size), None)
        if pick is not None:
          self._free.remove(pick)
      if pick is not None:
        state["fname"], fsize = pick
        mm = np.memmap(state["fname"], dtype=np.uint8, mode="r+", shape=(fsize,))
      else:
        fsize = size + size // 4          # slack so the next pair of similar size fits too
        state["fname"] = fname
        mm = np.memmap(fname, dtype=np.uint8, mode="w+", shape=(fsize,))
      state.update(mm=mm, lay=lay, n=n, fsize=fsize)
      bx, by, bvf, baf, _, _ = _block_views(mm, lay, *dims)
      bx[:] = px; by[:] = py
      for dst, src in zip(bvf, vf):
        dst[:] = src
      for dst, src in zip(baf, af):
        dst[:] = src
      mm.flush()
      tm["t_copied"] = time.perf_counter()
      tm["handoff_cpu_s"] = time.thread_time() - c_hand
      self._stagger_start(n)
      tm["t_submitted"] = time.perf_counter()
      try:
        mid = self.pool.submit(_proc_mid, state["fname"], fsize, n, *dims)
      except BaseException:
        with self._lock:
          self._lp_running -= 1
        raise

      def on_mid(f):
        tm["t_worker_back"] = time.perf_counter()
        with self._lock:
          self._n_done += 1               # the pair has left the worker stage (whatever the outcome)
          self._lp_running -= 1

        def work():
          try:
            done.set_result(self._refine_stage(f.result(), state, dims, tm))
          except BaseException as e:          # surfaces in finish()
            done.set_exception(e)
        self.refine_pool.submit(work)

      mid.add_done_callback(on_mid)
    except BaseException as e:
      with self._lock:
        self._n_done += 1
      done.set_exception(e)

  def _stagger_start(self, n_path):
    """Worker-start staggering (long batches only: `run(expected=...)` of at least two generations of solves).
    The GPU stage delivers a pair every 0.15 s, a 2 h pair's LP takes 6-10 s: left alone, all `depth` workers start
    their first solve within a few seconds of each other, finish together, take the next queued pairs together --
    a limit cycle in which results leave the pipeline in bursts of `depth` (20 results within 3 s, then nothing for
    7 s) and every generation of solves hits the host's caches at once.  Consecutive solve STARTS are therefore kept
    `worker seconds per pair / depth` apart, and a pair is only handed to the pool when a worker is free for it: the
    phases of the workers end up evenly spread over one solve time and stay so; in steady state the spacing equals the
    pool's natural rate, so it costs nothing.  Before the first solve has come back its duration is estimated from the
    path length (fit points ~ n / 14; HiGHS' dual simplex on this LP: ~0.6 s at 2 400 points, growing like n^1.8)."""
    while True:
      with self._lock:
        if not (self.pace and self._stagger):
          self._lp_running += 1
          return
        ema = self._work_ema
        if ema is None:
          ema = 0.6 * (max(1.0, n_path / 14.0) / 2400.0) ** 1.8
        now = time.perf_counter()
        wait = self._last_lp_start + 0.97 * ema / self.depth - now
        if self._lp_running < self.depth and wait <= 0:
          self._lp_running += 1
          self._last_lp_start = now
          return
      time.sleep(min(max(wait, 0.002), 0.02))

  def _refine_stage(self, mid_result, state, dims, tm):
    tm["t_refine_start"] = time.perf_counter()
    c_ref = time.thread_time()
    clusters, med, wtm = mid_result
    busy = wtm.get("worker_s")
    if busy is not None:
      with self._lock:
        self._work_ema = busy if self._work_ema is None else 0.8 * self._work_ema + 0.2 * busy
    dev = tm.get("device", {})
    dev.update(wtm.pop("device", {}))
    tm.update(wtm); tm["device"] = dev
    ctx = self._thread_ctx()
    _, _, _, _, a_s, v_s = _block_views(state["mm"], state["lay"], *dims)
    out = _stage_refine(ctx, dict(median_slope=med), a_s, v_s, dims[0], dims[2], tm, clusters=clusters)
    tm["done_t"] = time.perf_counter()          # completion time (results are DELIVERED in submission order, later)
    tm["refine_cpu_s"] = time.thread_time() - c_ref
    del a_s, v_s
    mm = state.pop("mm")
    del mm
    with self._lock:
      self._free.append((state["fname"], state["fsize"]))
    return out

  def run(self, jobs, timings=None, window=None, expected=None):
    """Yields the results in submission order.  `window` = pairs admitted before the oldest result
    is waited for (default 2 x workers + 4 per GPU, or DALIGN_PIPELINE_WINDOW): callers that hold
    per-pair host memory until a pair's result arrives (combine --stretch_audio) narrow it.
    `expected` = number of pairs the caller is going to submit, when it knows: from two generations of solves
    on, the workers' solve starts are staggered (_stagger_start)."""
    self._stagger = expected is not None and expected >= 2 * self.depth
    import concurrent.futures as cf
    import os
    pending = []          # (future of the pair's result, tm) in submission order
    n_gpu = len(self.gpu_ctxs)
    if window is None:
      window = int(os.environ.get("DALIGN_PIPELINE_WINDOW", 2 * self.depth + 4 * n_gpu))
    window = max(1, int(window))

    def finish(entry):
      done, tm = entry
      out = done.result()
      if timings is not None:
        timings.append(tm)
      return out

    for k, job in enumerate(jobs):
      tm = {}
      g = k % n_gpu
      fname = os.path.join(self._dir, f"pair{self._seq}.bin")
      self._seq += 1
      done = cf.Future()
      with self._lock:
        self._queued[id(self.gpu_ctxs[g])] = self._queued.get(id(self.gpu_ctxs[g]), 0) + 1
      self.gpu_threads[g].submit(self._gpu_stage, self.gpu_ctxs[g], job, tm, fname, done)
      self.gpu_threads[g].submit(self._collect_idle, self.gpu_ctxs[g])
      pending.append((done, tm))
      # results are handed back in submission order, so pairs that finished behind a slow LP still
      # count as pending: the window has to be wider than the worker pool or workers idle
      while len(pending) >= window:
        yield finish(pending.pop(0))
    for g in range(n_gpu):          # the chain DPs still in flight have to be collected
      self.gpu_threads[g].submit(self._collect_chains, self.gpu_ctxs[g], 0)
    while pending:
      yield finish(pending.pop(0))
