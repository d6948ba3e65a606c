"""Host LP of the alignment path (describealign.py:769-858): the tree of warm starts (describealign_amd/lp_tree.py) must return
the optimum the reference's own `scipy.optimize.linprog` call returns, certified against the LP as the reference poses it, and
must step aside for that call whenever anything is off.  The instances are the fit points of synthetic pairs as the GPU stages
deliver them (tests/golden/lp/*.npz, written by tests/gpu_dump_fit_points.py on the GPU box): e1320 = configs[1]'s stand-in,
h0 = seed 0 of configs[3], e3600 / e7200s = the one- and two-hour pairs of the goldens, r7200 = a 2 h pair with a 0.3 % rate
difference."""
import os

import numpy as np
import pytest
import scipy.optimize

from describealign_amd import align as A
from describealign_amd import lp_tree as T

LP_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lp")
needs_binding = pytest.mark.skipif(not T.available(), reason="scipy without the _highspy binding: the product makes the reference's call")


def _instance(name):
  z = np.load(os.path.join(LP_DIR, name + ".npz"))
  return z["fx"], z["fy"]


def _bounds_arrays(n, n_col):
  lb = np.zeros(n_col); ub = np.full(n_col, np.inf)
  ub[4 * n - 2:6 * n - 2] = 2.0
  lb[-1] = -np.inf
  return lb, ub


def test_sub_lp_is_the_full_lp_restricted_to_a_run_of_points():
  """lp_tree.assemble on a slice of the path = the rows and columns of the whole LP that belong to that slice, coefficient for
  coefficient (the leaves and merges solve pieces of the reference's LP, not approximations of it)."""
  x, y = _instance("e600")
  n = len(x)
  c, Afull, b, bounds = A.build_trend_lp(x, y)
  jc = A.trend_jump_cost(x, y)
  assert np.array_equal(c[2 * n:3 * n - 1], jc)
  a, e = 400, 733
  nw = e - a
  cw, Aw, bw, lbw, ubw = T.assemble(x[a:e], y[a:e], jc[a:e - 1])
  sizes_c, sizes_r = T.block_sizes(nw)
  full_c, full_r = T.block_sizes(n)
  cols, at = [], 0
  for k, (sz, fsz) in enumerate(zip(sizes_c, full_c)):
    cols.append(at + (a if k < 12 else 0) + np.arange(sz)); at += fsz
  rows, at = [], 0
  for sz, fsz in zip(sizes_r, full_r):
    rows.append(at + a + np.arange(sz)); at += fsz
  cols = np.concatenate(cols); rows = np.concatenate(rows)
  sub = Afull.tocsr()[rows][:, cols].toarray()
  assert np.array_equal(sub, Aw.toarray()) and np.array_equal(cw, c[cols]) and np.array_equal(bw, b[rows])
  assert np.array_equal(lbw[:-1], np.zeros(len(cw) - 1)) and lbw[-1] == -np.inf and np.all(ubw[4 * nw - 2:6 * nw - 2] == 2.0)
  # slope_shift only moves the right-hand side of the slope rows
  _, A2, b2, _, _ = T.assemble(x[a:e], y[a:e], jc[a:e - 1], 1.25)
  assert np.array_equal(A2.toarray(), Aw.toarray()) and np.allclose(b2[:nw - 1], bw[:nw - 1] - 1.25) and np.array_equal(b2[nw - 1:], bw[nw - 1:])


@needs_binding
@pytest.mark.parametrize("name", ["e1320", "h0", "e3600"])
def test_tree_returns_the_reference_calls_optimum(name):
  """Same LP, same solver, another starting basis: the solution vector, the slopes and the smooth path equal the ones
  `linprog(method='highs-ds')` returns from the slack basis (to the solver's own tolerances), and the result carries its
  optimality certificate for the LP as posed."""
  x, y = _instance(name)
  n = len(x)
  c, Am, b, bounds = A.build_trend_lp(x, y)
  ref = scipy.optimize.linprog(c, A_eq=Am, b_eq=b, bounds=bounds, method="highs-ds")
  assert ref.success
  lp = A.solve_trend_lp(x, y, tree=True)
  assert lp["method"] == "tree", lp["method"]
  assert np.max(np.abs(lp["solution"] - ref.x)) < 1e-6
  assert abs(lp["median_slope"] - ref.x[-1]) < 1e-12
  assert abs(float(c @ lp["solution"]) - ref.fun) < 1e-7 * max(1.0, abs(ref.fun))
  cert = lp["tree"]["certificate"]
  assert cert["primal_infeasibility"] < 1e-6 and cert["dual_infeasibility"] < 1e-6 and cert["relative_gap"] < 1e-6
  plain = A.solve_trend_lp(x, y, tree=False)
  assert plain["method"] == "reference" and np.array_equal(plain["solution"], ref.x)
  assert np.max(np.abs(lp["smooth_y"] - plain["smooth_y"])) < 1e-6 and np.max(np.abs(lp["slopes"] - plain["slopes"])) < 1e-9
  # ... and therefore the same line clusters go to pass 2
  got = A.cluster_lines(lp["smooth_x"], lp["smooth_y"], lp["slopes"])
  want = A.cluster_lines(plain["smooth_x"], plain["smooth_y"], plain["slopes"])
  assert all(g.shape == w.shape for g, w in zip(got, want))
  assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
  assert np.max(np.abs(got[2] - want[2])) < 1e-6 and np.max(np.abs(got[3] - want[3])) < 1e-9


@needs_binding
def test_certificate_accepts_the_reference_solution_and_rejects_others():
  x, y = _instance("e600")
  n = len(x)
  c, Am, b, bounds = A.build_trend_lp(x, y)
  lb, ub = _bounds_arrays(n, len(c))
  ref = scipy.optimize.linprog(c, A_eq=Am, b_eq=b, bounds=bounds, method="highs-ds")
  ok, worst = T.kkt_certificate(c, Am, b, lb, ub, ref.x, ref.eqlin.marginals)
  assert ok, worst
  # a feasible but not optimal point (the fit passes one frame above): dual side / gap must object
  off = ref.x.copy()
  off[:n] += 1.0                                         # fit_err+ of every point one larger: still feasible (differences unchanged)
  ok, worst = T.kkt_certificate(c, Am, b, lb, ub, off, ref.eqlin.marginals)
  assert not ok and worst["relative_gap"] > 1e-3 and worst["primal_infeasibility"] < 1e-9
  # an infeasible point
  bad = ref.x.copy(); bad[2 * n + 5] += 0.5              # one jump moved: its slope row no longer holds
  ok, worst = T.kkt_certificate(c, Am, b, lb, ub, bad, ref.eqlin.marginals)
  assert not ok and worst["primal_infeasibility"] > 1e-5
  # the right point with wrong multipliers
  ok, worst = T.kkt_certificate(c, Am, b, lb, ub, ref.x, 0.5 * ref.eqlin.marginals)
  assert not ok


def test_tree_steps_aside_for_the_reference_call(monkeypatch):
  """No binding, a short path, a sub-LP that does not end optimal, a failed certificate, an exception: each time the result is
  the reference's own linprog call, bit for bit."""
  x, y = _instance("e600")
  c, Am, b, bounds = A.build_trend_lp(x, y)
  ref = scipy.optimize.linprog(c, A_eq=Am, b_eq=b, bounds=bounds, method="highs-ds")
  lp = A.solve_trend_lp(x, y)                             # 1 588 fit points: at the threshold's mercy, either way the optimum
  assert np.max(np.abs(lp["solution"] - ref.x)) < 1e-6
  monkeypatch.setattr(T, "MIN_POINTS", 10 ** 9)
  lp = A.solve_trend_lp(x, y)
  assert lp["method"].startswith("reference (tree: fewer than") and np.array_equal(lp["solution"], ref.x)
  monkeypatch.setattr(T, "MIN_POINTS", 100)
  monkeypatch.setattr(T, "available", lambda: False)
  lp = A.solve_trend_lp(x, y)
  assert "not usable" in lp["method"] and np.array_equal(lp["solution"], ref.x)
  if not T._core:
    return
  monkeypatch.setattr(T, "available", lambda: True)
  monkeypatch.setattr(T, "_run", lambda *a, **k: None)
  lp = A.solve_trend_lp(x, y)
  assert "did not end optimal" in lp["method"] and np.array_equal(lp["solution"], ref.x)
  monkeypatch.undo()
  monkeypatch.setattr(T, "MIN_POINTS", 100)
  refused = {"primal_infeasibility": 0.0, "dual_infeasibility": 1.0, "relative_gap": 0.0}       # the multipliers: one refactorisation is tried first
  monkeypatch.setattr(T, "kkt_certificate", lambda *a, **k: (False, dict(refused)))
  lp = A.solve_trend_lp(x, y)
  assert "certificate failed" in lp["method"] and np.array_equal(lp["solution"], ref.x) and lp["tree"].get("refactored") is True
  refused["relative_gap"] = 1.0                                                                  # a gap: no second chance
  lp = A.solve_trend_lp(x, y)
  assert "certificate failed" in lp["method"] and np.array_equal(lp["solution"], ref.x) and "refactored" not in lp["tree"]
  monkeypatch.setattr(T, "solve", lambda *a, **k: (_ for _ in ()).throw(ValueError("boom")))
  lp = A.solve_trend_lp(x, y)
  assert "ValueError: boom" in lp["method"] and np.array_equal(lp["solution"], ref.x)
  # DALIGN_LP_TREE=0: never tried
  monkeypatch.undo()
  monkeypatch.setenv("DALIGN_LP_TREE", "0")
  lp = A.solve_trend_lp(x, y)
  assert lp["method"] == "reference" and np.array_equal(lp["solution"], ref.x)


@needs_binding
def test_tree_on_a_rate_changed_pair_and_odd_tree_shapes():
  """The slope held below the root starts from a data estimate (here ~1.003) and is re-centred level by level; other leaf
  sizes and fan-ins 2 and 3 (the tree is laid out from the top: 2 x fan^k leaves): same optimum every time."""
  x, y = _instance("r7200")
  x, y = x[:4100], y[:4100]
  c, Am, b, bounds = A.build_trend_lp(x, y)
  ref = scipy.optimize.linprog(c, A_eq=Am, b_eq=b, bounds=bounds, method="highs-ds")
  jc = A.trend_jump_cost(x, y)
  assert abs(T.estimate_slope(x, y) - ref.x[-1]) < 1e-3
  for leaf, fan in ((320, 4), (455, 2), (333, 3)):
    os.environ["DALIGN_LP_FAN"] = str(fan)
    try:
      st = {}
      sol, row_dual, _ = T.solve(x, y, jc, leaf_points=leaf, stats=st)
    finally:
      os.environ.pop("DALIGN_LP_FAN", None)
    assert np.max(np.abs(sol - ref.x)) < 1e-6, (leaf, fan)
    lb, ub = _bounds_arrays(len(x), len(c))
    ok, worst = T.kkt_certificate(c, Am, b, lb, ub, sol, row_dual)
    assert ok, (leaf, fan, worst)
    assert st["leaves"] in (2 * fan, 2 * fan * fan, 2 * fan ** 3) and len(st["pivots_per_level"]) >= 3 and "_basis" in st


@needs_binding
def test_one_long_pair_spreads_its_sub_lps_over_helper_processes(tmp_path, monkeypatch):
  """lp_tree.solve_parallel: the leaves and the merges of a level on helper processes (python -m describealign_amd.lp_helper, fed
  through pipes) -- same optimum, same certificate; helpers that fail leave the LP to the same tree in this process; and because the
  helpers are plain subprocesses, a caller's script needs no `if __name__ == "__main__"` guard."""
  x, y = _instance("e3600")
  c, Am, b, bounds = A.build_trend_lp(x, y)
  ref = scipy.optimize.linprog(c, A_eq=Am, b_eq=b, bounds=bounds, method="highs-ds")
  lp = A.solve_trend_lp(x, y, procs=2)
  assert lp["method"] == "tree" and lp["tree"]["helper_processes"] == 2 and "helpers_failed" not in lp["tree"]
  assert np.max(np.abs(lp["solution"] - ref.x)) < 1e-6 and lp["tree"]["certificate"]["relative_gap"] < 1e-6
  # below PARALLEL_MIN_POINTS nothing is spread
  xs, ys = _instance("e1320")
  small = A.solve_trend_lp(xs, ys, procs=2)
  assert small["method"] == "tree" and "helper_processes" not in small["tree"]
  # helpers that die: the serial tree, still the optimum
  monkeypatch.setattr(T.HelperPool, "run", lambda self, tasks: (_ for _ in ()).throw(RuntimeError("lp helper process ended")))
  lp = A.solve_trend_lp(x, y, procs=2)
  assert lp["method"] == "tree" and "lp helper process ended" in lp["tree"]["helpers_failed"]
  assert np.max(np.abs(lp["solution"] - ref.x)) < 1e-6
  monkeypatch.undo()
  # a script without a main guard
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  script = tmp_path / "no_guard.py"
  script.write_text("import sys, numpy as np\nsys.path.insert(0, %r)\nfrom describealign_amd import align as A\n"
                    "z = np.load(%r)\nlp = A.solve_trend_lp(z['fx'], z['fy'], procs=2)\nprint('RESULT', lp['method'], lp['tree'].get('helper_processes'))\n"
                    % (root, os.path.join(LP_DIR, "e3600.npz")))
  import subprocess, sys
  res = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300)
  assert res.returncode == 0 and res.stdout.count("RESULT") == 1 and "RESULT tree 2" in res.stdout, res.stdout[-500:] + res.stderr[-1500:]
  # the policy: nothing for pairs under 80 minutes, never more than 8, an override
  assert A.lp_helper_procs(210 * 1800) == 0 and 2 <= A.lp_helper_procs(210 * 7200) <= 8
  monkeypatch.setenv("DALIGN_LP_PROCS", "0")
  assert A.lp_helper_procs(210 * 28800) == 0
