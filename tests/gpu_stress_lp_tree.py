"""GPU-box script (not a pytest): the host LP through the tree of warm starts (describealign_amd/lp_tree.py) against the reference's
own scipy.optimize.linprog call, on the LPs of random synthetic pairs as the device stages deliver them -- lengths 5 min .. 80 min,
mono / stereo, 0-14 offset jumps, with and without a rate difference between the files -- and what the difference does downstream:
both LP results go through clustering, the banded extension and the second DP, and the nodes (to 1e-9 s), the similarity and the
pass-2 path (same rows, audio frames and clusters; video positions to 1e-7 s) must agree.

  python tests/gpu_stress_lp_tree.py [pairs] [seed]

Prints one JSON line per pair and a summary line."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
  n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
  rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
  import torch  # noqa: F401
  from describealign_amd import _native, synth
  from describealign_amd import align as A
  ctx = _native.Context(0, _native.PREC_BF16)
  worst_diff, n_tree, n_nodes_equal, n_path_equal, t_ref, t_tree = 0.0, 0, 0, 0, 0.0, 0.0
  for k in range(n_pairs):
    sec = float(10 ** rng.uniform(2.48, 3.68))
    rate = float(rng.choice([0.0, 0.0, 0.0, 0.001, -0.0007, 0.02, -0.04]))
    pair = synth.make_pair(int(rng.integers(1 << 30)), sec, n_jumps=int(rng.integers(0, 15)), first_gap=float(rng.uniform(5, max(6.0, sec / 6))),
                           channels=int(rng.integers(1, 3)), rate_change=rate)
    vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
    n_ve, n_ae = len(vf[0]), len(af[0])
    tm = {}
    try:
      fx, fy, a_s, v_s = A._stage_match(ctx, vf, af, n_ve, n_ae, _native.MATCH_HASHED, tm)
    except RuntimeError as e:
      print(json.dumps(dict(pair=k, seconds=round(sec, 1), rate=rate, skipped=str(e))), flush=True)
      continue
    t0 = time.perf_counter(); ref = A.solve_trend_lp(fx, fy, tree=False); t1 = time.perf_counter()
    got = A.solve_trend_lp(fx, fy, tree=True); t2 = time.perf_counter()
    diff = float(np.max(np.abs(got["solution"] - ref["solution"])))
    outs = []
    for lp in (ref, got):
      t = {}
      try:
        outs.append(A._stage_refine(ctx, lp, a_s.copy(), v_s.copy(), n_ve, n_ae, t))
      except RuntimeError as e:
        outs.append(str(e))
    # the LP solutions agree to the solver's tolerance, not bit for bit, and a path row's video position is computed from the
    # cluster's fitted line: compared to 1e-9 s (nodes) / 1e-7 s (video position of a path row); rows, audio frames and
    # cluster numbers exactly
    node_err, path_err, same_rows = float("nan"), float("nan"), False
    if isinstance(outs[0], str) or isinstance(outs[1], str):
      same_nodes = same_path = (outs[0] == outs[1]) if (isinstance(outs[0], str) and isinstance(outs[1], str)) else False
    else:
      (x0, y0, s0, p0, _), (x1, y1, s1, p1, _) = outs
      same_len = len(x0) == len(x1)
      node_err = float(max(np.max(np.abs(x0 - x1)), np.max(np.abs(y0 - y1)))) if same_len else float("inf")
      same_nodes = same_len and node_err <= 1e-9 and abs(s0 - s1) <= 1e-9
      same_rows = p0.shape == p1.shape and np.array_equal(p0[:, 1], p1[:, 1]) and np.array_equal(p0[:, 2], p1[:, 2])
      path_err = float(np.max(np.abs(p0[:, 0] - p1[:, 0]))) if same_rows else float("inf")
      same_path = same_rows and path_err <= 1e-7
    worst_diff = max(worst_diff, diff)
    n_tree += got["method"] == "tree"; n_nodes_equal += bool(same_nodes); n_path_equal += bool(same_path)
    t_ref += t1 - t0; t_tree += t2 - t1
    print(json.dumps(dict(pair=k, seconds=round(sec, 1), rate=rate, channels=int(pair.video.shape[0]), fit_points=len(fx), method=got["method"],
                          reference_s=round(t1 - t0, 3), tree_s=round(t2 - t1, 3), max_diff=diff, refactored=bool(got["tree"].get("refactored", False)),
                          median_slope=ref["median_slope"], nodes_equal=bool(same_nodes), path_equal=bool(same_path),
                          max_node_difference_s=node_err, max_path_video_difference_s=path_err)), flush=True)
  print(json.dumps(dict(summary=True, pairs=n_pairs, solved_by_tree=int(n_tree), nodes_equal=int(n_nodes_equal), path_equal=int(n_path_equal),
                        worst_solution_difference=worst_diff, reference_seconds=round(t_ref, 1), tree_seconds=round(t_tree, 1))), flush=True)
  ctx.close()


if __name__ == "__main__":
  main()
