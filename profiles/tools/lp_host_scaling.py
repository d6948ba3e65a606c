"""Host probe (no GPU work): how does the throughput of the reference's L1 trend LP (scipy.optimize.linprog, HiGHS dual
simplex) scale with the number of concurrent single-threaded solves, as a function of the LP's size?

  python profiles/tools/lp_host_scaling.py [instance.npz] [seconds per configuration]

The instance's fit points are truncated to n, n/2, n/4 ... points; for each size W pinned worker processes (W over the
host's cores, placed as align.cpu_order places the pipeline's LP workers) solve that LP repeatedly until a deadline.
Prints one JSON line per (size, W): solves/s of the host, mean seconds per solve, and the points-per-second figure
(fit points x solves / s) that a windowed decomposition of a long pair's LP would be measured in.
"""
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


OPTIONS = {}          # linprog options of the variant under test (set from the command line before the pool forks)


def worker(args):
  cpu, fx, fy, t_start, deadline = args
  import scipy.optimize
  from describealign_amd import align as A
  try:
    os.sched_setaffinity(0, {cpu})
  except Exception:
    pass
  c, Am, b, bounds = A.build_trend_lp(fx, fy)
  while time.time() < t_start:
    time.sleep(0.01)
  n, busy = 0, 0.0
  while True:
    t0 = time.time()
    if t0 >= deadline and n > 0:
      break
    fit = scipy.optimize.linprog(c, A_eq=Am, b_eq=b, bounds=bounds, method="highs-ds", options=OPTIONS)
    assert fit.success
    busy += time.time() - t0
    n += 1
  return n, busy, time.time()


def main():
  inst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tests", "golden", "lp", "e7200s.npz")
  secs = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
  z = np.load(inst)
  fx, fy = z["fx"], z["fy"]
  from describealign_amd import align as A
  order = A.cpu_order()
  ncpu = len(order)
  plan = [(1, [16, 32, 64]), (2, [16, 32, 64, 128]), (4, [32, 64, 128]), (8, [32, 64, 128]), (16, [64, 128]), (32, [64, 128])]
  if len(sys.argv) > 3:
    # variant mode: `... instance seconds workers name=value ...` -- the full-size LP only, with these linprog options
    # (e.g. simplex_dual_edge_weight_strategy=devex presolve=False); the CPU-seconds per solve are what the quota-bound host stage pays
    plan = [(1, [int(sys.argv[3])])]
    for kv in sys.argv[4:]:
      k, v = kv.split("=", 1)
      OPTIONS[k] = {"True": True, "False": False}.get(v, v)
  for div, workers in plan:
    n = len(fx) // div
    for w in sorted(set(min(w, ncpu) for w in workers)):
      t_start = time.time() + 3.0 + 0.02 * w
      deadline = t_start + secs
      with mp.get_context("fork").Pool(w) as pool:
        res = pool.map(worker, [(order[k % ncpu], fx[:n], fy[:n], t_start, deadline) for k in range(w)], chunksize=1)
      solves = sum(r[0] for r in res)
      busy = sum(r[1] for r in res)
      elapsed = max(r[2] for r in res) - t_start
      print(json.dumps(dict(fit_points=n, columns=12 * n - 9, workers=w, solves=solves, elapsed_s=round(elapsed, 2),
                            solves_per_s=round(solves / elapsed, 3), s_per_solve=round(busy / solves, 4),
                            fit_points_per_s=round(n * solves / elapsed, 1), host_cpus=ncpu, options=OPTIONS)), flush=True)


if __name__ == "__main__":
  main()
