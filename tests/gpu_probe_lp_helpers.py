"""GPU-box probe (not a pytest): what does starting the LP helper processes cost a process that holds a long pair's PCM and a GPU
context, and when should they be started?   python tests/gpu_probe_lp_helpers.py [seconds] [early|late|none]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from describealign_amd import _native, synth, lp_tree, align as A
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 14400.0
mode = sys.argv[2] if len(sys.argv) > 2 else "late"
if mode == "early":                       # before the process has grown
  t0 = time.perf_counter(); lp_tree.available(); lp_tree.helper_pool(8).wait_ready(); print("helpers started early in %.2f s" % (time.perf_counter() - t0), flush=True)
pair = synth.make_pair(13, secs, n_jumps=20, first_gap=300.0, channels=1)
c = _native.Context(0, _native.PREC_BF16)
vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
if mode == "sync":                        # after PCM and context exist, but not beside the GPU stages
  t0 = time.perf_counter(); lp_tree.available(); lp_tree.helper_pool(8).wait_ready(); print("helpers started (big process) in %.2f s" % (time.perf_counter() - t0), flush=True)
if mode == "none":
  os.environ["DALIGN_LP_PROCS"] = "0"
for rep in range(2):
  tm = {}
  t0 = time.perf_counter()
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=c, timings=tm)
  print(mode, "rep", rep, "total %.2f" % (time.perf_counter() - t0), {k: round(v, 2) for k, v in tm.items() if k in ("match_s", "chain_s", "lp_s", "refine_s", "lp_helper_processes")}, flush=True)
