"""Scale check on the GPU box (not a pytest): BASELINE config 5 -- ONE long pair, its matching stage tiled
over `world` ranks (align.align_tiled).  With a single GPU the ranks are processes that share it and
exchange through gloo; on an 8-GPU node the same code runs one rank per GPU over RCCL.

  python tests/gpu_tiled_long_pair.py [seconds] [world]        # default 28800 (8 h), 8 ranks

Rank 0 prints one JSON line: sizes, stage times, recovered offsets vs the injected ones.
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import json, os, sys, time
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root)
secs = float(sys.argv[2])
import torch  # noqa: F401  (before libdalign.so)
from describealign_amd import _native, distrib, synth
from describealign_amd import align as A
g = distrib.Group(os.environ.get("DALIGN_DIST_BACKEND", "gloo"))
dev = g.local_rank if os.environ.get("DALIGN_DIST_BACKEND") == "nccl" else 0
ctx = _native.Context(dev, _native.PREC_BF16)
t0 = time.perf_counter()
pair = synth.make_pair(13, secs, n_jumps=int(round(secs / 720.0)), first_gap=300.0, channels=1)
t_gen = time.perf_counter() - t0
t0 = time.perf_counter()
vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
tm = {}
class FileLock:                       # ranks sharing ONE device take turns in the matching stage (scratch: tens of GB each)
  def __init__(self, path): self.path = path
  def __enter__(self):
    import fcntl
    self.f = open(self.path, "w"); fcntl.flock(self.f, fcntl.LOCK_EX)
  def __exit__(self, *a):
    import fcntl
    fcntl.flock(self.f, fcntl.LOCK_UN); self.f.close()
lock = FileLock(os.path.join(os.environ.get("TMPDIR", "/tmp"), "da_tiled.lock")) if g.backend == "gloo" and g.world > 1 else None
x, y, sim, path, med = A.align_tiled(vf, af, vf[0], af[0], g, ctx=ctx, timings=tm, match_lock=lock)
el = time.perf_counter() - t0
if g.rank == 0:
  offs = x - y
  truth = [pair.true_offset_at(float(t)) for t in y[::2] + 0.5]
  err = max(abs(o - t) for o, t in zip(offs[::2], truth))
  print(json.dumps(dict(seconds=secs, world=g.world, backend=g.backend, generate_s=round(t_gen, 1), align_s=round(el, 1),
                        realtime_factor=round(secs / el, 1), nodes=len(x), segments_expected=len(pair.jump_lengths),
                        similarity=round(float(sim), 2), matches=int(tm["n_matches"]), rows_of_rank0=list(tm["rows"]),
                        match_s=round(tm["match_s"], 2), gather_s=round(tm["gather_s"], 2), chain_s=round(tm.get("chain_s", 0), 2),
                        chain_ms_device=round(tm["device"].get("chain_ms", 0), 1), lp_s=round(tm.get("lp_s", 0), 2),
                        fit_points=int(tm.get("n_fit_points", 0)), refine_s=round(tm.get("refine_s", 0), 2),
                        max_offset_err_vs_injected_ms=round(1e3 * err, 3))), flush=True)
g.close(); ctx.close()
"""


def main():
  secs = float(sys.argv[1]) if len(sys.argv) > 1 else 28800.0
  world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
  script = os.path.join(os.environ.get("TMPDIR", "/tmp"), "da_tiled_worker.py")
  open(script, "w").write(WORKER)
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29577", WORLD_SIZE=str(world))
  t0 = time.perf_counter()
  procs = [subprocess.Popen([sys.executable, script, ROOT, str(secs)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                            stdout=subprocess.PIPE if r else None, stderr=subprocess.STDOUT if r else None, text=True)
           for r in range(world)]
  rc = [p.wait() for p in procs]
  for r, p in enumerate(procs):
    if r and rc[r] != 0:
      print("rank", r, "failed:", p.stdout.read()[-2000:], file=sys.stderr)
  print("all ranks done in %.1f s, exit codes %s" % (time.perf_counter() - t0, rc), file=sys.stderr)
  sys.exit(max(rc))


if __name__ == "__main__":
  main()
