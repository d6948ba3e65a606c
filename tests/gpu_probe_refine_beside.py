"""GPU-box script (not a pytest): does another context's pass-2 work (da_refine: kernels, device sort, copies, the host second
DP) beside the feeding loop reproduce the late GEMM starts of the batch pipeline (profiles/r05_pipeline_stalls.txt)?
Thread A: da_pair_stage back to back, DPs collected two stages late; threads B1..Bn: own contexts, _stage_refine in a loop.
  python tests/gpu_probe_refine_beside.py [refine_threads=0|1|4] [sleep_ms between refines=30] [what=refine|gpu|python|spin|sync]
what: refine = the whole stage (default); gpu = only da_refine (kernels, sort, copies, host second DP); python = only the stage's
numpy / Python parts; spin = a pure Python loop; sync = an empty kernel-less round trip (features of a 1 s clip) per iteration."""
import contextlib
import io
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from describealign_amd import _native, synth  # noqa: E402
from describealign_amd import align as A  # noqa: E402


def main():
  n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
  pause = float(sys.argv[2]) / 1e3 if len(sys.argv) > 2 else 0.03
  what = sys.argv[3] if len(sys.argv) > 3 else "refine"
  wl = bench.WORKLOADS["cfg1"]
  ctx = _native.Context(0, _native.PREC_F32)
  pair = synth.make_pair(5, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
  ctx.pcm_upload(0, pair.video); ctx.pcm_upload(1, pair.audio)
  vf = ctx.features_resident(0); af = ctx.features_resident(1)
  n_ve, n_ae = len(vf[0]), len(af[0])
  tm = {}
  with contextlib.redirect_stdout(io.StringIO()):
    fx, fy, a_scaled, v_scaled = A._stage_match(ctx, vf, af, n_ve, n_ae, _native.MATCH_HASHED, tm)
    lp = A.solve_trend_lp(fx, fy)
  stop = threading.Event()
  count = [0]

  def other():
    c2 = _native.Context(0, _native.PREC_F32)
    x0, x1, off, slo = A.cluster_lines(lp["smooth_x"], lp["smooth_y"], lp["slopes"])
    path, n_points = c2.refine(a_scaled, v_scaled, x0, x1, off, slo, min_len=A.min_path_length(n_ve, n_ae))
    import numpy as np
    tiny = np.zeros((1, 44100), dtype=np.int16)
    while not stop.is_set():
      if what == "refine":
        A._stage_refine(c2, lp, a_scaled, v_scaled, n_ve, n_ae, {})
      elif what == "gpu":
        c2.refine(a_scaled, v_scaled, x0, x1, off, slo, min_len=A.min_path_length(n_ve, n_ae))
      elif what == "python":
        A.cluster_lines(lp["smooth_x"], lp["smooth_y"], lp["slopes"]); A.nodes_and_similarity(path, x0, x1, off, slo, n_ve, n_ae) if hasattr(A, "nodes_and_similarity") else None
      elif what == "spin":
        t_end = time.perf_counter() + 0.03
        k = 0
        while time.perf_counter() < t_end:
          k += 1
      elif what == "sync":
        c2.pcm_upload(0, tiny); c2.features_resident(0, download=False)
      count[0] += 1
      time.sleep(pause)
    c2.close()

  threads = [threading.Thread(target=other) for _ in range(n_threads)]
  for t in threads:
    t.start()
  tickets, rows = [], []
  for r in range(40):
    t1 = time.perf_counter()
    v2, a2, n, ticket = ctx.pair_stage()
    t2 = time.perf_counter()
    st = ctx.stats()
    tickets.append(ticket)
    if len(tickets) > 2:
      ctx.chain_finish(tickets.pop(0))
    rows.append((1e3 * (t2 - t1), st["gemm_ms"]))
  stop.set()
  for t in threads:
    t.join()
  for t in tickets:
    ctx.chain_finish(t)
  keep = rows[4:]
  walls = sorted(w for w, _ in keep)
  late = sum(1 for w, g in keep if w - g > 6.0)
  print(json.dumps(dict(refine_threads=n_threads, what=what, refines=count[0], stage_ms=dict(p10=round(walls[len(walls) // 10], 2), p50=round(walls[len(walls) // 2], 2),
                                                                                 p90=round(walls[len(walls) * 9 // 10], 2), max=round(walls[-1], 2)),
                        stages=len(keep), late_stages=late, gemm_ms=round(sum(g for _, g in keep) / len(keep), 2))))
  ctx.close()


if __name__ == "__main__":
  main()
