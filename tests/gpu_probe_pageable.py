"""GPU-box script (not a pytest): does a copy from PAGEABLE host memory on another context's stream delay this context's
kernels while a CU-masked (= blocking) chain DP stream is busy?  Thread A: da_pair_stage back to back (as tests/gpu_probe_stage_loop.py);
thread B: its own context, uploads 7 MB from pageable / page-locked memory in a loop.
  python tests/gpu_probe_pageable.py [pageable|pinned|none]"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from describealign_amd import _native, synth  # noqa: E402


def main():
  kind = sys.argv[1] if len(sys.argv) > 1 else "pageable"
  wl = bench.WORKLOADS["cfg1"]
  ctx = _native.Context(0, _native.PREC_F32)
  pair = synth.make_pair(5, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
  ctx.pcm_upload(0, pair.video); ctx.pcm_upload(1, pair.audio)
  stop = threading.Event()
  count = [0]

  def other():
    c2 = _native.Context(0, _native.PREC_F32)
    n = 3_500_000
    buf = (_native.pinned_empty((1, n), np.int16) if kind == "pinned" else np.zeros((1, n), dtype=np.int16))
    while not stop.is_set():
      if kind == "pinned":
        c2.pcm_upload_async(0, buf); c2.features_resident(0, download=False)
      else:
        c2.pcm_upload(0, buf)                 # hipMemcpyAsync from pageable memory + stream synchronise
      count[0] += 1
      time.sleep(0.02)
    c2.close()

  th = None
  if kind != "none":
    th = threading.Thread(target=other); th.start()
  tickets, rows = [], []
  for r in range(24):
    t1 = time.perf_counter()
    vf, af, n, ticket = ctx.pair_stage()
    t2 = time.perf_counter()
    tickets.append(ticket)
    if len(tickets) > 2:
      ctx.chain_finish(tickets.pop(0))
    rows.append(1e3 * (t2 - t1))
  stop.set()
  if th:
    th.join()
  for t in tickets:
    ctx.chain_finish(t)
  keep = sorted(rows[3:])
  print(json.dumps(dict(other_thread=kind, uploads=count[0], stage_ms=dict(min=round(keep[0], 2), median=round(keep[len(keep) // 2], 2), max=round(keep[-1], 2)),
                        all=[round(x, 1) for x in rows[3:]])))
  ctx.close()


if __name__ == "__main__":
  main()
