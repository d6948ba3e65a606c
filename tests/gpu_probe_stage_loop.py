"""GPU-box script (not a pytest): da_pair_stage back to back on one resident pair, the chain DPs collected two stages late
(what a batch's GPU-feeding thread does, minus the pipeline around it): wall time per stage against its kernels.
  python tests/gpu_probe_stage_loop.py [cfg1|cfg2|cfg3] [repeats]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from describealign_amd import _native, synth  # noqa: E402


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
  reps = int(sys.argv[2]) if len(sys.argv) > 2 else 16
  wl = bench.WORKLOADS[name]
  prec = _native.PREC_F32 if wl["precision"] == "f32" else _native.PREC_BF16
  ctx = _native.Context(0, prec)
  pair = synth.make_pair(5, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
  ctx.pcm_upload(0, pair.video); ctx.pcm_upload(1, pair.audio)
  tickets, rows = [], []
  for r in range(reps):
    t0 = time.perf_counter()
    polled = [ctx.chain_done(t) for t in tickets]
    t1 = time.perf_counter()
    vf, af, n, ticket = ctx.pair_stage()
    t2 = time.perf_counter()
    st = ctx.stats()
    tickets.append(ticket)
    if len(tickets) > 2:
      ctx.chain_finish(tickets.pop(0))
    t3 = time.perf_counter()
    rows.append(dict(poll=1e3 * (t1 - t0), stage=1e3 * (t2 - t1), finish_old=1e3 * (t3 - t2), total=1e3 * (t3 - t0),
                     kernels=st["gemm_ms"] + st["verify_ms"] + st["prep_ms"] + st["features_ms"], gemm_ms=st["gemm_ms"], polled=sum(polled)))
  for t in tickets:
    ctx.chain_finish(t)
  keep = rows[3:]
  print(json.dumps(dict(workload=name, chain_cus=os.environ.get("DALIGN_CHAIN_CUS", "default"),
                        mean_ms={k: round(sum(r[k] for r in keep) / len(keep), 3) for k in keep[0]})))
  ctx.close()


if __name__ == "__main__":
  main()
