"""GPU-box script (not a pytest): where does the wall time of one pair's matching stage go beyond its kernels?
  python tests/gpu_probe_gaps.py [cfg1|cfg2|cfg3] [repeats]
Sequential loop over one resident pair, no pipeline, no other threads: features x 2, match_begin, match_finish, chain_begin,
chain_finish -- wall time of every call beside the kernel times da_stats reports."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from describealign_amd import _native, synth  # noqa: E402


def main():
  wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg1"]
  reps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
  prec = _native.PREC_F32 if wl["precision"] == "f32" else _native.PREC_BF16
  ctx = _native.Context(0, prec)
  pair = synth.make_pair(5, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
  ctx.pcm_upload(0, pair.video); ctx.pcm_upload(1, pair.audio)
  rows = []
  ticket = None
  for r in range(reps):
    t = [time.perf_counter()]
    vf = ctx.features_resident(0); af = ctx.features_resident(1); t.append(time.perf_counter())
    ctx.match_begin(vf, af); t.append(time.perf_counter())
    if ticket is not None:
      ctx.chain_finish(ticket)
    t.append(time.perf_counter())
    ctx.match_finish(); t.append(time.perf_counter())
    st = ctx.stats()
    ticket = ctx.chain_begin(); t.append(time.perf_counter())
    d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
    rows.append(dict(features=d[0], match_begin=d[1], chain_finish_prev=d[2], match_finish=d[3], chain_begin=d[4], total=1e3 * (t[-1] - t[0]),
                     gemm_ms=st["gemm_ms"], verify_ms=st["verify_ms"], verify_kernel_ms=st["verify_kernel_ms"], prep_ms=st["prep_ms"], feat_ms=st["features_ms"],
                     survivors=st["survivors"], matches=st["matches"]))
  ctx.chain_finish(ticket)
  keep = rows[2:]
  mean = {k: round(sum(r[k] for r in keep) / len(keep), 3) for k in keep[0]}
  mean["kernels_ms"] = round(mean["gemm_ms"] + mean["verify_ms"] + mean["prep_ms"] + 2 * mean["feat_ms"], 3)
  mean["gap_ms"] = round(mean["total"] - mean["kernels_ms"], 3)
  print(json.dumps(dict(workload=sys.argv[1] if len(sys.argv) > 1 else "cfg1", chain_cus=os.environ.get("DALIGN_CHAIN_CUS", ""), lib=os.path.basename(os.environ.get("DALIGN_LIB", "libdalign.so")), mean_ms=mean)))
  ctx.close()


if __name__ == "__main__":
  main()
