"""GPU-box script (not a pytest): does the similarity GEMM run slower when the device idled before it?
  python tests/gpu_probe_idle.py [cfg2|cfg1]
The same resident pair matched back to back, then with pauses of 0.1 / 0.3 / 1.0 s between the launches."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from describealign_amd import _native, synth  # noqa: E402


def main():
  name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
  wl = bench.WORKLOADS[name]
  prec = _native.PREC_F32 if wl["precision"] == "f32" else _native.PREC_BF16
  ctx = _native.Context(0, prec)
  pair = synth.make_pair(5, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
  ctx.pcm_upload(0, pair.video); ctx.pcm_upload(1, pair.audio)
  vf = ctx.features_resident(0); af = ctx.features_resident(1)
  out = {}
  for pause in (0.0, 0.1, 0.3, 1.0, 0.0):
    ms = []
    for r in range(8):
      time.sleep(pause)
      ctx.match_begin(vf, af); ctx.match_finish()
      st = ctx.stats()
      ms.append(st["gemm_ms"])
    out[f"pause_{pause}_s" + ("_again" if f"pause_{pause}_s" in out else "")] = dict(gemm_ms=[round(x, 2) for x in ms[2:]], mean=round(sum(ms[2:]) / len(ms[2:]), 2),
                                                                                      verify_ms=round(st["verify_ms"], 2))
  print(json.dumps(dict(workload=name, result=out)))
  ctx.close()


if __name__ == "__main__":
  main()
