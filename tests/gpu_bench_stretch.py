"""Kernel-level timing of the audio replacement path on the GPU box (not a test).

  python tests/gpu_bench_stretch.py [seconds] [rate_change]

Builds a stereo pair of `seconds` (default 1320) with 10 offset jumps, aligns it through the HIP
path to get the nodes, then times da_stretch_resident (loudness matching + replace + peak
normalisation + int16).  With rate_change (e.g. 0.02) the AD runs at a different speed, so every
interval goes through the pitch-preserving stretcher instead of the resampler."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from describealign_amd import _native, synth, align as A

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 1320.0
rate = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
with_oracle = len(sys.argv) > 3 and sys.argv[3] == "oracle"
pair = synth.make_pair(5, secs, n_jumps=10, first_gap=min(200.0, secs / 6), channels=2, rate_change=rate)
c = _native.Context(0, _native.PREC_F32)
c.pcm_upload(0, pair.video); c.pcm_upload(1, pair.audio)
vf = c.features_resident(0); af = c.features_resident(1)
res = A.align(vf, af, vf[0], af[0], ctx=c)
x, y = np.asarray(res[0]), np.asarray(res[1])
best = None
for rep in range(3):
  t0 = time.perf_counter()
  out, fac = c.stretch_resident(x, y, False)
  wall = time.perf_counter() - t0
  st = c.stats()
  if best is None or wall < best[0]:
    best = (wall, st)
wall, st = best
n_v, n_a = pair.video.shape[1], pair.audio.shape[1]
rb = st["resample_bytes"]
line = dict(seconds=secs, rate_change=rate, nodes=len(x), wall_ms=round(wall * 1e3, 2),
            audio_hours_per_s=round(secs / 3600.0 / wall, 3),
            prepare_ms=round(st["stretch_prepare_ms"], 3), resample_ms=round(st["resample_ms"], 3),
            resample_points=st["resample_points"],
            resample_GBs=round(rb / max(st["resample_ms"], 1e-9) / 1e6, 1) if st["resample_ms"] else None,
            correlate_ms=round(st["correlate_ms"], 3), correlate_windows=st["correlate_windows"],
            viterbi_ms=round(st["viterbi_ms"], 3), splice_ms=round(st["splice_ms"], 3), splice_points=st["splice_points"],
            finish_ms=round(st["stretch_finish_ms"], 3), schedules=[len(s) for s in c.stretch_schedules()],
            factors=[round(float(f), 4) for f in fac])
if with_oracle:      # the CPU restatement on the same nodes: timing beside the GPU path and full-size parity
  from oracle import stretch_oracle as SO
  t0 = time.perf_counter()
  v, a = pair.video.astype(np.float16), pair.audio.astype(np.float16)
  SO.match_loudness(v, a)
  SO.replace_aligned_segments(v, a, x, y, False)
  SO.normalise_peak(v)
  want = v.astype(np.int16).T
  cpu = time.perf_counter() - t0
  line["cpu_oracle_s"] = round(cpu, 2)
  line["cpu_audio_hours_per_s"] = round(secs / 3600.0 / cpu, 5)
  line["samples_differing_from_oracle"] = int((want != out).sum())
  line["samples_total"] = int(want.size)
print(json.dumps(line))
c.close()
