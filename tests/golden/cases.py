"""Deterministic input builders shared by make_golden.py (which runs the reference in the
build container) and the tests (which run anywhere).  Inputs are regenerated from seeds and
checked against the sha1 stored in each fixture, so fixtures hold expected outputs only."""
from __future__ import annotations

import hashlib

import numpy as np

from describealign_amd import synth


def sha1_of(*arrays) -> str:
  h = hashlib.sha1()
  for a in arrays:
    h.update(np.ascontiguousarray(a).tobytes())
  return h.hexdigest()


def feature_clip(name: str) -> np.ndarray:
  """int16 (C, N) clips exercising the feature kernels' edge cases."""
  sr = synth.SAMPLE_RATE
  if name == "mono6":          # N multiple of 210
    return synth.programme(11, 6 * sr).astype(np.int16)[None, :]
  if name == "ragged":         # floor(N/105) odd -> energy row one longer than the others
    return synth.programme(12, 6 * sr + 157).astype(np.int16)[None, :]
  if name == "stereo":
    a = synth.programme(13, 5 * sr + 333)
    b = (9 * a) // 10 + (synth._noise_i16(13, 30, 0, len(a)).astype(np.int64) >> 5)
    return np.stack([a, np.clip(b, -32767, 32767)]).astype(np.int16)
  if name == "loud":           # most samples beyond the exact-integer range of float16
    a = synth.programme(14, 4 * sr + 41).astype(np.int64) * 3
    return np.clip(a, -32768, 32767).astype(np.int16)[None, :]
  if name == "short":          # half a second
    return synth.programme(15, sr // 2).astype(np.int16)[None, :]
  if name == "silence":        # digital silence with one burst
    a = np.zeros(3 * sr + 77, dtype=np.int16)
    a[sr:sr + 9000] = synth.programme(16, 9000).astype(np.int16)
    return a[None, :]
  if name == "stereo_anti":    # channels that partly cancel in the mono mix; odd sample values
    a = synth.programme(17, 3 * sr + 5)
    b = -a + (synth._noise_i16(17, 30, 0, len(a)).astype(np.int64) >> 3) | 1
    return np.stack([a, np.clip(b, -32767, 32767)]).astype(np.int16)
  raise KeyError(name)


FEATURE_CLIPS = ["mono6", "ragged", "stereo", "loud", "short", "silence", "stereo_anti"]


ALIGN_CASES = {
  # name: kwargs for synth.make_pair
  "a40":    dict(seed=21, video_seconds=40.0, jumps=([0.0, 20.0], [5.0, 2.0])),
  "e180":   dict(seed=1, video_seconds=180.0, jumps=([0.0, 90.0], [12.5, 3.0])),
  "e180s":  dict(seed=2, video_seconds=180.0, jumps=([0.0, 90.0], [12.5, 3.0]), channels=2),
  "e600":   dict(seed=3, video_seconds=600.0, n_jumps=5, first_gap=60.0),
  "rate2":  dict(seed=4, video_seconds=300.0, jumps=([0.0], [20.0]), rate_change=0.02),
  "e1320":  dict(seed=5, video_seconds=1320.0, n_jumps=10, first_gap=200.0),
}


def align_case(name: str) -> synth.SynthPair:
  if name == "mismatch":
    return synth.unrelated_pair(31, 120.0, 130.0)
  return synth.make_pair(**ALIGN_CASES[name])
