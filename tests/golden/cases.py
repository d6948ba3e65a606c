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
  "rateneg600": dict(seed=44, video_seconds=600.0, n_jumps=4, first_gap=30.0, rate_change=-0.015),       # the AD copy runs 1.5 % slow
  "j600s":  dict(seed=45, video_seconds=600.0, n_jumps=12, first_gap=8.0, channels=2),                 # a jump every ~45 s, stereo, short intro
  "e1320":  dict(seed=5, video_seconds=1320.0, n_jumps=10, first_gap=200.0),
  # BASELINE configs at their stated sizes, recorded from the reference itself (436 s and ~40 min of
  # reference time, > 10 GB of Python objects for the 2 h pair): the fixtures hold nodes / similarity / slope only
  "e1800":  dict(seed=0, video_seconds=1800.0, n_jumps=10, first_gap=120.0),                # = seed 0 of configs[3] (the batch of 32 half-hour pairs)
  "rate1800": dict(seed=41, video_seconds=1800.0, n_jumps=10, first_gap=120.0, rate_change=0.003),   # configs[3] size with a 0.3 % rate difference between the files
  "j1800":  dict(seed=43, video_seconds=1800.0, n_jumps=25, first_gap=30.0, channels=2),                # many short segments (a jump every ~70 s), stereo
  "e3600":  dict(seed=6, video_seconds=3600.0, n_jumps=10, first_gap=200.0),
  "e7200s": dict(seed=5, video_seconds=7200.0, n_jumps=10, first_gap=200.0, channels=2),   # = bench.py's configs[2] pair (rank 0)
}


def align_case(name: str) -> synth.SynthPair:
  if name == "mismatch":
    return synth.unrelated_pair(31, 120.0, 130.0)
  return synth.make_pair(**ALIGN_CASES[name])


# --------------------------------------------------------------------------- --stretch_audio cases
# name: (channels, seconds of video, seconds of audio, seeds, node times).  Node times are
# (audio_desc_times, video_times) as align() would return them; the slopes are chosen so that
# every branch of the reference's replace_aligned_segments (describealign.py:387-416) is taken:
#   |1-slope| <= .005            -> quadratic resampling
#   offset >= 10000 samples      -> stretch with the 10 base lags
#   1000 < offset < 10000        -> 18 lags
#   offset <= 1000               -> every lag 30..511
#   dy < 2 s or |1-slope| > .1   -> left untouched
STRETCH_CASES = {
  "mix_stereo": dict(channels=2, video_seconds=18.5, audio_seconds=19.0, seed=41,
                     video_times=[0.0, 3.0, 7.0, 8.5, 17.0, 18.4],
                     audio_times=[0.3, 3.303, 7.383, 8.858, 17.103, 18.553]),
  "fine_mono":  dict(channels=1, video_seconds=16.0, audio_seconds=16.5, seed=43,
                     video_times=[0.25, 3.25, 6.25, 11.25, 13.75, 15.9],
                     audio_times=[0.10, 3.118, 6.100, 11.100, 14.100, 16.2491]),
  "edge_mono":  dict(channels=1, video_seconds=9.0, audio_seconds=8.905, seed=45,      # AD 220 samples shorter than its last
                     video_times=[0.0, 4.0, 8.9],                                       # node says: resampling reads past the
                     audio_times=[0.0, 4.004, 8.91]),                                   # end of the data (zeros)
}


def stretch_case(name: str):
  """-> (video int16 (C,N), audio int16 (C,M), audio_times, video_times)."""
  c = STRETCH_CASES[name]
  sr = synth.SAMPLE_RATE
  nv, na = int(c["video_seconds"] * sr), int(c["audio_seconds"] * sr)
  ch = c["channels"]
  vid = [synth.programme(c["seed"], nv)]
  aud = [(synth.programme(c["seed"] + 1, na).astype(np.int64) * 5) // 8]      # quieter AD: loudness matching scales the video
  if ch == 2:
    vid.append((7 * vid[0].astype(np.int64)) // 8 + (synth._noise_i16(c["seed"], 30, 0, nv).astype(np.int64) >> 4))
    aud.append((3 * synth.programme(c["seed"] + 2, na).astype(np.int64)) // 2)    # louder AD channel: the AD is scaled
  vid = np.clip(np.stack(vid), -32767, 32767).astype(np.int16)
  aud = np.clip(np.stack(aud), -32767, 32767).astype(np.int16)
  return vid, aud, np.array(c["audio_times"], dtype=np.float64), np.array(c["video_times"], dtype=np.float64)


def stretch_case_f16(name: str):
  """float16 inputs for replace_aligned_segments as combine() hands them over: PCM held as
  float16 (describealign.py:156) after a per-channel loudness gain (:1144-1148; fixed gains here
  so the fixture does not depend on a reduction order)."""
  vid, aud, x, y = stretch_case(name)
  v, a = vid.astype(np.float16), aud.astype(np.float16)
  for c in range(v.shape[0]):
    if c == 0:
      v[c] /= np.float64(1.1861478)
    else:
      a[c] *= np.float64(0.5779034)
  return v, a, x, y


# Whole --stretch_audio block of combine() (describealign.py:1096-1159): stereo pairs that align.
COMBINE_STRETCH_CASES = {
  "cs60": dict(seed=60, video_seconds=60.0, jumps=([0.0, 25.0], [4.0, 1.5]), channels=2),
}


def combine_stretch_case(name: str) -> synth.SynthPair:
  return synth.make_pair(**COMBINE_STRETCH_CASES[name])
