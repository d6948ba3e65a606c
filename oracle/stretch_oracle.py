"""CPU oracle for the audio replacement path (--stretch_audio) -- TEST INFRASTRUCTURE, NOT PRODUCT.

A numpy restatement of `replace_aligned_segments` in julbean/describealign v2.0.8
(`describealign.py:230-416`) plus the loudness matching / peak normalisation that brackets it
in `combine()` (`:1135-1153`).  Only tests/ may import this module; the product
(describealign_amd/) never does.

The reference writes this as nested closures and a recursive generator; here it is a flat set
of array functions, one per step, so each HIP kernel has a stage to be compared with:

  resample_quadratic   :233-244   chunked quadratic-spline resampling (scipy interp1d)
  correlation_chunks   :253-270   which samples each block of windows is computed from
  jump_correlations    :271-296   windowed Pearson correlation at fixed lags (one chunk)
  jump_table           :319-326   per 512-sample window: best position + loss for every lag
  drift_viterbi        :311-345   Viterbi over (window, drift) choosing where to jump
  drift_backtrack      :346-362   the jump schedule [(input index, signed distance)]
  splice               :363-385   copy segments with 512-sample Hann cross-fades
  stretch              :298-385   the four steps above for one segment
  replace_aligned_segments :387-416  per segment: skip / resample / stretch

Pinning: tests/test_oracle_golden.py checks every function here against fixtures that
tests/golden/make_golden.py recorded by running the reference itself in the build container
(jump schedules and full output waveforms for the cases in tests/golden/cases.py STRETCH_CASES).
match_loudness / normalise_peak restate lines that are written inline in the reference's combine();
they are pinned by running that combine(stretch_audio=True) itself with only its file I/O replaced
(COMBINE_STRETCH_CASES: sha1 of the scaled tracks, of the array handed to the writer and of its
int16 frames).

Third-party arithmetic not under /root/reference: `scipy.interpolate.interp1d(kind='quadratic')`
(scipy 1.15.3 here; `scipy~=1.10` in the reference's requirements.txt:5) -- called here exactly as
the reference calls it (`:239-242`).  It is `make_interp_spline(x, y, k=2)`: knots at the data
mid-points with the first and last mid-point removed and the end points tripled, coefficients
from the banded collocation system, values by de Boor's recurrence.
"""
from __future__ import annotations

import numpy as np
import scipy.interpolate
import scipy.signal

SAMPLE_RATE = 44100                 # AUDIO_SAMPLE_RATE                     (describealign.py:31)
MAX_RATE_DIFF = 0.1                 # MAX_RATE_RATIO_DIFF_ALIGN             (:33)
MIN_REPLACE_SECONDS = 2             # MIN_DURATION_TO_REPLACE_SECONDS       (:34)
JND_RATE = 0.005                    # JUST_NOTICEABLE_DIFF_IN_FREQ_RATIO    (:35)
MIN_OFFSET = 30                     # MIN_STRETCH_OFFSET                    (:36)
WINDOW = 512                        # window_size                           (:251, :298)
MAX_DRIFT = 3 * WINDOW              # max_drift                             (:298)
N_DRIFT = 2 * MAX_DRIFT + 1         # drift_window_size                     (:299)
CACHED = 50                         # max_cached_chunks                     (:254)
RESAMPLE_CHUNK = 10 ** 5            # chunk_size                            (:234)
BASE_JUMPS = (506, 451, 284, 410, 480, 379, 308, 430, 265, 494)          # (:303)


# --------------------------------------------------------------------------- resampling

def resample_quadratic(audio: np.ndarray, points: np.ndarray) -> np.ndarray:
  """describealign.py:233-244.  `audio` float16 (C, N); `points` ascending float64 sample
  positions.  Every block of 1e5 points gets its own spline through the samples from two
  before its first point to two after its last; positions past the data give 0."""
  n_audio = audio.shape[1]
  pieces = []
  for lo in range(0, len(points), RESAMPLE_CHUNK):
    blk = points[lo:lo + RESAMPLE_CHUNK]
    b0 = max(int(blk[0] - 2), 0)
    b1 = min(int(blk[-1] + 2), n_audio)
    f = scipy.interpolate.interp1d(np.arange(b0, b1), audio[:, b0:b1], copy=False, bounds_error=False,
                                   fill_value=0, kind="quadratic", assume_sorted=True)
    pieces.append(f(blk).astype(np.float16))
  return np.hstack(pieces)


# --------------------------------------------------------------------------- lag correlations

def jump_list(total_offset: int):
  """describealign.py:303-308: which lags the stretcher may jump by."""
  mag = abs(int(total_offset))
  if mag >= 10000:
    return list(BASE_JUMPS)
  if mag > 1000:
    return list(BASE_JUMPS) + [MIN_OFFSET + (1 << b) - 1 for b in range(8)]
  return list(range(MIN_OFFSET, WINDOW))


def correlation_chunks(n: int):
  """describealign.py:253-270 unrolled.  Returns [(sample_begin, sample_end, w_lo, w_hi)]: the
  windows with local index w_lo..w_hi-1 of a chunk are computed from samples
  [sample_begin, sample_end) only (that limits which positions are valid and sets the chunk's
  epsilon).  Chunks overlap by two windows; every chunk but the first drops its first window."""
  limit = (CACHED + 2) * 1.1 * WINDOW
  if n <= limit:
    return [(0, n, 0, n // WINDOW)]
  out = []
  begin, lo = 0, 0
  while True:
    rest = n - begin
    if rest <= limit:
      out.append((begin, n, lo, rest // WINDOW))
      return out
    out.append((begin, begin + (CACHED + 1) * WINDOW, lo, CACHED))
    begin += (CACHED - 1) * WINDOW
    lo = 1


def _windowed(values_f32: np.ndarray) -> np.ndarray:
  """Sliding 512-sums the way the reference forms them (:275-277, :283-284): float64 running
  sum, then the difference of the running sum 512 apart."""
  cs = np.cumsum(values_f32, dtype=np.float64)
  head = cs[WINDOW - 1:].copy()
  head[1:] -= cs[:len(cs) - WINDOW]
  return head


def jump_correlations(chunk: np.ndarray, backwards: bool, jumps) -> np.ndarray:
  """describealign.py:271-293 for one chunk (C, L) float16.  Returns (L-511, J): row p, column j
  = Pearson correlation of the window starting at p with the window `jumps[j]` later
  (or earlier when `backwards`); -inf where the second window leaves the chunk."""
  L = chunk.shape[1]
  if L < 3 * WINDOW - 1:
    raise RuntimeError("Invalid state in Pearson generator.")
  P = L - WINDOW + 1
  x32 = chunk.astype(np.float32)
  energy = _windowed(np.sum(x32 ** 2, axis=0))
  eps = 1e-4 * max(1, np.max(energy))
  rms = np.sqrt(energy + eps)
  out = np.full((P, len(jumps)), -np.inf)
  for j, lag in enumerate(jumps):
    lag = int(lag)
    dots = _windowed(np.sum(x32[:, lag:] * chunk[:, :L - lag], axis=0)) + eps      # length P - lag
    if backwards:
      out[lag:, j] = dots / rms[:P - lag]
    else:
      out[:P - lag, j] = dots / rms[lag:]
  return out / rms[:, None]


def jump_table(seg: np.ndarray, backwards: bool, jumps):
  """describealign.py:319-326 over the whole segment: for every 512-sample window and every lag
  the position inside the window with the highest correlation, and 1 - that correlation."""
  n = seg.shape[1]
  nw = n // WINDOW
  J = len(jumps)
  where = np.zeros((nw, J), dtype=np.int16)
  loss = np.zeros((nw, J))
  cols = np.arange(J)
  for begin, end, w_lo, w_hi in correlation_chunks(n):
    corr = jump_correlations(seg[:, begin:end], backwards, jumps)
    w0 = begin // WINDOW
    for w in range(w_lo, w_hi):
      if w0 + w >= nw:
        break
      rows = corr[w * WINDOW:(w + 1) * WINDOW]
      at = np.argmax(rows, axis=0)
      where[w0 + w] = at
      loss[w0 + w] = 1 - rows[at, cols]
  return where, loss


# --------------------------------------------------------------------------- drift Viterbi

def _offset_at(total: int, nw: int, w: int) -> int:
  return (total * min(nw - 1, max(0, w))) // (nw - 1)                  # (:310-311)


def _offset_step(total: int, nw: int, w: int) -> int:
  return abs(_offset_at(total, nw, w) - _offset_at(total, nw, w - 1))   # (:316-317)


def drift_viterbi(loss: np.ndarray, jumps, total: int) -> np.ndarray:
  """describealign.py:318-345.  State = (window, drift in [-1536, 1536]); a step either keeps
  the drift schedule (cost 0) or jumps by one lag from two windows back (cost = that window's
  loss).  Returns the int16 back-pointers (0 = no jump, k+1 = lag k)."""
  nw, J = loss.shape
  back = np.zeros((nw, N_DRIFT), dtype=np.int16)
  hist = np.full((3, N_DRIFT), np.inf)
  hist[1:, MAX_DRIFT] = 0
  prev_step = 0
  cols = np.arange(N_DRIFT)
  for w in range(nw):
    step = _offset_step(total, nw, w)
    two = step + prev_step
    opts = np.full((J + 1, N_DRIFT), np.inf)
    opts[0, :N_DRIFT - step] = hist[(w - 1) % 3, step:]
    older = hist[(w - 2) % 3]
    for k, lag in enumerate(jumps):
      lag = int(lag)
      cut = two - lag
      opts[k + 1, lag:N_DRIFT - max(0, cut)] = older[two:N_DRIFT + min(0, cut)] + loss[w, k]
    pick = np.argmin(opts, axis=0)
    back[w] = pick
    hist[w % 3] = opts[pick, cols]
    prev_step = step
  return back


def drift_backtrack(back: np.ndarray, where: np.ndarray, jumps, total: int) -> np.ndarray:
  """describealign.py:346-368.  Returns (K, 2) int64: input index of each jump and its signed
  distance (negative = repeat samples, when the output is longer than the input)."""
  nw = back.shape[0]
  drift = MAX_DRIFT
  found = []
  skip = False
  for w in range(nw - 1, -1, -1):
    drift += _offset_step(total, nw, w + 1)
    if skip:
      skip = False
      continue
    k = int(back[w, drift]) - 1
    if k < 0:
      continue
    lag = int(jumps[k])
    found.append((w * WINDOW + int(where[w, k]), lag))
    drift -= lag
    skip = True
  sched = np.array(found[::-1], dtype=np.int64).reshape(-1, 2)
  if total > 0:
    sched[:, 1] *= -1
  return sched


def splice(seg: np.ndarray, out: np.ndarray, sched: np.ndarray) -> None:
  """describealign.py:369-385.  Copies the runs between jumps into `out` (float16, in place),
  cross-fading 512 samples after every jump with the two halves of hann(1025)."""
  n = seg.shape[1]
  starts = np.concatenate(([0], sched[:, 0] + sched[:, 1]))
  ends = np.concatenate((sched[:, 0], [n]))
  o_end = np.cumsum(ends - starts)
  o_start = np.concatenate(([0], o_end[:-1]))
  bump = scipy.signal.windows.hann(2 * WINDOW + 1)
  rise, fall = bump[:WINDOW], bump[WINDOW:-1]
  out[:, :WINDOW] = seg[:, :WINDOW]
  for a, b, oa, ob in zip(starts, ends, o_start, o_end):
    out[:, oa:oa + WINDOW] *= fall
    out[:, oa:oa + WINDOW] += seg[:, a:a + WINDOW] * rise
    out[:, oa + WINDOW:ob + WINDOW] = seg[:, a + WINDOW:b + WINDOW]


def stretch_plan(seg: np.ndarray, n_out: int):
  """Everything of `stretch` (:298-368) up to the jump schedule; returns (jumps, where, loss,
  back, sched) so each stage can be compared."""
  total = n_out - seg.shape[1]
  jumps = jump_list(total)
  where, loss = jump_table(seg, total > 0, jumps)
  back = drift_viterbi(loss, jumps, total)
  sched = drift_backtrack(back, where, jumps, total)
  return jumps, where, loss, back, sched


def stretch(seg: np.ndarray, out: np.ndarray):
  """describealign.py:298-385: pitch-preserving change of length by jump-and-cross-fade."""
  sched = stretch_plan(seg, out.shape[1])[4]
  splice(seg, out, sched)
  return sched


# --------------------------------------------------------------------------- driver

def segment_plan(audio_times, video_times, no_pitch_correction: bool):
  """describealign.py:387-411.  Per node interval: ('skip' | 'resample' | 'stretch',
  x0, x1, y0, y1) in samples."""
  xs = (np.asarray(audio_times) * SAMPLE_RATE).astype(int)
  ys = (np.asarray(video_times) * SAMPLE_RATE).astype(int)
  dx, dy = np.diff(xs), np.diff(ys)
  with np.errstate(divide="ignore", invalid="ignore"):
    slope = dx / dy
  plan = []
  for k in range(len(xs) - 1):
    if dy[k] < MIN_REPLACE_SECONDS * SAMPLE_RATE or np.abs(1 - slope[k]) > MAX_RATE_DIFF:
      kind = "skip"
    elif no_pitch_correction or np.abs(1 - slope[k]) <= JND_RATE or abs(dy[k] - dx[k]) < MIN_OFFSET:
      kind = "resample"
    else:
      kind = "stretch"
    plan.append((kind, int(xs[k]), int(xs[k + 1]), int(ys[k]), int(ys[k + 1])))
  return plan


def replace_aligned_segments(video: np.ndarray, audio: np.ndarray, audio_times, video_times,
                             no_pitch_correction: bool = False):
  """describealign.py:230-416.  `video`, `audio` float16 (C, N); `video` is modified in place.
  Returns the jump schedules of the stretched segments (for the tests)."""
  schedules = []
  for kind, x0, x1, y0, y1 in segment_plan(audio_times, video_times, no_pitch_correction):
    if kind == "skip":
      continue
    dest = video[:, y0:y1]
    if kind == "resample":
      dest[:] = resample_quadratic(audio, np.linspace(x0, x1, num=y1 - y0, endpoint=False))
    else:
      schedules.append(stretch(audio[:, x0:x1], dest))
  return schedules


def match_loudness(video: np.ndarray, audio: np.ndarray):
  """describealign.py:1135-1148: per channel, scale the louder of the two tracks down to the
  RMS deviation of the other (float16 arrays, in place).  Returns the scale factors."""
  def spread(a):
    mean = np.mean(a, dtype=np.float64)
    return np.sqrt(np.einsum("ij,ij->i", a, a, dtype=np.float64) / np.prod(a.shape) - mean ** 2)
  factor = spread(video) / spread(audio)
  for c, f in enumerate(factor):
    if f > 1:
      video[c] /= f
    else:
      audio[c] *= f
  return factor


def normalise_peak(video: np.ndarray) -> None:
  """describealign.py:1153: rescale to +/- 32766 (float16, in place)."""
  video *= (2 ** 15 - 2.) / np.max(np.abs(video))
