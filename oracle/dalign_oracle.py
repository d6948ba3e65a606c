"""CPU oracle for the alignment hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A numpy restatement of the algorithm in julbean/describealign v2.0.8
(`describealign.py:545-1027`), written stage by stage so that every stage of the HIP
path can be checked against it.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product (describealign_amd/) never does.

Pinning: every stage below is checked in tests/test_oracle_golden.py against fixtures
that tests/golden/make_golden.py recorded by running the reference itself in the build
container (features for 7 clips; hash candidates, verified matches, pass-1 path, LP
input/solution, clusters, pass-2 points and final path for a 40 s pair; end-to-end nodes
for 180 s - 1320 s pairs).  The reference has no tests of its own.

Third-party arithmetic on the path that is not under /root/reference:
`scipy.optimize.linprog` (HiGHS; scipy 1.15.3 here, `scipy~=1.10` pinned by the reference's
requirements.txt:5) -- called here exactly as the reference calls it (`:841-844`); and
`sortedcontainers.SortedList` (`:654`, `:946`) -- restated with `bisect` on plain lists.
"""
from __future__ import annotations

import bisect
from collections import defaultdict

import numpy as np
import scipy.interpolate
import scipy.optimize
import scipy.signal
import scipy.sparse

FRAME_RATE = 210            # feature frames per second            (describealign.py:546,559,577)
WIN = 41                    # correlation / norm window, frames    (:596-597: 2*21-1)
HALF = 21                   # samples_per_node                     (:596)
QUIET = 0.5                 # energy threshold for "not quiet"     (:629, :657)
N_TAPS = 7                  # hash taps per feature                (:610)
TAP_STEP = 6                # (:611)
TAP_START = 2               # bins_start = 21 - 1 - 37//2          (:613)


# --------------------------------------------------------------------------- features

def pcm_to_half(pcm_i16: np.ndarray) -> np.ndarray:
  """describealign.py:156 -- PCM is held as float16 (lossy above |2048|)."""
  return np.asarray(pcm_i16).astype(np.float16)


def _inner_hann(m: int) -> np.ndarray:
  """hann(m+2)[1:-1] as float32, normalised to unit sum (:551-552, :563-564, :569-570)."""
  w = scipy.signal.windows.hann(m + 2)[1:-1].astype(np.float32)
  return w / np.sum(w)


def energy(arr: np.ndarray) -> np.ndarray:
  """describealign.py:545-555.  Mean square per 105-sample block over all channels,
  13-tap Hann smoothing at 420 Hz, log10(1+e)/2, every 2nd value."""
  C, N = arr.shape
  nb = N // 105
  blk = arr[:, :nb * 105].reshape(C, nb, 105).astype(np.float32)
  e = np.einsum("cbk,cbk->b", blk, blk, dtype=np.float32) / np.float32(105 * C)
  sm = np.convolve(e, _inner_hann(13), mode="same")
  return (np.log10(1 + sm) / 2.0)[::2]


def zero_crossings(arr: np.ndarray) -> np.ndarray:
  """describealign.py:557-566.  Sign changes (first sample compared with 'positive') per
  210-sample block summed over channels, doubled for mono, 13-tap Hann smoothing."""
  C, N = arr.shape
  neg = np.signbit(arr)
  prev = np.concatenate([np.zeros((C, 1), dtype=bool), neg[:, :-1]], axis=1)
  flips = neg != prev
  nf = N // 210
  cnt = flips[:, :nf * 210].reshape(C, nf, 210).sum(axis=(0, 2)).astype(np.float32)
  if C == 1:
    cnt *= 2
  return np.convolve(cnt, _inner_hann(13), mode="same")


def _group_fir(x: np.ndarray, d: int, blur: int) -> np.ndarray:
  """describealign.py:568-573 (`downsample_blur`) written out as one FIR.

  With G[g, i] = x[g*d + i] and W = normalised hann(d*blur+2)[1:-1], the reference's sum of
  per-phase 'same' convolutions is  out[m] = sum_{k<blur} sum_{i<d} W[i + d*k] * G[m + c - k, i]
  with c = (blur-1)//2 and G zero outside [0, len(x)//d)."""
  w = _inner_hann(d * blur)
  n = len(x) - (len(x) % d)
  G = x[:n].reshape(-1, d)
  ng = G.shape[0]
  c = (blur - 1) // 2
  out_dtype = np.result_type(G.dtype, np.float32)
  if ng < blur:
    # np.convolve(..., 'same') returns max(len) samples; the reference then sums arrays of length
    # `blur`.  Only reachable for < 0.1 s of audio; mirror numpy exactly via convolve itself.
    return sum(np.convolve(x[:n][i::d], w[i::d], mode="same") for i in range(d))
  out = np.zeros(ng, dtype=out_dtype)
  Gp = np.zeros((ng + blur - 1, d), dtype=out_dtype)
  Gp[blur - 1 - c: blur - 1 - c + ng] = G          # Gp[r] = G[r - (blur-1-c)]
  for k in range(blur):
    # rows m + c - k  ->  Gp index m + c - k + blur - 1 - c = m + blur - 1 - k
    out += Gp[blur - 1 - k: blur - 1 - k + ng] @ w[d * k: d * k + d].astype(out_dtype)
  return out


def freq_bands(arr: np.ndarray):
  """describealign.py:575-593.  Three log band energies from a cascaded decimating
  low-pass bank (decimations 5, 7, 6); the last band is float64 in the reference."""
  mono = np.mean(arr, axis=0) if arr.shape[0] > 1 else arr[0]     # stays float16 (:576)
  mono = mono[:len(mono) - (len(mono) % 210)]
  bands = []
  x = mono
  decim = 1
  for d in (5, 7, 6):
    last = (d == 6)
    low = np.zeros(1, dtype=np.int64) if last else _group_fir(x, d, 3)
    decim *= d
    G = x.reshape(-1, d)
    be = sum((G[:, i] - low) ** 2 for i in range(d))
    fb = _group_fir(be, FRAME_RATE // decim, 15) / 210
    bands.append(np.log10(1 + fb) / 2.0)
    x = low
  return bands


def features(pcm_i16: np.ndarray):
  """The five feature rows combine() builds (describealign.py:1101-1104)."""
  arr = pcm_to_half(pcm_i16)
  return [energy(arr), zero_crossings(arr)] + freq_bands(arr)


# --------------------------------------------------------------------------- align: stage 1/2

def _hann_f64():
  w = scipy.signal.windows.hann(2 * HALF + 1)[1:-1]           # 41 taps, float64 (:597)
  return w, w / np.sum(w)


def mean_sub(f: np.ndarray) -> np.ndarray:
  """f - local Hann mean (describealign.py:598-599, :605-606)."""
  _, wn = _hann_f64()
  return f - np.convolve(wn, f, mode="same")[:len(f)]


def window_norm(ms: np.ndarray) -> np.ndarray:
  """L2 norm of each 41-frame window, floored at .001 (describealign.py:600-602)."""
  return np.clip(np.convolve(np.ones(WIN), ms ** 2, mode="valid") ** 0.5, 0.001, None)


def _taps(ms: np.ndarray, nrm: np.ndarray) -> np.ndarray:
  """(L-40, 7) matrix of ms[i + 2 + 6b] / norm[i]  (describealign.py:623-624, :639-640)."""
  n = len(nrm)
  cols = [ms[TAP_START + TAP_STEP * b: TAP_START + TAP_STEP * b + n] for b in range(N_TAPS)]
  return np.stack(cols, axis=1) / nrm[:, None]


def video_digits(ms, nrm):
  """Base-7 digits and the 'also probe digit+1' flags (describealign.py:625-628)."""
  u = np.clip(8 * _taps(ms, nrm) + 3.3, 0, 6)
  return np.floor(u).astype(np.int64), (u % 1) > 0.6


def audio_digits(ms, nrm):
  """Single-probe digits (describealign.py:641-643)."""
  return np.clip(np.floor(8 * _taps(ms, nrm) + 3.5).astype(np.int64), 0, 6)


POW7 = 7 ** np.arange(N_TAPS)


def video_rows(video_energy):
  """Every 4th non-quiet video frame (describealign.py:629-630)."""
  nq = video_energy[:-WIN] > QUIET
  return np.arange(len(video_energy) - WIN)[nq][::4]


def audio_rows(audio_energy):
  """Non-quiet audio frames (describealign.py:657-658)."""
  nq = audio_energy[:-WIN] > QUIET
  return np.arange(len(audio_energy) - WIN)[nq]


def _feature_hits(vd, vflag, vrows, ad, arows, n_video):
  """All (i, v) with audio code == one of video frame v's probe codes, as sorted keys i*n_video+v.

  Equivalent to the dict-of-sets insert (:630-633) + lookup (:659): frame v is stored under
  base + sum of 7^b over every subset of its flagged digits."""
  base = vd[vrows] @ POW7
  flags = vflag[vrows]
  # enumerate subsets of flagged digits (<=128 per frame), grouped by flag pattern
  pat = flags @ (1 << np.arange(N_TAPS))
  codes, owners = [], []
  for p in np.unique(pat):
    sel = np.nonzero(pat == p)[0]
    bits = [b for b in range(N_TAPS) if (p >> b) & 1]
    offs = np.zeros(1, dtype=np.int64)
    for b in bits:
      offs = np.concatenate([offs, offs + POW7[b]])
    codes.append((base[sel][:, None] + offs[None, :]).ravel())
    owners.append(np.repeat(vrows[sel], len(offs)))
  codes = np.concatenate(codes)
  owners = np.concatenate(owners)
  order = np.argsort(codes, kind="stable")
  codes, owners = codes[order], owners[order]
  acode = ad[arows] @ POW7
  lo = np.searchsorted(codes, acode, side="left")
  hi = np.searchsorted(codes, acode, side="right")
  cnt = hi - lo
  ii = np.repeat(arows, cnt)
  start = np.repeat(lo, cnt)
  within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
  vv = owners[start + within]
  return np.unique(ii.astype(np.int64) * n_video + vv)


def candidates(ms_v, nrm_v, ms_a, nrm_a, video_energy, audio_energy):
  """(i, v) pairs passing the hash vote: >=2 of features 0-2 and (feature 3 or 4)
  (describealign.py:649-653, :659-660).  Returns arrays sorted by (i, v)."""
  vrows, arows = video_rows(video_energy), audio_rows(audio_energy)
  n_video = len(video_energy) + 1
  hits = []
  for j in range(5):
    vd, vf = video_digits(ms_v[j], nrm_v[j])
    ad = audio_digits(ms_a[j], nrm_a[j])
    hits.append(_feature_hits(vd, vf, vrows, ad, arows, n_video))
  k012, c012 = np.unique(np.concatenate(hits[:3]), return_counts=True)
  two = k012[c012 >= 2]
  k34 = np.union1d(hits[3], hits[4])
  keys = np.intersect1d(two, k34, assume_unique=True)
  return (keys // n_video).astype(np.int64), (keys % n_video).astype(np.int64)


PROB_EXP = 2.9
PROB_MAX = 1e-8


def verify(ci, cv, ms_v, nrm_v, ms_a, nrm_a, chunk=1 << 16):
  """Windowed correlations, Naive-Bayes product, threshold and quality for candidate pairs
  (describealign.py:662-673).  Returns (corr[n,3], keep mask, qual)."""
  n = len(ci)
  corr = np.empty((n, 3))
  for j in range(3):
    Aw = np.lib.stride_tricks.sliding_window_view(ms_a[j], WIN)
    Vw = np.lib.stride_tricks.sliding_window_view(ms_v[j], WIN)
    for s in range(0, n, chunk):
      a, v = ci[s:s + chunk], cv[s:s + chunk]
      corr[s:s + chunk, j] = np.einsum("nk,nk->n", Aw[a], Vw[v]) / (nrm_a[j][a] * nrm_v[j][v])
  prob = np.prod(np.maximum(1e-8, 1 - corr), axis=1) ** PROB_EXP
  keep = ~(prob > PROB_MAX)
  with np.errstate(divide="ignore"):
    qual = np.minimum(50, (prob / 1e-12) ** (-1.0 / 3))
  return corr, keep, qual


def chain(mi, mv, mq):
  """Heaviest chain non-decreasing in both coordinates (describealign.py:654-656, :674-697).

  Points must be sorted by (i, v).  The frontier is a staircase of (v, cum) with cum strictly
  increasing; a new point takes the last frontier entry with v' <= v as predecessor, evicts
  entries to its right whose cum is not larger, and is inserted after any equal-v entries.
  Returns the indices (into the input arrays) of the chain's points, in order."""
  fv = [-1]        # frontier video index
  fc = [0.0]       # frontier cumulative quality
  fid = [-1]       # frontier point id
  pred = np.full(len(mi), -1, dtype=np.int64)
  for p in range(len(mi)):
    v = int(mv[p])
    pos = bisect.bisect_right(fv, v)
    cum = fc[pos - 1] + float(mq[p])
    pred[p] = fid[pos - 1]
    end = pos
    while end < len(fv) and fc[end] <= cum:
      end += 1
    fv[pos:end] = [v]
    fc[pos:end] = [cum]
    fid[pos:end] = [p]
  out = []
  p = fid[-1]
  while p >= 0:
    out.append(p)
    p = pred[p]
  out.reverse()
  return np.array(out, dtype=np.int64)


MISMATCH_MSG = "Alignment failed, are the input files mismatched?"


def min_path_len(n_video, n_audio):
  return max(min(n_video, n_audio) / 500.0, 5 * FRAME_RATE)     # (:698, :991)


# --------------------------------------------------------------------------- align: stage 3

def continuity_err(x, y, deriv=False):
  """describealign.py:702-724: disagreement of each path point with lines fitted through
  half-Hann-smoothed points 10 indices apart, looking forward and backward."""
  _, wn = _hann_f64()
  h = wn[:HALF - 1] / np.sum(wn[:HALF - 1])          # 20 taps
  half = HALF // 2                                    # 10
  delay = HALF + half - 2                             # 29
  xf = np.convolve(x, h, mode="valid"); yf = np.convolve(y, h, mode="valid")
  sf = (yf[half:] - yf[:-half]) / (xf[half:] - xf[:-half])
  of = yf[:-half] - xf[:-half] * sf
  xp = np.convolve(x, h[::-1], mode="valid"); yp = np.convolve(y, h[::-1], mode="valid")
  sp = (yp[half:] - yp[:-half]) / (xp[half:] - xp[:-half])
  op = yp[half:] - xp[half:] * sp
  n = len(x) - (1 if deriv else 0)
  err = np.full(n, np.inf)
  dly = delay - (1 if deriv else 0)
  err[:-dly] = np.abs(sf * x[:-delay] + of - y[:-delay])
  err[dly:] = np.minimum(err[dly:], np.abs(sp * x[delay:] + op - y[delay:]))
  return err


def scale_features(video_features, audio_features, x, y):
  """describealign.py:733-741: per-feature least-squares gain video->audio at the path, both in
  units of the audio feature's std; first three features stacked as (L, 3)."""
  a_s, v_s = [], []
  for vf, af in zip(video_features, audio_features):
    sd = np.std(af)
    g = np.linalg.lstsq(vf[y][:, None], af[x], rcond=None)[0]
    a_s.append(af / sd)
    v_s.append(vf * g / sd)
  na = min(len(f) for f in a_s[:3]); nv = min(len(f) for f in v_s[:3])
  return (np.stack([f[:na] for f in a_s[:3]], axis=1), np.stack([f[:nv] for f in v_s[:3]], axis=1))


def compress_path(x, y):
  """describealign.py:743-767: runs of 70 points collapse to their mean when every point of
  the run is within 3 frames of the locally smoothed line; equal-x points are merged."""
  _, wn = _hann_f64()
  sm = lambda a: np.convolve(wn, a, mode="same")[:len(a)]
  sx, sy = sm(x), sm(y)
  with np.errstate(divide="ignore", invalid="ignore"):
    sl = np.diff(sy) / np.diff(sx)
    dev = sl * x[:-1] + (sy[:-1] - sx[:-1] * sl) - y[:-1]
  cx, cy = list(x[:10]), list(y[:10])
  i = None
  for i in range(10, len(x) - 80, 70):
    if np.all(np.abs(dev[i:i + 70]) < 3):
      cx.append(np.mean(x[i:i + 70])); cy.append(np.mean(y[i:i + 70]))
    else:
      cx.extend(x[i:i + 70]); cy.extend(y[i:i + 70])
  # the reference's tail call takes at most 70 more points (`extend_all(i+70)` with num=70, :755),
  # so up to 10 trailing points can be dropped; mirrored here.
  cx.extend(x[i + 70:i + 140]); cy.extend(y[i + 70:i + 140])
  groups = defaultdict(list)
  order = []
  last = -1
  for a, v in zip(cx, cy):
    groups[a].append(v)
    if a != last:
      order.append(a); last = a
  ux = np.array(order)
  uy = np.array([np.mean(groups[a]) for a in ux])
  return ux, uy


def build_lp(x, y):
  """describealign.py:773-840: the L1 trend-fit LP (variable/row layout: SURVEY appendix A.6)."""
  n = len(x)
  dx, dy = np.diff(x), np.diff(y)
  jump = np.full(n - 1, 10.0) / np.maximum(1, np.sqrt(continuity_err(x, y, deriv=True) / 3.0))
  c = np.concatenate([np.ones(2 * n), jump, jump, np.full(2 * n, 0.01), np.full(2 * (n - 1), 3.0),
                      np.full(2 * (n - 1), 0.001), np.full(2 * (n - 2), 40000.0), [0.0]])
  S = scipy.sparse
  D = S.diags([-1.0 / dx, 1.0 / dx], offsets=[0, 1], shape=(n - 1, n)).tocsc()
  J = S.diags([1.0 / dx], offsets=[0], shape=(n - 1, n - 1)).tocsc()
  Z = lambda r, k: S.csc_matrix((r, k))
  row1 = S.hstack([D, -D, J, -J, Z(n - 1, 2 * n), J, -J, J, -J, Z(n - 1, 2 * n - 4), np.ones((n - 1, 1))])
  T = S.diags([-1.0, 1.0], offsets=[0, 1], shape=(n - 1, n)).tocsc()
  I1 = S.eye(n - 1)
  row2 = S.hstack([Z(n - 1, 4 * n - 2), T, -T, -I1, I1, Z(n - 1, 4 * n - 6), Z(n - 1, 1)])
  R = S.diags([-1.0 / dx[:-1], 1.0 / dx[1:]], offsets=[0, 1], shape=(n - 2, n - 1)).tocsc()
  I2 = S.eye(n - 2)
  row3 = S.hstack([Z(n - 2, 8 * n - 4), R, -R, -I2, I2, Z(n - 2, 1)])
  A = S.vstack([row1, row2, row3])
  b = np.concatenate([dy / dx, np.zeros(2 * n - 3)])
  bounds = [[0, None]] * (4 * n - 2) + [[0, 2.0]] * (2 * n) + [[0, None]] * (6 * n - 8) + [[None, None]]
  return c, A, b, bounds


LP_FAIL_MSG = "Smooth Alignment L1-Min Optimization Failed!"


def solve_lp(x, y):
  """describealign.py:841-858."""
  c, A, b, bounds = build_lp(x, y)
  fit = scipy.optimize.linprog(c, A_eq=A, b_eq=b, bounds=bounds, method="highs-ds")
  if not fit.success and fit.status == 4:
    fit = scipy.optimize.linprog(c, A_eq=A, b_eq=b, bounds=bounds, method="highs-ipm")
  if not fit.success:
    raise RuntimeError(LP_FAIL_MSG)
  n = len(x)
  fit_err = fit.x[:n] - fit.x[n:2 * n]
  rate_jump = fit.x[8 * n - 4:9 * n - 5] - fit.x[9 * n - 5:10 * n - 6]
  median_slope = fit.x[-1]
  slopes = median_slope + rate_jump / np.diff(x)
  return dict(sol=fit.x, fit_err=fit_err, slopes=slopes, median_slope=median_slope,
              smooth_x=np.asarray(x, dtype=np.float64), smooth_y=y - fit_err)


# --------------------------------------------------------------------------- align: stage 4

def line_clusters(sx, sy, slopes):
  """describealign.py:861-893: group fit points by (rounded slope, rounded offset) of the
  segments on either side, greedily merge groups whose end points lie within 3 frames of a
  bigger group's line, keep spans > 10 with > 5 points, refit each by least squares.
  Returns [(x array, offset, slope)]."""
  ext = np.concatenate([slopes[:1], slopes, slopes[-1:]])
  groups = defaultdict(list)
  for i, (px, py) in enumerate(zip(sx, sy)):
    for s in ext[i:i + 2]:
      if s < 0.1 or s > 10:
        continue
      groups[(round(s, 6), int(round(py - s * px, 0)))].append((px, py))
  clusters = []
  done = set()
  for key, pts in sorted(groups.items(), key=lambda kv: -len(kv[1])):
    if key in done:
      continue
    s, o = key
    cur = pts
    clusters.append(cur)
    done.add(key)
    del groups[key]
    for key2, pts2 in list(groups.items()):
      if abs(pts2[0][1] - (pts2[0][0] * s + o)) < 3 and abs(pts2[-1][1] - (pts2[-1][0] * s + o)) < 3:
        cur.extend(groups[key2])
        done.add(key2)
        del groups[key2]
  clusters = [sorted(c) for c in clusters]
  clusters = [c for c in clusters if abs(c[0][0] - c[-1][0]) > 10 and len(c) > 5]
  out = []
  for c in clusters:
    cx, cy = np.array(c).T
    sol = np.linalg.lstsq(np.stack([np.ones(len(cx)), cx], axis=1), cy, rcond=None)[0]
    out.append((cx, sol[0], sol[1]))
  return out


EXTEND = FRAME_RATE * 30


def _limits(cx, offset, slope, n_audio, n_video, extend, margin=4):
  lo = max(int(cx[0]) - extend, 0)
  hi = min(int(cx[-1]) + extend, n_audio - 1)
  lo = max(lo, int(np.ceil((margin - offset) / slope)))
  hi = min(hi, int(np.floor((n_video - margin - offset) / slope)))
  return lo, hi


def extend_clusters(clusters, a_scaled, v_scaled):
  """describealign.py:895-944: sub-frame offset refinement, then evaluation of each cluster's
  line +-30 s beyond its span against linearly interpolated video features.
  Returns per-audio-frame sorted lists of (j, cluster, qual) and the refined offsets."""
  n_audio, n_video = len(a_scaled), len(v_scaled)
  interp = scipy.interpolate.make_interp_spline(np.arange(n_video), v_scaled, k=1)
  a_max = np.max(a_scaled[:, 0]); v_max = np.max(v_scaled[:, 0])
  points = [[] for _ in range(n_audio)]
  seen = set()
  refined = []
  for ci, (cx, offset, slope) in enumerate(clusters):
    lo, hi = _limits(cx, offset, slope, n_audio, n_video, 0)
    if hi < lo + 5:
      refined.append(np.nan)
      continue
    span = cx
    if hi > lo + 100:
      xs = np.arange(lo, hi)
      span = xs      # the reference rebinds `x` here (:917), so the +-30 s limits below start from it
      am = a_scaled[lo:hi]; vm = interp(slope * xs + offset)
      err = am[1:-1] - vm[1:-1]
      ok = np.mean(err, axis=-1) < 0.1
      if np.count_nonzero(ok) > 50:
        dv = ((vm[2:] - vm[:-2]) / 2.0)[ok]
        err = err[ok]
        sol, resid, _, _ = np.linalg.lstsq(dv.reshape(-1, 1), err.flat, rcond=None)
        explained = 1 - (resid / np.sum(err ** 2))
        z = np.sqrt(explained * np.prod(err.shape)) - 1.0
        if z > 8 and abs(sol[0]) < 2:
          offset += sol[0]
    refined.append(offset)
    lo, hi = _limits(span, offset, slope, n_audio, n_video, EXTEND)
    xs = np.arange(lo, hi)
    ys = slope * xs + offset
    am = a_scaled[lo:hi]; vm = interp(ys)
    q = np.sum(-0.5 - np.log10(1e-4 + np.abs(am - vm)), axis=1)
    q *= np.clip(vm[:, 0] + 2.5 - v_max, 0, 1)
    q += np.clip(am[:, 0] + 2.5 - a_max, 0, 1) * 0.1
    for i, j, qq in zip(xs.tolist(), ys.tolist(), q.tolist()):
      key = (i, int(j))
      if key not in seen:
        seen.add(key)
        points[i].append((j, ci, qq))
  return [sorted(p) for p in points], refined


def second_dp(points, n_clusters, n_video):
  """describealign.py:946-990.  Frontier DP over the banded points with a -1000 free-jump
  penalty, a -50 same-cluster rejoin penalty and a local (3 video frames, 2 audio frames)
  continuation whose cluster switches cost 100 + 100*skew^2.  Returns rows
  (video_idx, audio_idx, cluster, qual, cum_qual)."""
  fj = [0]                                   # frontier keys (video index), sorted
  fe = [(0, 0, -1, 0, 0)]                    # frontier entries
  cl_best = [(0, 0, 0, -1000)] * n_clusters  # (j, i, qual, cum - 50) per cluster
  cl_best = list(cl_best)
  back = {}
  cache = np.full((n_video, 5), -np.inf)
  cache[0] = (0, 0, -1, 0, 0)
  fmin = [np.inf] * (len(points) + 1)
  for i in range(len(points) - 1, -1, -1):
    m = points[i][0][0] if points[i] else np.inf     # points[i] is sorted, so [0] is the min tuple
    fmin[i] = min(m, fmin[i + 1])
  for i, pts in enumerate(points):
    for j, ci, q in pts:
      pos = bisect.bisect_right(fj, j)
      pj, pi, pc, pq, best = fe[pos - 1]
      last = cl_best[ci]
      if last[3] >= best:
        pj, pi, pq, best = last
        pc = ci
      for jj in range(max(0, int(j) - 2), int(j) + 1):
        node = cache[jj].tolist()
        if ci != node[2]:
          node[4] -= 100 + 100 * ((j - node[0]) - (i - node[1])) ** 2
        if node[1] >= (i - 2) and node[0] <= j and node[4] >= best:
          pj, pi, pc, pq, best = node
      cum = best + q
      cache[int(j)] = (j, i, ci, q, cum)
      cj = cum - 1000
      if fe[pos - 1][4] < cj:
        end = pos
        while end < len(fe) and fe[end][4] <= cj:
          end += 1
        fj[pos:end] = [j]
        fe[pos:end] = [(j, i, ci, q, cj)]
      if fmin[i] == j and pos > 1:
        del fj[:pos - 1]
        del fe[:pos - 1]
      cc = cum - 50
      if last[3] < cc:
        cl_best[ci] = (j, i, q, cc)
      back[(j, i)] = (pj, pi, pc, pq, best)
  path = [fe[-1]]
  while path[-1][:2] in back:
    path.append(back[path[-1][:2]])
  path.pop()
  path.reverse()
  return np.array(path, dtype=np.float64).reshape(-1, 5)


def finish(path, n_audio_scaled, n_video_scaled, n_audio_energy, n_video_energy):
  """describealign.py:993-1027: similarity percentage, nodes at cluster changes, end
  extrapolation; path columns 0-1 converted to seconds."""
  y, x, cl, q, _ = path.T
  solid = (q == 0) | (q > 0.3)
  sim = 100 * max(len(set(x[solid])) / n_audio_scaled, len(set(y[solid])) / n_video_scaled)
  nodes = []
  if cl[0] == cl[1]:
    nodes.append((x[0], y[0]))
  for k in range(len(x) - 1):
    if cl[k] != cl[k + 1]:
      nodes.append((x[k] - 0.1, y[k] - 0.1))
      nodes.append((x[k + 1] + 0.1, y[k + 1] + 0.1))
  if cl[-2] == cl[-1]:
    nodes.append((x[-1], y[-1]))
  nx, ny = np.array(nodes).T / float(FRAME_RATE)
  if nx[1] - nx[0] > 2:
    s = (ny[1] - ny[0]) / (nx[1] - nx[0])
    nx[0] = 0
    ny[0] = ny[1] - nx[1] * s
    if ny[0] < 0:
      nx[0] = nx[1] - ny[1] / s
      ny[0] = 0
  if nx[-1] - nx[-2] > 2:
    s = (ny[-1] - ny[-2]) / (nx[-1] - nx[-2])
    nx[-1] = (n_audio_energy - 1) / float(FRAME_RATE)
    ny[-1] = ny[-2] + (nx[-1] - nx[-2]) * s
    v_end = (n_video_energy - 1) / float(FRAME_RATE)
    if ny[-1] > v_end:
      ny[-1] = v_end
      nx[-1] = nx[-2] + (ny[-1] - ny[-2]) / s
  out = path.copy()
  out[:, :2] /= float(FRAME_RATE)
  return nx, ny, sim, out


# --------------------------------------------------------------------------- whole align()

def align(video_features, audio_features, video_energy, audio_energy, stages=None):
  """describealign.py:595-1027.  Returns (audio_times, video_times, similarity_percent, path,
  median_slope); `stages`, if a dict, receives the intermediates."""
  st = stages if stages is not None else {}
  ms_v = [mean_sub(np.asarray(f)) for f in video_features]
  ms_a = [mean_sub(np.asarray(f)) for f in audio_features]
  nrm_v = [window_norm(m) for m in ms_v]
  nrm_a = [window_norm(m) for m in ms_a]
  ci, cv = candidates(ms_v, nrm_v, ms_a, nrm_a, video_energy, audio_energy)
  corr, keep, qual = verify(ci, cv, ms_v, nrm_v, ms_a, nrm_a)
  mi, mv, mq = ci[keep], cv[keep], qual[keep]
  st.update(cand_i=ci, cand_v=cv, corr=corr, m_i=mi, m_v=mv, m_q=mq)
  idx = chain(mi, mv, mq)
  if len(idx) < min_path_len(len(video_energy), len(audio_energy)):
    raise RuntimeError(MISMATCH_MSG)
  x, y = mi[idx], mv[idx]
  st.update(p1_x=x, p1_y=y)
  ok = continuity_err(x, y) < 3
  x, y = x[ok], y[ok]
  a_scaled, v_scaled = scale_features(video_features, audio_features, x, y)
  fx, fy = compress_path(x, y)
  lp = solve_lp(fx, fy)
  st.update(lp_x=fx, lp_y=fy, lp=lp, a_scaled=a_scaled, v_scaled=v_scaled)
  clusters = line_clusters(lp["smooth_x"], lp["smooth_y"], lp["slopes"])
  points, refined = extend_clusters(clusters, a_scaled, v_scaled)
  st.update(clusters=clusters, points=points, refined_offsets=refined)
  path = second_dp(points, len(clusters), len(v_scaled))
  if len(path) < min_path_len(len(video_energy), len(audio_energy)):
    raise RuntimeError(MISMATCH_MSG)
  nx, ny, sim, path = finish(path, len(a_scaled), len(v_scaled), len(audio_energy), len(video_energy))
  return nx, ny, sim, path, lp["median_slope"]
