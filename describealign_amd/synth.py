"""Deterministic synthetic programme / audio-description PCM pairs with known offsets.

Everything here is integer arithmetic on numpy uint64/int64 arrays (a counter-based
splitmix64 generator, box filters via integer cumulative sums, power-of-two envelope
segments so every scale is a shift), so the same seed gives bit-identical int16 PCM on
every machine, numpy build, chunk size and thread count.  That lets tests pin golden
fixtures by (seed, sha1-of-PCM) instead of shipping megabytes of PCM.

Recipe: a "programme" is the sum of two band-limited noise
carriers (roughly 0.2-1.2 kHz and 2-6 kHz) with a slowly varying mix, under a ~5 Hz
piecewise-linear random syllable envelope in which ~20 % of the envelope nodes are silent,
peak-limited to +-20000.  The audio-description (AD) track is 0.7 x the programme re-timed
by inserting gaps (the injected offset jumps) plus 0.6 x an independent "narration"
programme that is gated on for ~40 % of the time.
"""
from __future__ import annotations

import hashlib
import os
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field

import numpy as np

SAMPLE_RATE = 44100
PEAK = 20000

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)

_ENV_SHIFT = 13    # 8192 samples  = 0.186 s  (~5.4 Hz syllable rate)
_MIX_SHIFT = 16    # 65536 samples = 1.49 s
_GATE_SHIFT = 17   # 131072 samples = 2.97 s
_CHUNK = 1 << 21


def _key(seed: int, stream: int) -> np.uint64:
  return np.uint64((int(seed) * 0xD1342543DE82EF95 + int(stream) * 0xA24BAED4963EE407
                    + 0x632BE59BD9B4E019) & (2**64 - 1))


def rand_u64(seed: int, stream: int, n: int, start: int = 0) -> np.ndarray:
  """n uint64 values of the (seed, stream) sequence starting at position `start`."""
  with np.errstate(over="ignore"):
    x = np.arange(start, start + n, dtype=np.uint64) * _GOLD + _key(seed, stream)
    x = (x ^ (x >> np.uint64(30))) * _M1
    x = (x ^ (x >> np.uint64(27))) * _M2
    return x ^ (x >> np.uint64(31))


def _noise_i16(seed: int, stream: int, start: int, n: int) -> np.ndarray:
  """int16 white noise samples [start, start+n) of the stream; 4 samples per u64 word."""
  w0 = start >> 2
  w1 = (start + n + 3) >> 2
  words = rand_u64(seed, stream, w1 - w0, w0)
  return words.view(np.int16)[start - 4 * w0: start - 4 * w0 + n]


def _box_band(seed: int, stream: int, a: int, n: int, short: int, long: int) -> np.ndarray:
  """Zero-DC band-pass of the noise stream on [a, a+n):
  long*boxsum(short) - short*boxsum(long); samples before position 0 are zero."""
  h = min(a, long)
  cs = np.zeros(n + long + 1, dtype=np.int64)
  np.cumsum(_noise_i16(seed, stream, a - h, n + h), out=cs[long - h + 1:])
  cs[:long - h + 1] = 0
  top = cs[long + 1:long + 1 + n]
  return long * (top - cs[long + 1 - short:long + 1 - short + n]) - short * (top - cs[1:1 + n])


def _lerp(levels: np.ndarray, shift: int, a: int, n: int) -> np.ndarray:
  """Integer linear interpolation between per-segment node levels; result scaled by 2**shift."""
  k = np.arange(a, a + n, dtype=np.int64)
  idx = k >> shift
  r = k & ((1 << shift) - 1)
  lo = levels[idx]
  return (lo << shift) + (levels[idx + 1] - lo) * r


def _programme_chunk(seed: int, sb: int, a: int, n: int, mix_lv, env_lv, out: np.ndarray):
  low = _box_band(seed, sb + 1, a, n, 36, 220) >> 12     # ~0.2-1.2 kHz
  high = _box_band(seed, sb + 2, a, n, 7, 22) >> 6       # ~2-6 kHz
  w = _lerp(mix_lv, _MIX_SHIFT, a, n)                    # 0 .. 64<<16
  carrier = (low * w + high * ((64 << _MIX_SHIFT) - w)) >> (6 + _MIX_SHIFT)
  env = _lerp(env_lv, _ENV_SHIFT, a, n)                  # 0 .. 1023<<13
  o = (carrier * env) >> (10 + _ENV_SHIFT)
  np.clip(o, -PEAK, PEAK, out=o)
  out[a:a + n] = o


def _threads() -> int:
  return max(1, min(8, (os.cpu_count() or 1)))


def programme(seed: int, n: int, stream_base: int = 0, dtype=np.int32) -> np.ndarray:
  """n samples of the programme for (seed, stream_base); |x| <= PEAK."""
  nm = (n >> _MIX_SHIFT) + 2
  mix_lv = (rand_u64(seed, stream_base + 3, nm) >> np.uint64(58)).astype(np.int64)        # 0..63
  ne = (n >> _ENV_SHIFT) + 2
  u = rand_u64(seed, stream_base + 4, ne)
  env_lv = ((u >> np.uint64(54)).astype(np.int64) % 768) + 256                            # 256..1023
  env_lv[(u & np.uint64(0xFFFF)).astype(np.int64) < int(0.20 * 65536)] = 0                # silent nodes
  out = np.empty(n, dtype=dtype)
  starts = list(range(0, n, _CHUNK))
  with ThreadPoolExecutor(_threads()) as ex:
    list(ex.map(lambda a: _programme_chunk(seed, stream_base, a, min(_CHUNK, n - a), mix_lv, env_lv, out),
                starts))
  return out


@dataclass
class SynthPair:
  video: np.ndarray            # int16 (C, Nv)
  audio: np.ndarray            # int16 (C, Na)
  jump_video_times: list = field(default_factory=list)   # video-timeline seconds where AD-only material is inserted
  jump_lengths: list = field(default_factory=list)       # seconds inserted at each
  seed: int = 0
  rate_change: float = 0.0

  @property
  def video_seconds(self) -> float:
    return self.video.shape[1] / SAMPLE_RATE

  @property
  def audio_seconds(self) -> float:
    return self.audio.shape[1] / SAMPLE_RATE

  def offsets(self):
    """[(video_time_start, audio_time - video_time)] piecewise-constant truth (rate_change == 0)."""
    out, acc = [], 0.0
    for t, g in zip(self.jump_video_times, self.jump_lengths):
      acc += g
      out.append((t, acc))
    return out

  def true_offset_at(self, video_time: float) -> float:
    acc = 0.0
    for t, g in zip(self.jump_video_times, self.jump_lengths):
      if video_time >= t:
        acc += g
    return acc

  def sha1(self) -> str:
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(self.video).tobytes())
    h.update(np.ascontiguousarray(self.audio).tobytes())
    return h.hexdigest()


def make_jumps(seed: int, video_seconds: float, n_jumps: int, first_gap: float):
  """First gap at video time 0 (podcast-style intro), then n_jumps gaps of 1-6 s, one per slot."""
  times, lengths = [0.0], [float(first_gap)]
  if n_jumps > 0:
    u = rand_u64(seed, 90, 2 * n_jumps)
    slot = video_seconds / (n_jumps + 1)
    for k in range(n_jumps):
      frac = float(u[2 * k] >> np.uint64(40)) / float(1 << 24)
      t = slot * (k + 1) + slot * 0.5 * (frac - 0.5)
      glen = 1.0 + 5.0 * float(u[2 * k + 1] >> np.uint64(40)) / float(1 << 24)
      times.append(round(t, 3))
      lengths.append(round(glen, 3))
  return times, lengths


def make_pair(seed: int, video_seconds: float, n_jumps: int = 10, first_gap: float = 200.0,
              channels: int = 1, jumps=None, rate_change: float = 0.0) -> SynthPair:
  """Build a (video, AD) pair.  `jumps` = (times, lengths) overrides the random placement.

  rate_change != 0 re-times the AD copy of the programme by nearest-sample index mapping so
  that d(video)/d(audio) = 1 + rate_change (exercises the non-unit-slope paths).
  """
  nv = int(round(video_seconds * SAMPLE_RATE))
  prog = programme(seed, nv, 0)
  times, lengths = jumps if jumps is not None else make_jumps(seed, video_seconds, n_jumps, first_gap)
  cuts = [int(round(t * SAMPLE_RATE)) for t in times] + [nv]
  gaps = [int(round(g * SAMPLE_RATE)) for g in lengths]
  if rate_change != 0.0:
    na_prog = int(nv / (1.0 + rate_change))
    src = (np.arange(na_prog, dtype=np.int64) * nv) // na_prog
    prog_ad = prog[src]
    cuts = [int(c * na_prog // nv) for c in cuts]
  else:
    prog_ad = prog
  na = len(prog_ad) + sum(gaps)
  audio = programme(seed, na, 10)                       # narration
  seg_bits = _GATE_SHIFT
  ng = (na >> seg_bits) + 2
  gate_lv = ((rand_u64(seed, 20, ng) >> np.uint64(48)).astype(np.int64) < int(0.40 * 65536)).astype(np.int64)
  gate_lv[: max(1, gaps[0] >> seg_bits)] = 1            # narration is on during the intro gap
  for a in range(0, na, _CHUNK):
    n = min(_CHUNK, na - a)
    g = _lerp(gate_lv, seg_bits, a, n)                  # 0 .. 1<<17
    audio[a:a + n] = (audio[a:a + n].astype(np.int64) * g * 6) >> seg_bits
  pos = 0
  if cuts[0] > 0:
    audio[:cuts[0]] += 7 * prog_ad[:cuts[0]]
    pos = cuts[0]
  for k, g in enumerate(gaps):
    pos += g
    n = cuts[k + 1] - cuts[k]
    audio[pos:pos + n] += 7 * prog_ad[cuts[k]:cuts[k + 1]]
    pos += n
  audio //= 10
  np.clip(audio, -32767, 32767, out=audio)
  video = prog
  if channels == 1:
    v = video.astype(np.int16)[None, :]
    a = audio.astype(np.int16)[None, :]
  else:
    # second channel = 0.9 x first + independent low-level noise
    v2 = np.clip((9 * video) // 10 + (_noise_i16(seed, 30, 0, nv) >> 7), -32767, 32767)
    a2 = np.clip((9 * audio) // 10 + (_noise_i16(seed, 31, 0, na) >> 7), -32767, 32767)
    v = np.stack([video.astype(np.int16), v2.astype(np.int16)])
    a = np.stack([audio.astype(np.int16), a2.astype(np.int16)])
  return SynthPair(video=v, audio=a, jump_video_times=list(times), jump_lengths=list(lengths),
                   seed=seed, rate_change=rate_change)


def unrelated_pair(seed: int, video_seconds: float, audio_seconds: float) -> SynthPair:
  """Two independent programmes (expects the 'mismatched files' alignment failure)."""
  v = programme(seed, int(video_seconds * SAMPLE_RATE), 0).astype(np.int16)[None, :]
  a = programme(seed + 7919, int(audio_seconds * SAMPLE_RATE), 40).astype(np.int16)[None, :]
  return SynthPair(video=v, audio=a, seed=seed)
