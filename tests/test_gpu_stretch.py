"""Parity of the audio replacement path (--stretch_audio; describealign.py:230-416, :1135-1153)
through the C ABI against fixtures recorded from the reference and against the oracle.
Needs a real MI355X: run with `pytest -m gpu`."""
import json
import os

import numpy as np
import pytest

import cases
from oracle import stretch_oracle as SO

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))


@pytest.fixture(scope="module")
def ctx():
  from describealign_amd import _native
  c = _native.Context(0, _native.PREC_F32)
  yield c
  c.close()


def _bits(a):
  return np.ascontiguousarray(a).view(np.uint16)


def _canon(a):
  """float16 bit patterns with -0 folded onto +0.  Inside digital silence the spline rings at
  ~1e-20, which rounds to a zero whose sign depends on the last bit of the banded solve (LAPACK's
  LU in scipy, a Thomas recurrence on the GPU); every other value is compared bit for bit, and
  the int16 output of the path is identical either way."""
  b = _bits(a).copy()
  b[(b & 0x7fff) == 0] = 0
  return b


def _report(got, want, what):
  """float16 arrays must agree bit for bit; on failure say how far apart they are."""
  if np.array_equal(_canon(got), _canon(want)):
    return
  bad = np.flatnonzero((_canon(got) != _canon(want)).any(axis=0))
  d = np.abs(got[:, bad].astype(np.float64) - want[:, bad].astype(np.float64))
  raise AssertionError(f"{what}: {len(bad)} of {got.shape[1]} samples differ, first at {bad[:5]}, max |diff| {d.max()}")


@pytest.mark.parametrize("name", list(cases.STRETCH_CASES))
def test_replace_segments_matches_reference_fixture(ctx, name):
  """Jump schedules identical, every replaced interval identical float16 bits, untouched
  intervals untouched."""
  g = np.load(os.path.join(GOLD, f"stretch_{name}.npz"))
  meta = INDEX["stretch"][name]
  v, a, x, y = cases.stretch_case_f16(name)
  before = v.copy()
  ctx.replace_segments(v, a, x, y, False)
  sched = ctx.stretch_schedules()
  assert len(sched) == meta["n_schedules"]
  for k, s in enumerate(sched):
    assert np.array_equal(s, g[f"sched{k}"]), f"jump schedule {k}: {s.tolist()} vs {g[f'sched{k}'].tolist()}"
  for k, (kind, x0, x1, y0, y1) in enumerate(SO.segment_plan(x, y, False)):
    if k in meta["replaced_intervals"]:
      _report(v[:, y0:y1], g[f"seg{k}"].view(np.float16), f"{name} interval {k} ({kind})")
    else:
      assert np.array_equal(_bits(v[:, y0:y1]), _bits(before[:, y0:y1])), f"interval {k} should be untouched"


@pytest.mark.parametrize("name", list(cases.STRETCH_CASES))
def test_replace_segments_without_pitch_correction(ctx, name):
  """no_pitch_correction: every kept interval is resampled.  The oracle's output for this input is
  pinned to the reference by sha1 in tests/test_oracle_golden.py."""
  v, a, x, y = cases.stretch_case_f16(name)
  want = v.copy()
  SO.replace_aligned_segments(want, a, x, y, True)
  ctx.replace_segments(v, a, x, y, True)
  assert ctx.stretch_schedules() == []
  _report(v, want, name)


@pytest.mark.parametrize("name", ["mix_stereo", "edge_mono"])
def test_stretch_resident_matches_oracle(ctx, name):
  """The whole --stretch_audio block from int16 PCM to int16 interleaved frames."""
  vid, aud, x, y = cases.stretch_case(name)
  v, a = vid.astype(np.float16), aud.astype(np.float16)
  want_f = SO.match_loudness(v, a)
  SO.replace_aligned_segments(v, a, x, y, False)
  SO.normalise_peak(v)
  want = v.astype(np.int16).T
  ctx.pcm_upload(0, vid); ctx.pcm_upload(1, aud)
  got, fac = ctx.stretch_resident(x, y, False)
  np.testing.assert_allclose(fac, want_f, rtol=1e-12)
  assert got.shape == want.shape
  nbad = int((got != want).sum())
  assert nbad == 0, f"{nbad} of {want.size} samples differ; max |diff| {np.abs(got.astype(int) - want.astype(int)).max()}"
  # interleaved upload gives the same result
  ctx.pcm_upload(0, np.ascontiguousarray(vid.T)); ctx.pcm_upload(1, np.ascontiguousarray(aud.T))
  got2, _ = ctx.stretch_resident(x, y, False)
  assert np.array_equal(got, got2)


def test_resample_many_blocks_vs_oracle(ctx):
  """A 12 s interval is 6 blocks of 1e5 points, each with its own spline end conditions."""
  rng = np.random.default_rng(7)
  n = 14 * 44100
  a = (rng.standard_normal((2, n)) * 4000).astype(np.float16)
  v = np.zeros((2, n), dtype=np.float16)
  x = np.array([0.25, 12.28, 13.9]); y = np.array([0.5, 12.5, 13.8])
  want = v.copy()
  SO.replace_aligned_segments(want, a, x, y, False)
  ctx.replace_segments(v, a, x, y, False)
  _report(v, want, "random stereo")


def test_stretch_random_intervals_vs_oracle(ctx):
  """Both directions and all three lag sets on noise with silent stretches (ties in the arg-max)."""
  rng = np.random.default_rng(11)
  n = 20 * 44100
  a = (rng.standard_normal((1, n)) * 3000).astype(np.float16)
  a[:, 3 * 44100:4 * 44100] = 0                               # digital silence inside a stretched interval
  env = np.clip(np.sin(np.arange(n) / 9000.0), 0, 1) ** 2
  a = (a.astype(np.float32) * env.astype(np.float32)).astype(np.float16)
  v = np.zeros((1, n), dtype=np.float16)
  y = np.array([0.0, 6.0, 9.0, 15.5, 19.0]); x = np.array([0.1, 6.25, 9.232, 15.6, 19.12])
  kinds = [p[0] for p in SO.segment_plan(x, y, False)]
  assert kinds.count("stretch") >= 3, kinds
  want = v.copy()
  want_s = SO.replace_aligned_segments(want, a, x, y, False)
  ctx.replace_segments(v, a, x, y, False)
  got_s = ctx.stretch_schedules()
  assert len(got_s) == len(want_s)
  for k, (g_, w_) in enumerate(zip(got_s, want_s)):
    assert np.array_equal(g_, w_), f"schedule {k}"
  _report(v, want, "random mono")


def test_replace_segments_errors(ctx):
  v = np.zeros((1, 44100 * 5), dtype=np.float16); a = np.zeros((1, 44100 * 5), dtype=np.float16)
  with pytest.raises(RuntimeError, match="outside"):
    ctx.replace_segments(v, a, np.array([0.0, 6.03]), np.array([0.0, 6.0]), True)     # video interval past the end
  with pytest.raises(RuntimeError, match="two nodes"):
    ctx.replace_segments(v, a, np.array([0.0]), np.array([0.0]), False)
  with pytest.raises(RuntimeError, match="finite"):
    ctx.replace_segments(v, a, np.array([0.0, np.nan]), np.array([0.0, 3.0]), False)
  with pytest.raises(ValueError):
    ctx.replace_segments(v.astype(np.float32), a, np.array([0.0, 3.0]), np.array([0.0, 3.0]), False)


def test_combine_stretch_audio_end_to_end(ctx, tmp_path):
  """python -m describealign_amd.combine --stretch_audio on two stereo .wav files: the written track
  equals the oracle's replace pipeline run with the nodes the alignment found."""
  import wave
  from describealign_amd import combine, media, synth
  pair = synth.make_pair(seed=21, video_seconds=40.0, jumps=([0.0, 20.0], [5.0, 2.0]), channels=2)
  vfile, afile = str(tmp_path / "show.wav"), str(tmp_path / "show_ad.wav")
  media.write_wav(vfile, pair.video); media.write_wav(afile, pair.audio)
  out_dir, plot_dir = str(tmp_path / "out"), str(tmp_path / "plots")
  os.makedirs(out_dir); os.makedirs(plot_dir)
  res = combine.process_pair(vfile, afile, True, ctx, stretch_audio=True, output_dir=out_dir, alignment_dir=plot_dir)
  x, y = res["audio_desc_times"], res["video_times"]
  offs = np.round(np.asarray(x) - np.asarray(y), 1)
  assert 5.0 in offs and 7.0 in offs, offs                        # the two injected offsets
  v, a = pair.video.astype(np.float16), pair.audio.astype(np.float16)
  SO.match_loudness(v, a)
  SO.replace_aligned_segments(v, a, x, y, False)
  SO.normalise_peak(v)
  want = v.astype(np.int16)
  if media.find_ffmpeg() is None:
    with wave.open(os.path.join(out_dir, "ad_show.wav"), "rb") as w:
      got = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").reshape(-1, 2).T
    assert np.array_equal(got, want)
  text = open(res["report"]).read()
  assert "'stretch_audio': True" in text


@pytest.mark.parametrize("seed", range(6))
def test_replace_segments_random_plans_vs_oracle(ctx, seed):
  """Random node lists over noise with silent gaps: every interval kind, both jump directions,
  all three lag sets, mono and stereo, ragged interval lengths."""
  rng = np.random.default_rng(100 + seed)
  ch = 1 + seed % 2
  rates = [1.0, 1.003, 0.9965, 1.02, 0.985, 1.06, 0.93, 1.0072, 0.9931, 1.2]
  ys, xs = [0.37 * rng.random()], [0.5 * rng.random()]
  for k in range(5 + seed % 3):
    dy = float(rng.choice([1.5, 2.1, 2.6, 3.3, 4.2])) + 0.01 * rng.random()
    r = float(rng.choice(rates))
    ys.append(ys[-1] + dy); xs.append(xs[-1] + dy * r)
  n_v = int(ys[-1] * 44100) + 1000
  n_a = int(xs[-1] * 44100) + 1000
  a = rng.standard_normal((ch, n_a)) * 2500
  env = np.clip(np.sin(np.arange(n_a) / (3000.0 + 500 * seed)) + 0.3, 0, 1)
  a = (a * env).astype(np.float16)
  a[:, n_a // 3: n_a // 3 + 20000] = 0                     # digital silence
  v = (rng.standard_normal((ch, n_v)) * 2500).astype(np.float16)
  x, y = np.array(xs), np.array(ys)
  want = v.copy()
  want_s = SO.replace_aligned_segments(want, a, x, y, False)
  ctx.replace_segments(v, a, x, y, False)
  got_s = ctx.stretch_schedules()
  assert len(got_s) == len(want_s), [p[0] for p in SO.segment_plan(x, y, False)]
  for k, (g_, w_) in enumerate(zip(got_s, want_s)):
    assert np.array_equal(g_, w_), f"schedule {k}: {g_.tolist()} vs {w_.tolist()}"
  _report(v, want, f"seed {seed}")


def test_interval_too_short_for_the_correlation_generator(ctx):
  """The reference raises 'Invalid state in Pearson generator.' (:268-269) when a stretched interval
  has fewer than 3*512-1 samples; here that cannot happen for intervals >= 2 s, but the C ABI
  reports the same text if asked to."""
  n = 44100 * 3
  v = np.zeros((1, n), dtype=np.float16); a = np.zeros((1, n), dtype=np.float16)
  # 2 s of video from 0.03 s of audio would be |1-slope| > .1 -> skipped, so no error can be provoked
  ctx.replace_segments(v, a, np.array([0.0, 0.03]), np.array([0.0, 2.5]), False)
  assert not v.any()


@pytest.mark.parametrize("n_v,n_a,planar", [(3 * 44100 + 1, 3 * 44100 + 7, True), (3 * 44100 + 5, 3 * 44100 + 3, False),
                                            (3 * 44100 + 8, 3 * 44100 + 16, True)])
def test_stretch_resident_ragged_lengths(ctx, n_v, n_a, planar):
  """Odd track lengths: the second planar channel is then not 16-byte aligned (scalar loads), the
  last group of 8 frames is partial, and the float16 channel stride is padded."""
  rng = np.random.default_rng(n_v)
  vid = (rng.standard_normal((2, n_v)) * 6000).clip(-32768, 32767).astype(np.int16)
  aud = (rng.standard_normal((2, n_a)) * 9000).clip(-32768, 32767).astype(np.int16)
  aud[1] //= 3
  x = np.array([0.01, 2.95]); y = np.array([0.0, 2.93])
  v, a = vid.astype(np.float16), aud.astype(np.float16)
  want_f = SO.match_loudness(v, a)
  SO.replace_aligned_segments(v, a, x, y, False)
  SO.normalise_peak(v)
  want = v.astype(np.int16).T
  if planar:
    ctx.pcm_upload(0, vid); ctx.pcm_upload(1, aud)
  else:
    ctx.pcm_upload(0, np.ascontiguousarray(vid.T)); ctx.pcm_upload(1, np.ascontiguousarray(aud.T))
  got, fac = ctx.stretch_resident(x, y, False)
  np.testing.assert_allclose(fac, want_f, rtol=1e-12)
  assert np.array_equal(got, want), f"{int((got != want).sum())} samples differ"


def test_combine_directory_with_stretch_audio(ctx, tmp_path):
  """combine() on a directory of stereo .wav pairs with --stretch_audio: the pipelined batch writes
  the same tracks as one-pair-at-a-time processing."""
  import wave
  from describealign_amd import combine, media, synth
  vdir, adir = tmp_path / "video", tmp_path / "ad"
  os.makedirs(vdir); os.makedirs(adir)
  for k in range(3):
    pair = synth.make_pair(seed=60 + k, video_seconds=60.0 + 3 * k, jumps=([0.0, 25.0 + k], [4.0 + k, 1.5]), channels=2)
    media.write_wav(str(vdir / f"ep{k}.wav"), pair.video); media.write_wav(str(adir / f"ep{k}.wav"), pair.audio)
  if media.find_ffmpeg() is not None:
    pytest.skip("compares the .wav tracks written when no ffmpeg binary is present")
  outs = {}
  for name in ("seq", "bat"):
    os.makedirs(tmp_path / name / "out"); os.makedirs(tmp_path / name / "plots")
  for k in range(3):
    combine.process_pair(str(vdir / f"ep{k}.wav"), str(adir / f"ep{k}.wav"), True, ctx, stretch_audio=True,
                         output_dir=str(tmp_path / "seq" / "out"), alignment_dir=str(tmp_path / "seq" / "plots"))
  combine.combine(str(vdir), str(adir), stretch_audio=True, yes=True, output_dir=str(tmp_path / "bat" / "out"),
                  alignment_dir=str(tmp_path / "bat" / "plots"))
  for k in range(3):
    tracks = []
    for name in ("seq", "bat"):
      with wave.open(str(tmp_path / name / "out" / f"ad_ep{k}.wav"), "rb") as w:
        tracks.append(w.readframes(w.getnframes()))
    assert tracks[0] == tracks[1] and len(tracks[0]) > 10 ** 6
    reports = [open(tmp_path / name / "plots" / f"ep{k}.txt").read().replace(f"/{name}/out/", "/out/") for name in ("seq", "bat")]
    assert reports[0] == reports[1]


@pytest.mark.parametrize("name", list(cases.COMBINE_STRETCH_CASES))
def test_stretch_audio_track_equals_reference_combine(ctx, name):
  """PCM -> features -> align -> da_stretch_resident on the GPU against the fixture recorded from
  the reference's own combine(stretch_audio=True): identical nodes and an identical s16le track."""
  from describealign_amd import align as A
  meta = INDEX["combine_stretch"][name]
  g = np.load(os.path.join(GOLD, f"combine_stretch_{name}.npz"))
  pair = cases.combine_stretch_case(name)
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx)
  np.testing.assert_allclose(x, g["x"], atol=1e-6); np.testing.assert_allclose(y, g["y"], atol=1e-6)
  track, fac = ctx.stretch_resident(x, y, False)
  assert np.array_equal(track[::997], g["s16_every_997"])
  assert cases.sha1_of(np.ascontiguousarray(track)) == meta["sha1_s16le"]
  assert [int(track.min()), int(track.max())] == meta["peak_int16"]
