"""CPU-only tests: C-ABI surface, host-side align stages against the reference goldens,
report format, combine() helpers, and the world_size-2 gloo path of the batch sharding."""
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))


@pytest.fixture(scope="module")
def lib():
  from describealign_amd import _native
  if not os.path.exists(_native.LIB_PATH):
    _native.build()
  return _native.load()


def test_library_exports_every_declared_symbol(lib):
  header = open(os.path.join(ROOT, "include", "dalign.h")).read()
  header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
  declared = set(re.findall(r"\b(da_[a-z_0-9]+)\s*\(", header))
  from describealign_amd import _native
  assert declared == set(_native.EXPORTS), declared ^ set(_native.EXPORTS)
  for name in declared:
    assert getattr(lib, name) is not None
  assert lib.da_abi_version() == _native.ABI_VERSION


def test_chain_dp_host_only_matches_reference_and_oracle(lib):
  """da_chain is host C++ and accepts a NULL context, so it is exercised here without a GPU:
  against the reference's recorded pass-1 path and against the oracle on tie-heavy input."""
  from describealign_amd import _native
  from oracle import dalign_oracle as O
  g = np.load(os.path.join(GOLD, "align_a40.npz"))
  pi, pv = _native.chain_host(g["m_i"], g["m_v"], g["m_q"])
  assert np.array_equal(pi, g["p1_x"]) and np.array_equal(pv, g["p1_y"])
  rng = np.random.default_rng(11)
  for trial in range(3):
    n = 30000
    i = np.sort(rng.integers(0, 4000, n)); v = rng.integers(0, 900, n) * 4
    keys = np.unique(i.astype(np.int64) * 100000 + v)
    i, v = (keys // 100000).astype(np.int32), (keys % 100000).astype(np.int32)
    q = rng.choice([50.0, 50.0, 25.0, 12.5, 0.75], len(i))        # capped qualities: many equal sums
    idx = O.chain(i, v, q)
    pi, pv = _native.chain_host(i, v, q)
    assert np.array_equal(pi, i[idx]) and np.array_equal(pv, v[idx])
  with pytest.raises(RuntimeError, match="Alignment failed, are the input files mismatched"):
    _native.chain_host(np.arange(5, dtype=np.int32), np.arange(5, dtype=np.int32), np.ones(5), min_len=1050)
  # unsorted input is rejected, not mis-processed
  with pytest.raises(RuntimeError):
    _native.chain_host(np.array([3, 1], dtype=np.int32), np.array([0, 0], dtype=np.int32), np.ones(2))


def test_column_decomposition_of_the_chain_dp_equals_the_recurrence(lib, tmp_path):
  """The device chain DP (k_chain_columns) cuts the video ranks into columns that hand one record per
  audio row to the right, takes matches 64 at a time and updates its tree in two phases.
  tests/chain_col_model.cpp restates exactly that decomposition on the CPU; here it is compiled and
  checked against the host utility (plain Fenwick recurrence) on random instances with many equal sums,
  for column widths from 1 rank to wider than the input and for weight-balanced columns (the kernel's
  formula), rows wider than a window and more than one 256-row batch.  (The kernel itself is checked against the same utility in the GPU tests.)"""
  import ctypes as C
  from describealign_amd import _native
  so = str(tmp_path / "chain_col_model.so")
  subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(os.path.dirname(__file__), "chain_col_model.cpp")], check=True)
  model = C.CDLL(so)
  vp = C.c_void_p

  def run(i, v, q, w):
    n = len(i); pred = np.empty(n, np.int32); best = C.c_int64(-1)
    model.chain_col_model(vp(i.ctypes.data), vp(v.ctypes.data), vp(q.ctypes.data), C.c_int64(n), C.c_int(w), vp(pred.ctypes.data), C.byref(best))
    ids, p = [], best.value
    while p >= 0:
      ids.append(p); p = int(pred[p])
    ids = np.array(ids[::-1], np.int64)
    return i[ids], v[ids]

  rng = np.random.default_rng(21)
  for trial in range(36):
    n = int(rng.integers(1, 30000)); rows = int(rng.integers(1, 3000)); cols = int(rng.integers(1, 4000))
    # positive: columns of w ranks; negative: -w weight-balanced columns (what the kernel launches)
    w = int(rng.choice([1, 2, 3, 7, 64, 100, 256, 1000, 4096, -1, -2, -5, -17, -128, -1000]))
    key = np.unique(rng.integers(0, rows, n).astype(np.int64) << 32 | rng.integers(0, cols, n) * 4)
    i = (key >> 32).astype(np.int32); v = (key & 0xffffffff).astype(np.int32)
    q = rng.choice([50.0, 50.0, 12.5, 3.25, 0.75], len(i)) if trial % 3 else rng.uniform(0.001, 50, len(i))
    wi, wv = _native.chain_host(i, v, q)
    gi, gv = run(i, v, np.ascontiguousarray(q, dtype=np.float64), w)
    assert len(gi) == len(wi) and np.array_equal(gi, wi) and np.array_equal(gv, wv), (trial, n, rows, cols, w)


def test_no_cpu_fallback_context_fails_loudly():
  import torch
  if torch.cuda.is_available():
    pytest.skip("a GPU is visible")
  from describealign_amd import _native
  with pytest.raises(RuntimeError, match="no usable gfx950"):
    _native.Context(0)


def test_product_never_imports_the_oracle():
  pkg = os.path.join(ROOT, "describealign_amd")
  for fn in os.listdir(pkg):
    if fn.endswith(".py"):
      src = open(os.path.join(pkg, fn)).read()
      assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# ", ""), fn


@pytest.fixture(scope="module")
def a40():
  g = np.load(os.path.join(GOLD, "align_a40.npz"))
  return g, [g[f"vf{k}"] for k in range(5)], [g[f"af{k}"] for k in range(5)]


def test_host_pass1_and_lp_match_reference(a40):
  from describealign_amd import align as A
  g, vf, af = a40
  x, y = g["p1_x"], g["p1_y"]
  keep = A.continuity_error(x, y) < 3
  x, y = x[keep], y[keep]
  a, v = A.scale_feature_stacks(vf, af, x, y)
  np.testing.assert_allclose(a, g["a_scaled"], rtol=1e-13); np.testing.assert_allclose(v, g["v_scaled"], rtol=1e-13)
  fx, fy = A.compress_path(x, y)
  assert np.array_equal(fx, g["lp_x"]) and np.array_equal(fy, g["lp_y"])
  c, M, b, _ = A.build_trend_lp(fx, fy)
  assert np.array_equal(c, g["lp_c"]) and np.array_equal(b, g["lp_b"])
  assert np.array_equal(M.data, g["lp_A_data"]) and np.array_equal(M.indices, g["lp_A_indices"])
  assert np.array_equal(M.indptr, g["lp_A_indptr"]) and tuple(M.shape) == tuple(g["lp_A_shape"])
  lp = A.solve_trend_lp(fx, fy)
  np.testing.assert_allclose(lp["solution"], g["lp_sol"], atol=1e-9)
  np.testing.assert_allclose(lp["slopes"], g["slopes"], atol=1e-12)


def test_host_clusters_and_nodes_match_reference(a40):
  from describealign_amd import align as A
  g, vf, af = a40
  sp = g["smooth_path"]
  x0, x1, off, slo = A.cluster_lines(sp[:, 0], sp[:, 1], g["slopes"])
  assert np.array_equal(x0, g["cl_x0"]) and np.array_equal(x1, g["cl_x1"])
  np.testing.assert_allclose(off, g["cl_offset"], atol=1e-9); np.testing.assert_allclose(slo, g["cl_slope"], atol=1e-12)
  p = g["path2"].copy(); p[:, :2] *= 210.0
  nx, ny, sim = A.nodes_and_similarity(p, len(g["a_scaled"]), len(g["v_scaled"]), len(af[0]), len(vf[0]))
  np.testing.assert_allclose(nx, g["x"], atol=1e-9); np.testing.assert_allclose(ny, g["y"], atol=1e-9)
  assert abs(sim - float(g["sim"])) < 1e-9


def test_report_text_and_setts_match_reference(tmp_path):
  from describealign_amd import report
  meta = INDEX["align"]["e180"]
  g = np.load(os.path.join(GOLD, "align_e180.npz"))
  x, y, sim, med = g["x"], g["y"], float(g["sim"]), float(g["med"])
  assert report.encode_fit_as_ffmpeg_expr(x, y, y[0] - x[0]) == meta["setts"]
  want = meta["report_txt"].splitlines()
  got = report.report_lines(x, y, sim, med, False, False, "<ffmpeg command>")
  assert len(got) == len(want)
  for a, b in zip(got, want):
    if b.startswith("Script Hash:"):
      assert a.startswith("Script Hash:")
    else:
      assert a == b
  # plot + text files are written under the reference's names
  path = np.zeros((len(g["path20"]) * 20, 5)); path[::20] = g["path20"]
  report.plot_alignment(str(tmp_path / "e180"), path, x, y, sim, med, False, False, "<ffmpeg command>")
  assert (tmp_path / "e180.png").stat().st_size > 10000
  assert (tmp_path / "e180.txt").read_text().splitlines()[3] == want[3]


def test_file_pairing_and_sharding(tmp_path):
  from describealign_amd import combine as Cb
  from describealign_amd.distrib import shard_pairs
  for n in ("ep10.mp4", "ep2.mp4", "ep1.mp4", "notes.txt"):
    (tmp_path / n).write_bytes(b"x")
  files, alt = Cb.get_sorted_filenames(str(tmp_path), Cb.VIDEO_EXTENSIONS, Cb.AUDIO_EXTENSIONS)
  assert [os.path.basename(f) for f in files] == ["ep1.mp4", "ep2.mp4", "ep10.mp4"] and alt == [0, 0, 0]
  with pytest.raises(RuntimeError, match="No file or directory found"):
    Cb.get_sorted_filenames(str(tmp_path / "missing"), Cb.VIDEO_EXTENSIONS)
  shards = shard_pairs(32, 8)
  assert sorted(sum(shards, [])) == list(range(32)) and all(len(s) == 4 for s in shards)
  assert shard_pairs(3, 8)[3:] == [[]] * 5


def test_file_pairing_errors_and_alt_extensions(tmp_path):
  """The remaining branches of get_sorted_filenames (:94-121): list input, missing list entry,
  empty directory, no valid extension (the reference's five-line message), audio files accepted
  through the alternative extensions and flagged 1."""
  from describealign_amd import combine as Cb
  d = tmp_path / "in"; d.mkdir()
  with pytest.raises(RuntimeError, match="Empty input directory"):
    Cb.get_sorted_filenames(str(d), Cb.VIDEO_EXTENSIONS)
  for n in ("b2.mkv", "b10.mkv", "a.flac", "x.txt"):
    (d / n).write_bytes(b"x")
  files, alt = Cb.get_sorted_filenames(str(d), Cb.VIDEO_EXTENSIONS, Cb.AUDIO_EXTENSIONS)
  assert [os.path.basename(f) for f in files] == ["a.flac", "b2.mkv", "b10.mkv"] and alt == [1, 0, 0]
  files, alt = Cb.get_sorted_filenames([str(d / "b10.mkv"), str(d / "b2.mkv")], Cb.VIDEO_EXTENSIONS)
  assert [os.path.basename(f) for f in files] == ["b2.mkv", "b10.mkv"] and alt == [0, 0]
  with pytest.raises(RuntimeError, match="No file found at input path"):
    Cb.get_sorted_filenames([str(d / "nope.mkv")], Cb.VIDEO_EXTENSIONS)
  with pytest.raises(RuntimeError) as e:
    Cb.get_sorted_filenames(str(d / "x.txt"), Cb.VIDEO_EXTENSIONS)
  msg = str(e.value).split("\n")
  assert msg[0] == "No files with valid extensions found at input path:" and msg[1].strip().endswith("x.txt")
  assert msg[2] == "Did you accidentally put the audio filepath before the video filepath?"
  assert msg[5] == "valid extensions for this input are:" and len(msg) == 7


def test_combine_preamble_asks_what_the_reference_asks(tmp_path, monkeypatch, capsys):
  """combine()'s interactive preamble (describealign.py:1033-1062): an audio file on the video side and the listed pairing are
  each confirmed with Enter (skipped by yes=True); the pairing is listed by file name; unequal counts are the reference's error."""
  from describealign_amd import combine as Cb
  vids, auds = tmp_path / "v", tmp_path / "a"
  vids.mkdir(); auds.mkdir()
  for n in ("ep2.mkv", "ep1.flac"):
    (vids / n).write_bytes(b"x")
  for n in ("ep1.mp3", "ep2.mp3"):
    (auds / n).write_bytes(b"x")
  asked, ran = [], []
  monkeypatch.setattr("builtins.input", lambda prompt="": asked.append(prompt) or "")
  monkeypatch.setattr(Cb, "_worker", lambda device, idx, pairs, kwargs, prec, *a: ran.append([(os.path.basename(v), os.path.basename(a_), alt) for v, a_, alt in pairs]))
  out_dir, plot_dir = str(tmp_path / "out"), str(tmp_path / "plots")
  Cb.combine(str(vids), str(auds), output_dir=out_dir, alignment_dir=plot_dir)
  assert asked == ["If this was intended, press Enter to continue...", "If they are correct, press Enter to continue..."]
  assert ran == [[("ep1.flac", "ep1.mp3", 1), ("ep2.mkv", "ep2.mp3", 0)]]
  lines = capsys.readouterr().out.split("\n")
  a = lines.index("One or more audio files found in video input. Was this intentional?")
  assert lines[a - 1] == "" and lines[a + 1] == "If not, press ctrl+c to kill this script." and lines[a + 2] == ""
  b = lines.index("ep1.flac")
  assert lines[b:b + 6] == ["ep1.flac", "ep1.mp3", "", "ep2.mkv", "ep2.mp3", ""]
  assert lines[b + 6:b + 9] == ["Are the above input file pairings correct?", "If not, press ctrl+c to kill this script.", ""]
  assert lines[b + 9].startswith("Processing files with") and lines[-2].startswith("All files processed.")
  assert os.path.isdir(out_dir) and os.path.isdir(plot_dir)
  asked.clear()
  Cb.combine(str(vids), str(auds), yes=True, output_dir=out_dir, alignment_dir=plot_dir)
  assert asked == [] and "Was this intentional" not in capsys.readouterr().out
  (auds / "ep3.mp3").write_bytes(b"x")
  with pytest.raises(RuntimeError) as e:
    Cb.combine(str(vids), str(auds), yes=True, output_dir=out_dir, alignment_dir=plot_dir)
  assert str(e.value).split("\n") == ["Number of valid files in input paths are not the same.", "The video path has 2 files",
                                       "The audio path has 3 files"]


def test_key_frame_time_follows_the_reference():
  """get_closest_key_frame_time (:451-458): expectations recorded from the reference's function fed
  the same key-frame tables; the parser reads what `ffprobe -of json -show_frames` prints."""
  from describealign_amd import combine as Cb
  for kf, t, want in (([0.0, 2.5, 5.0, 7.5, 10.0], 6.1, 6.25), ([0.0, 2.5], 9.0, 5.75), ([], 3.0, 1.5),
                      ([4.0, 8.0], 1.0, 4.0), ([0.0, 5.0], 5.0, 5.0)):
    assert Cb.closest_key_frame_time(kf, t) == want
  canned = json.dumps({"frames": [{"pts_time": "0.000000"}, {"pts_time": "2.502500", "side_data_list": [{}]},
                                  {"pkt_pos": "1"}, {"pts_time": "5.005000"}],
                       "streams": [], "format": {}})
  got = Cb.parse_key_frame_times(canned)
  assert got.tolist() == [0.0, 2.5025, 5.005]
  assert Cb.closest_key_frame_time(got, 3.0) == (2.5025 + 5.005) / 2
  if Cb.media.find_ffprobe() is None:
    with pytest.raises(RuntimeError, match="no ffprobe binary"):
      Cb.get_key_frame_data("video.mp4", 3.0)


def test_media_native_reader_only_takes_what_ffmpeg_would_return_untouched(tmp_path):
  """float / extensible WAV headers, other channel counts and ragged raw files must not
  crash or be silently mis-decoded by the native reader."""
  import wave
  from describealign_amd import media
  st = np.stack([np.arange(-5, 5, dtype=np.int16), np.arange(10, 0, -1, dtype=np.int16) * 3])
  media.write_wav(str(tmp_path / "st.wav"), st)
  assert np.array_equal(media.parse_audio_from_file(str(tmp_path / "st.wav"), 2), st)
  mono = media.parse_audio_from_file(str(tmp_path / "st.wav"), 1)      # no ffmpeg here: (L + R + 1) >> 1
  if media.find_ffmpeg() is None:
    assert np.array_equal(mono[0], (st[0].astype(np.int32) + st[1] + 1) >> 1)
  (tmp_path / "bad.wav").write_bytes(b"RIFF\x00\x00\x00\x00WAVEjunk")
  (tmp_path / "odd.raw").write_bytes(b"\x01\x00\x02\x00\x03\x00")
  if media.find_ffmpeg() is None:
    for name, ch in (("bad.wav", 1), ("odd.raw", 2)):
      with pytest.raises(RuntimeError, match="cannot decode"):
        media.parse_audio_from_file(str(tmp_path / name), ch)
  assert media.parse_audio_from_file(str(tmp_path / "odd.raw"), 1).tolist() == [[1, 2, 3]]
  with wave.open(str(tmp_path / "r48.wav"), "wb") as w:
    w.setnchannels(1); w.setsampwidth(2); w.setframerate(48000); w.writeframes(b"\x00\x00" * 8)
  assert media._read_native(str(tmp_path / "r48.wav"), ".wav", 1) is None


def _install_fake_decoder(tmp_path, monkeypatch):
  sys.path.insert(0, os.path.join(ROOT, "tests", "doubles"))
  import fake_decoder
  bindir = tmp_path / "bin"; bindir.mkdir()
  fake_decoder.install(bindir)
  monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ.get("PATH", ""))


def test_decoder_pipe_is_read_in_place_and_equals_the_reference_array(tmp_path, monkeypatch):
  """parse_audio_from_file through a decoder process (describealign.py:149-157): the frames are read from the pipe
  with readinto() -- into one buffer when the length is known, else piece by piece -- and come back as the (C, N)
  view of the interleaved frames; same values as decoding the WAV natively; a failing decoder raises the reference's
  error.  The decoder here is a test double (tests/doubles/fake_decoder.py): this image has no ffmpeg."""
  from describealign_amd import media
  _install_fake_decoder(tmp_path, monkeypatch)
  assert media.find_ffmpeg() == str(tmp_path / "bin" / "ffmpeg")
  rng = np.random.default_rng(11)
  st = rng.integers(-32768, 32768, size=(2, 300001), dtype=np.int16)
  media.write_wav(str(tmp_path / "clip.wav"), st)
  os.rename(tmp_path / "clip.wav", tmp_path / "clip.mka")             # not a name the native reader takes: goes to the decoder
  monkeypatch.setattr(media, "PIECE_BYTES", 1 << 18)                  # several pieces, a ragged last one
  got = media.parse_audio_from_file(str(tmp_path / "clip.mka"), 2)
  assert got.shape == st.shape and got.dtype == np.int16 and np.array_equal(got, st)
  assert got.T.flags.c_contiguous                                     # the decoder's own layout, not a transposed copy
  mono = media.parse_audio_from_file(str(tmp_path / "clip.mka"), 1)
  assert np.array_equal(mono[0], (st[0].astype(np.int32) + st[1] + 1) >> 1)
  taken = []
  def alloc(shape):
    taken.append(shape); return np.empty(shape, dtype=np.int16)
  again = media.parse_audio_from_file(str(tmp_path / "clip.mka"), 2, alloc)
  assert taken == [(300001, 2)] and np.array_equal(again, st)
  # the byte stream itself, in caller-sized bites
  with media.PcmSource(str(tmp_path / "clip.mka"), 2) as src:
    buf = np.empty(100003, dtype=np.int16); parts = []
    while True:
      k = src.readinto(buf)
      parts.append(buf[:k // 2].copy())
      if k < buf.nbytes:
        break
  assert np.array_equal(np.concatenate(parts).reshape(-1, 2).T, st)
  os.rename(tmp_path / "clip.mka", tmp_path / "broken.mka")
  with pytest.raises(RuntimeError, match="FFmpeg error"):
    media.parse_audio_from_file(str(tmp_path / "broken.mka"), 2)


def test_key_frame_and_track_probes_run_against_the_ffprobe_double(tmp_path, monkeypatch):
  """get_key_frame_data / get_closest_key_frame_time (describealign.py:443-458) and is_first_video_track_ad (:460-462)
  executed as subprocesses against the ffprobe test double (key frames every 2.5 s inside the requested
  -read_intervals window; a file called *described* carries an audio-description disposition)."""
  from describealign_amd import combine
  _install_fake_decoder(tmp_path, monkeypatch)
  (tmp_path / "show.mkv").write_bytes(b"x"); (tmp_path / "described_show.mkv").write_bytes(b"x")
  times = combine.get_key_frame_data(str(tmp_path / "show.mkv"), 201.81)
  assert times[0] == 0.0 and np.allclose(np.diff(times), 2.5) and times[-1] <= 241.81 < times[-1] + 2.5      # window = time + 40
  assert combine.get_closest_key_frame_time(str(tmp_path / "show.mkv"), 201.81) == 201.25                    # between 200.0 and 202.5
  assert combine.get_closest_key_frame_time(str(tmp_path / "show.mkv"), 7.5) == 8.75                         # on a key frame: it counts as "earlier"
  assert combine.is_first_video_track_ad(str(tmp_path / "described_show.mkv")) is True
  assert combine.is_first_video_track_ad(str(tmp_path / "show.mkv")) is False


def test_float16_pcm_full_scale_does_not_wrap():
  """the reference's float16 array holds 32768.0 for samples 32760..32767; converting it
  back for the int16 kernel input must clamp, not wrap to -32768."""
  from describealign_amd import features
  pcm = np.array([[32767, 32760, 32759, -32768, 0, 2049, -2051]], dtype=np.int16)
  f16 = pcm.astype(np.float16)
  back = features.as_pcm_int16(f16)
  assert back.dtype == np.int16 and back[0, 0] == 32767 and back[0, 1] == 32767 and back[0, 2] == 32752 and back[0, 3] == -32768
  # the kernel's own int16 -> float16 rounding then reproduces the reference's array exactly
  assert np.array_equal(back.astype(np.float16), f16)


def test_wav_round_trip(tmp_path):
  from describealign_amd import media, synth
  pcm = synth.programme(9, 44100).astype(np.int16)[None, :]
  media.write_wav(str(tmp_path / "a.wav"), pcm)
  back = media.parse_audio_from_file(str(tmp_path / "a.wav"), 1)
  assert back.dtype == np.int16 and np.array_equal(back, pcm)


def test_synth_is_deterministic_and_chunk_invariant():
  from describealign_amd import synth
  a = synth.programme(5, 600000, 0)
  old = synth._CHUNK
  try:
    synth._CHUNK = 1 << 16
    b = synth.programme(5, 600000, 0)
  finally:
    synth._CHUNK = old
  assert np.array_equal(a, b)
  p = synth.make_pair(1, 20.0, jumps=([0.0, 10.0], [2.0, 1.0]))
  assert p.audio.shape[1] == p.video.shape[1] + 3 * 44100 and p.true_offset_at(12.0) == 3.0


def test_row_blocks_cover_exactly():
  from describealign_amd.distrib import row_blocks
  for n, w in ((10, 3), (7, 8), (0, 2), (1000003, 8)):
    b = row_blocks(n, w)
    assert len(b) == w and b[0][0] == 0 and b[-1][1] == n
    assert all(b[k][1] == b[k + 1][0] for k in range(w - 1))
    assert max(e - s for s, e in b) - min(e - s for s, e in b) <= 1


_GLOO_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
from describealign_amd import distrib
g = distrib.Group("gloo")
shards = distrib.shard_pairs(5, g.world)
mine = shards[g.rank]
g.barrier()
m = g.max_over_ranks(10.0 + g.rank)
s = g.sum_over_ranks(float(len(mine)))
assert m == 10.0 + g.world - 1, m
assert s == 5.0, s
# one failing rank must be seen by every rank (align_tiled aborts the gather everywhere instead of deadlocking)
assert g.all_ok(True) is True
assert g.all_ok(g.rank != g.world - 1) is False
# the one exchange step of the tiled single-pair mode: ragged per-rank match lists gathered on rank 0
import numpy as np
n = 3 + 4 * g.rank
mi = (np.arange(n) + 100 * g.rank).astype(np.int32); mv = (np.arange(n) * 4).astype(np.int32)
mq = np.linspace(0.5, 50.0, n) + g.rank
class FakeCtx:                      # what Group.gather_matches_to_root needs of a context on the gloo path
  _gathered = None
  def match_fetch(self, k):
    return mi[:k], mv[:k], mq[:k]
ctx = FakeCtx()
total = g.gather_matches_to_root(ctx, n)
if g.rank == 0:
  gi, gv, gq = ctx._gathered
  assert total == 3 + 7 and gi.dtype == np.int32 and gq.dtype == np.float64
  assert np.array_equal(gi[:3], np.arange(3)) and np.array_equal(gi[3:], np.arange(7) + 100)
  assert np.array_equal(gv[3:], np.arange(7) * 4) and np.array_equal(gq[3:], np.linspace(0.5, 50.0, 7) + 1)
else:
  assert total is None and ctx._gathered is None
# feature rows of a pair that only rank 0 holds (bench.py's tiled workload: rank 0 synthesises, everyone matches)
rows = [np.arange(7, dtype=np.float32) * 0.5, np.arange(3, dtype=np.float32) - 1.0, np.zeros(0, dtype=np.float32)] if g.rank == 0 else None
got, meta = g.broadcast_rows(rows, ([0.0, 2.5], [1.0, 3.0]) if g.rank == 0 else None)
assert len(got) == 3 and got[0].dtype == np.float32 and np.array_equal(got[0], np.arange(7) * 0.5) and np.array_equal(got[1], [-1.0, 0.0, 1.0]) and len(got[2]) == 0
assert meta == ([0.0, 2.5], [1.0, 3.0])
# rank 0 finishes the pair alone and broadcasts the result; failures are raised everywhere
res = (np.array([0.0, 1.5, 9.25]), np.array([0.5, 2.0, 9.0]), 87.5, np.arange(20.0).reshape(4, 5), 1.0001) if g.rank == 0 else None
x, y, sim, path, med = g.broadcast_result(res)
assert np.array_equal(x, [0.0, 1.5, 9.25]) and np.array_equal(y, [0.5, 2.0, 9.0]) and sim == 87.5 and med == 1.0001
assert path.shape == (4, 5) and path[3, 4] == 19.0
try:
  g.broadcast_result(None, "Alignment failed, are the input files mismatched?" if g.rank == 0 else None)
  raise SystemExit("no error raised")
except RuntimeError as e:
  assert "mismatched" in str(e)
# decisions every rank derives for itself (the row blocks of a tiled pair): rank 0's everywhere, a deviating rank fails all
assert g.agree_on([0, 5, 9]) == [0, 5, 9]
try:
  g.agree_on([0, 5 + g.rank, 9])
  raise SystemExit("no error raised")
except RuntimeError as e:
  assert "different values" in str(e)
# the RCCL branch itself (exact-size point-to-point transfers into library-owned memory), moved over gloo between
# two real ranks: the "device" arrays are host memory here, everything else is the code RCCL runs
import ctypes, torch
class P2PCtx:
  def __init__(self):
    self.k = self.q = None; self.committed = None
  def match_import_reserve(self, total):
    self.k = np.full(max(1, total), -1, dtype=np.int64); self.q = np.full(max(1, total), -1.0)
    return self.k.ctypes.data, self.q.ctypes.data
  def match_export_device(self, pk, pq, k):
    ctypes.memmove(pk, ((mi[:k].astype(np.int64) << 32) | mv[:k]).ctypes.data, 8 * k)
    ctypes.memmove(pq, np.ascontiguousarray(mq[:k]).ctypes.data, 8 * k)
  def match_import_commit(self, total):
    self.committed = total
pctx = P2PCtx()
def host_view(ptr, n, typestr, dtype, device):
  a = pctx.k if ptr == pctx.k.ctypes.data else pctx.q
  assert a.ctypes.data == ptr and len(a) == n
  return torch.from_numpy(a)
distrib._device_view = host_view
for n_here, counts in ((n, [3, 7]), (0 if g.rank == 1 else n, [3, 0])):
  total = g._gather_device(pctx, n_here, counts)
  if g.rank == 0:
    assert total == sum(counts) == pctx.committed
    assert np.array_equal(pctx.k[:3] >> 32, np.arange(3)) and np.array_equal(pctx.q[:3], mq[:3])
    if counts[1]:
      assert np.array_equal(pctx.k[3:] >> 32, np.arange(7) + 100) and np.array_equal(pctx.k[3:] & 0xffffffff, np.arange(7) * 4)
      assert np.array_equal(pctx.q[3:], np.linspace(0.5, 50.0, 7) + 1)
  else:
    assert total is None
# rank 0 cannot reserve: raised on BOTH ranks before a single transfer is posted (nobody blocks in a send)
class NoRoom(P2PCtx):
  def match_import_reserve(self, total):
    raise MemoryError("da_match_import_reserve: out of device memory")
try:
  g._gather_device(NoRoom(), n, [3, 7])
  raise SystemExit("no error raised")
except MemoryError:
  assert g.rank == 0
except RuntimeError as e:
  assert g.rank == 1 and "another rank" in str(e)
g.barrier()
g.close()
print("rank", g.rank, "ok", mine)
"""


class _RecordingDist:
  """Stand-in for torch.distributed that records the point-to-point operations Group._gather_device posts."""
  irecv, isend = "irecv", "isend"

  class ReduceOp:
    MIN = "min"

  class P2POp:
    def __init__(self, op, tensor, peer):
      self.op, self.tensor, self.peer = op, tensor, peer

  class _Req:
    def wait(self):
      pass

  def __init__(self):
    self.posted = []

  def all_reduce(self, t, op=None):
    pass

  def batch_isend_irecv(self, ops):
    self.posted.extend(ops)
    return [self._Req() for _ in ops]


@pytest.mark.parametrize("counts", [[5, 9], [4, 0, 7, 1, 0, 3, 2, 6], [0, 0, 5], [0, 0]])
def test_p2p_gather_posts_exact_slices_in_rank_order(counts, monkeypatch):
  """The RCCL gather of a tiled pair (distrib.Group._gather_device): on rank 0 one receive per non-empty rank into exactly
  that rank's slice of the reserved arrays, in rank order; every other rank sends exactly its list; an empty rank posts nothing."""
  import torch
  from describealign_amd import distrib
  world, offs = len(counts), distrib.gather_offsets(counts)
  assert offs[0] == 0 and offs[-1] == sum(counts) and all(offs[r + 1] - offs[r] == counts[r] for r in range(world))

  class Ctx:
    def __init__(self):
      self.calls = []
      self.k = np.zeros(max(1, offs[-1]), dtype=np.int64); self.q = np.zeros(max(1, offs[-1]))
    def match_import_reserve(self, total):
      self.calls.append(("reserve", total)); return self.k.ctypes.data, self.q.ctypes.data
    def match_export_device(self, pk, pq, n):
      self.calls.append(("export", pk, pq, n))
    def match_import_commit(self, total):
      self.calls.append(("commit", total))

  for rank in range(world):
    g = object.__new__(distrib.Group)
    g.rank, g.local_rank, g.world, g.backend, g.device = rank, rank, world, "nccl", torch.device("cpu")
    g.dist = _RecordingDist()
    ctx = Ctx()
    monkeypatch.setattr(distrib, "_device_view", lambda ptr, n, ts, dt, dev: torch.from_numpy(ctx.k if ptr == ctx.k.ctypes.data else ctx.q))
    got = g._gather_device(ctx, counts[rank], counts)
    ops = g.dist.posted
    if rank == 0:
      assert got == offs[-1] and ctx.calls[0] == ("reserve", offs[-1]) and ctx.calls[-1] == ("commit", offs[-1])
      senders = [r for r in range(1, world) if counts[r]]
      assert [o.peer for o in ops] == [r for r in senders for _ in (0, 1)] and all(o.op == "irecv" for o in ops)
      for o in ops:
        base = ctx.k if o.tensor.dtype == torch.int64 else ctx.q
        assert o.tensor.numel() == counts[o.peer] and o.tensor.data_ptr() == base.ctypes.data + 8 * offs[o.peer]
      assert [o.tensor.dtype for o in ops] == [torch.int64, torch.float64] * len(senders)
      exports = [c for c in ctx.calls if c[0] == "export"]
      assert exports == ([("export", ctx.k.ctypes.data, ctx.q.ctypes.data, counts[0])] if counts[0] else [])
    else:
      assert got is None and not any(c[0] in ("reserve", "commit") for c in ctx.calls)
      if counts[rank]:
        assert [(o.op, o.peer, o.tensor.numel(), o.tensor.dtype) for o in ops] == \
               [("isend", 0, counts[rank], torch.int64), ("isend", 0, counts[rank], torch.float64)]
        assert ctx.calls == [("export", ops[0].tensor.data_ptr(), ops[1].tensor.data_ptr(), counts[rank])]
      else:
        assert ops == [] and ctx.calls == []


def test_two_rank_gloo_sharding(tmp_path):
  """N>1 path of bench.py / combine(): ranks own disjoint pairs and only meet at barriers and
  the max-over-ranks reduction (gloo stands in for RCCL on CPU)."""
  script = tmp_path / "w.py"
  script.write_text(_GLOO_WORKER)
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
  procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
  outs = [p.communicate(timeout=180)[0] for p in procs]
  assert all(p.returncode == 0 for p in procs), outs
  assert "ok [0, 2, 4]" in outs[0] and "ok [1, 3]" in outs[1]


def test_lp_pricing_rule_does_not_change_the_fit():
  """align.solve_trend_lp can ask HiGHS' dual simplex for Dantzig pricing (an option; faster on an
  idle core, slower under the pipeline's contention); the optimum is the one the reference's call
  (steepest edge, the default here too) finds."""
  from describealign_amd import align as A
  rng = np.random.default_rng(5)
  for n, rate in ((400, 0.0), (900, 0.02)):
    x = np.sort(rng.choice(np.arange(2000, 2000 + 140 * n), n, replace=False)).astype(np.float64)
    y = x * (1 + rate) - 42000.0
    for j in np.sort(rng.choice(np.arange(20000, 140 * n - 20000), 4, replace=False)):
      y[x >= j] -= rng.integers(200, 1200)
    y += rng.integers(-1, 2, n) * 0.5 + (rng.random(n) < 0.02) * rng.integers(-40, 40, n)
    fast = A.solve_trend_lp(x, y, pricing="dantzig")
    ref = A.solve_trend_lp(x, y)
    assert np.abs(fast["solution"] - ref["solution"]).max() < 1e-7
    assert np.array_equal(np.round(fast["slopes"], 6), np.round(ref["slopes"], 6))
    assert abs(fast["median_slope"] - ref["median_slope"]) < 1e-12


def _trend_points(n, seed, rate=0.0, second_rate=None):
  rng = np.random.default_rng(seed)
  x = np.sort(rng.choice(np.arange(2000, 2000 + 140 * n), n, replace=False)).astype(np.float64)
  y = x * (1 + rate) - 42000.0
  if second_rate is not None:                      # the rate changes half-way through the file
    h = n // 2
    y[h:] = y[h] + (x[h:] - x[h]) * (1 + second_rate)
  for j in np.sort(rng.choice(np.arange(20000, 140 * n - 20000), 4, replace=False)):
    y[x >= j] -= rng.integers(200, 1200)
  y += rng.integers(-1, 2, n) * 0.5 + (rng.random(n) < 0.02) * rng.integers(-40, 40, n)
  return x, y


@pytest.mark.parametrize("n,seed,rate,second", [(500, 1, 0.0, None), (800, 2, 0.02, None), (700, 3, -0.04, None),
                                                 (600, 4, 0.0, 0.03), (900, 5, 0.01, -0.02)])
def test_lp_without_rate_terms_is_certified_or_falls_back(n, seed, rate, second):
  """align.solve_trend_lp first solves the LP without the rate_jump / rate_change columns and keeps
  that optimum only when the dual certificate proves it optimal for the reference's full LP; either
  way the fit equals the one the reference's call produces."""
  from describealign_amd import align as A
  x, y = _trend_points(n, seed, rate, second)
  c, Amat, b, bounds = A.build_trend_lp(x, y)
  reduced = A._solve_without_rate_terms(c, Amat, b, bounds, n, np.diff(x))
  full = A.solve_trend_lp(x, y, reduce=False)
  used = A.solve_trend_lp(x, y)
  rate_jump = full["solution"][8 * n - 4:10 * n - 6]
  if second is None:
    assert reduced is not None, "a constant rate difference is absorbed by median_slope: the certificate should hold"
    assert np.abs(rate_jump).max() < 1e-8
  else:
    assert reduced is None, "a rate that changes inside the file needs rate_jump: the certificate must fail"
    assert np.abs(rate_jump).max() > 1.0
  assert np.abs(used["solution"] - full["solution"]).max() < 1e-6
  assert np.array_equal(np.round(used["slopes"], 6), np.round(full["slopes"], 6))
  assert abs(used["median_slope"] - full["median_slope"]) < 1e-10
  # the certificate is a statement about dual feasibility: check it against the full problem's own duals
  if reduced is not None:
    import scipy.optimize
    ref = scipy.optimize.linprog(c, A_eq=Amat, b_eq=b, bounds=bounds, method="highs-ds")
    assert abs(ref.fun - float(c @ reduced)) < 1e-6 * max(1.0, abs(ref.fun))


def test_lp_retries_with_interior_point_on_status_4_as_the_reference_does(monkeypatch):
  """describealign.py:843-844: when the dual simplex stops with status 4 (numerical difficulties) the reference
  solves the same LP again with method='highs-ipm'.  HiGHS does not fail on its own on these LPs, so the first
  call is made to report status 4: the retry must be the interior-point method on the identical problem, its
  solution must be used (it is the same optimum to 1e-6 frames), and any other failure must raise the reference's error."""
  import scipy.optimize
  from describealign_amd import align as A
  x, y = _trend_points(600, 3, 0.01, None)
  want = A.solve_trend_lp(x, y, reduce=False)
  real = scipy.optimize.linprog
  calls = []

  def flaky(c, **kw):
    calls.append(kw.get("method"))
    fit = real(c, **kw)
    if len(calls) == 1:
      fit = scipy.optimize.OptimizeResult(x=None, success=False, status=4, message="forced: numerical difficulties")
    return fit

  monkeypatch.setattr(scipy.optimize, "linprog", flaky)
  got = A.solve_trend_lp(x, y, reduce=False)
  assert calls == ["highs-ds", "highs-ipm"]
  assert np.abs(got["solution"] - want["solution"]).max() < 1e-6
  assert abs(got["median_slope"] - want["median_slope"]) < 1e-9

  calls.clear()
  monkeypatch.setattr(scipy.optimize, "linprog",
                      lambda c, **kw: scipy.optimize.OptimizeResult(x=None, success=False, status=2, message="forced: infeasible"))
  with pytest.raises(RuntimeError, match="Smooth Alignment L1-Min Optimization Failed"):
    A.solve_trend_lp(x, y, reduce=False)


def test_stretch_audio_command_lines_follow_the_reference_options():
  """write_replaced_media_to_disk with a media array (describealign.py:468-487): the new stereo
  track is piped in as s16le; with a video it is muxed in front of the original streams and tagged
  as the described track, without one it is stored on its own."""
  from describealign_amd import combine
  with_video = combine._replaced_media_command("ffmpeg", "out/ad_show.mkv", "show.mkv")
  assert with_video[:11] == ["ffmpeg", "-f", "s16le", "-ac", "2", "-acodec", "pcm_s16le", "-ar", "44100", "-i", "pipe:"]
  joined = " ".join(with_video)
  for needle in ("-i show.mkv", "-acodec copy", "-vcodec copy", "-scodec copy", "-max_interleave_delta 0",
                 "-c:a:0 aac", "-disposition:a:0 default+visual_impaired+descriptions", "-metadata:s:a:0 title=AD",
                 "-disposition:a:1 original", "-metadata:s:a:1 title=original", "out/ad_show.mkv -y"):
    assert needle in joined, needle
  # a video whose first audio track already is an audio description (output of a previous run, :478-480)
  again = combine._replaced_media_command("ffmpeg", "out/ad_show.mkv", "show.mkv", first_track_is_ad=True)
  k = again.index("-disposition:a:1")
  assert again[k + 1] == "visual_impaired+descriptions" and "title=original" not in again
  probe = '{"streams": [{"index": 1, "codec_type": "audio", "disposition": {"default": 1, "descriptions": 1, "visual_impaired": 0}}, {"index": 2}]}'
  assert combine.parse_first_audio_track_is_ad(probe) is True
  assert combine.parse_first_audio_track_is_ad('{"streams": [{"disposition": {"descriptions": 0, "visual_impaired": 0}}]}') is False
  assert combine.parse_first_audio_track_is_ad('{"streams": []}') is False
  audio_only = combine._replaced_media_command("ffmpeg", "out/ad_show.wav", None)
  assert audio_only[-2:] == ["out/ad_show.wav", "-y"] and "-vcodec" not in audio_only


def test_cli_accepts_the_reference_flags():
  from describealign_amd import combine
  import argparse
  # parse only: no GPU is touched before combine() runs
  real = combine.combine
  seen = {}
  combine.combine = lambda *a, **k: seen.update(args=a, kwargs=k)
  try:
    combine.command_line_interface(["v.mp4", "a.mp3", "--stretch_audio", "--yes", "--prepend", "x_",
                                    "--no_pitch_correction", "--gpus", "2", "--precision", "bf16"])
  finally:
    combine.combine = real
  flat = list(seen["args"]) + list(seen["kwargs"].values())
  assert "v.mp4" in flat and "a.mp3" in flat and "x_" in flat and True in flat


def test_worker_count_follows_the_cpu_time_budget(monkeypatch, tmp_path):
  """The host stage is bound by CPU time: under a cgroup quota (the GPU box: cpu.max = 16 CPUs on a 2 x 64-core host) a rank
  gets a quarter more workers than its share of the quota; without a quota three per four physical cores, at most 64."""
  from describealign_amd import align as A
  primary = list(range(128)); secondary = list(range(128, 256))
  domain = {c: (c % 128) // 8 for c in range(256)}
  monkeypatch.setattr(A, "_cpu_topology", lambda cpus=None: (primary, secondary, domain))
  monkeypatch.setattr(A, "cpu_quota", lambda: 16.0)
  assert A.default_worker_count(1) == 20
  assert A.default_worker_count(8) == 3
  assert A.default_worker_count(2) == 10
  monkeypatch.setattr(A, "cpu_quota", lambda: None)
  assert A.default_worker_count(1) == 64
  assert A.default_worker_count(8) == 12
  monkeypatch.setattr(A, "cpu_quota", lambda: 512.0)          # a quota above the core count is no limit
  assert A.default_worker_count(1) == 64
  small = ([0, 1, 2, 3], [4, 5, 6, 7], {c: 0 for c in range(8)})
  monkeypatch.setattr(A, "_cpu_topology", lambda cpus=None: small)
  monkeypatch.setattr(A, "cpu_quota", lambda: None)
  assert A.default_worker_count(1) == 3 and A.default_worker_count(4) == 2


def test_cpu_quota_reads_the_cgroup_files(tmp_path):
  """cpu_quota(): cgroup v2 `cpu.max` (the GPU box: "1600000 100000" at the container's root), the tightest limit up the path,
  "max" = none; cgroup v1 quota / period; nothing readable = None.  And whatever layout the test host has: None or positive."""
  from describealign_amd import align as A
  q = A.cpu_quota()
  assert q is None or q > 0

  def tree(files):
    root = tmp_path / f"fs{len(list(tmp_path.iterdir()))}"
    for name, text in files.items():
      f = root / name
      f.parent.mkdir(parents=True, exist_ok=True)
      f.write_text(text)
    return str(root)

  assert A.cpu_quota(tree({"proc/self/cgroup": "0::/\n", "sys/fs/cgroup/cpu.max": "1600000 100000\n"})) == 16.0
  assert A.cpu_quota(tree({"proc/self/cgroup": "0::/\n", "sys/fs/cgroup/cpu.max": "max 100000\n"})) is None
  nested = tree({"proc/self/cgroup": "0::/pods/job7\n", "sys/fs/cgroup/cpu.max": "max 100000\n", "sys/fs/cgroup/pods/cpu.max": "3200000 100000\n",
                 "sys/fs/cgroup/pods/job7/cpu.max": "800000 100000\n"})
  assert A.cpu_quota(nested) == 8.0
  v1 = tree({"proc/self/cgroup": "3:cpu,cpuacct:/batch\n2:memory:/x\n", "sys/fs/cgroup/cpu,cpuacct/batch/cpu.cfs_quota_us": "250000\n",
             "sys/fs/cgroup/cpu,cpuacct/batch/cpu.cfs_period_us": "100000\n"})
  assert A.cpu_quota(v1) == 2.5
  unlimited_v1 = tree({"proc/self/cgroup": "3:cpu:/\n", "sys/fs/cgroup/cpu/cpu.cfs_quota_us": "-1\n", "sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"})
  assert A.cpu_quota(unlimited_v1) is None
  assert A.cpu_quota(tree({"proc/self/cgroup": "0::/\n"})) is None


def test_cpu_order_is_a_permutation_with_physical_cores_first():
  from describealign_amd import align as A
  cpus = sorted(os.sched_getaffinity(0))
  order = A.cpu_order()
  assert sorted(order) == cpus
  def siblings(c):
    try:
      txt = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
    except OSError:
      return {c}
    out = set()
    for part in txt.split(","):
      if "-" in part:
        a, b = part.split("-"); out.update(range(int(a), int(b) + 1))
      elif part:
        out.add(int(part))
    return out & set(cpus)
  n_cores = len({min(siblings(c)) for c in cpus})
  first = order[:n_cores]
  assert len({min(siblings(c)) for c in first}) == n_cores, "the first n_cores entries must be distinct physical cores"


def test_bench_refuses_more_rccl_ranks_than_gpus():
  """`python bench.py --gpus N` starts its N ranks itself; with the RCCL backend and fewer than N visible GPUs it must exit
  non-zero with a message and print no result line (never a silent one-GPU measurement labelled N)."""
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "DALIGN_DIST_BACKEND")}
  res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--workload", "cfg-small"],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
  assert res.returncode != 0 and "needs 64 visible GPUs" in res.stderr and not res.stdout.strip()


def test_result_buffers_are_handed_out_again_once_dropped(monkeypatch):
  """Context._recycled: the rows of a pair are views of one pooled buffer; when the caller has dropped them all the same
  memory must be handed out for the next pair (and never while a single row view is still alive)."""
  import ctypes as C
  from describealign_amd import _native

  def fake_pinned(shape, dtype=np.int16):          # what pinned_empty builds, without the device allocation
    count = int(np.prod(shape))
    buf = (C.c_uint8 * max(1, count * np.dtype(dtype).itemsize))()
    return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)

  monkeypatch.setattr(_native, "pinned_empty", fake_pinned)

  class Holder:
    _pool = []
  h = Holder()
  a = _native.Context._recycled(h, (5, 100), np.float32)
  addr = a.ctypes.data
  rows = [a[0, :90]] + [a[k, :89] for k in range(1, 5)]
  del a
  b = _native.Context._recycled(h, (5, 100), np.float32)          # the rows of the first buffer are still held
  assert b.ctypes.data != addr and len(h._pool) == 2
  keep = rows.pop()
  del rows
  c = _native.Context._recycled(h, (5, 100), np.float32)          # one row view left: still not free
  assert c.ctypes.data not in (addr, b.ctypes.data) and len(h._pool) == 3
  del keep
  d = _native.Context._recycled(h, (5, 100), np.float32)
  assert d.ctypes.data == addr and d.shape == (5, 100) and len(h._pool) == 3
  e = _native.Context._recycled(h, (5, 64), np.float32)           # every buffer is held: a new one
  assert e.shape == (5, 64) and len(h._pool) == 4
  # a directory's files all differ in length: a free buffer that is large enough (and not wastefully so) is reused
  e_addr = e.ctypes.data
  del e
  f = _native.Context._recycled(h, (5, 70), np.float32)
  assert f.ctypes.data == e_addr and f.shape == (5, 70) and f.flags.c_contiguous and len(h._pool) == 4
  g = _native.Context._recycled(h, (5, 400000), np.float32)       # far larger: its own buffer, rounded up for the next file of about this size
  assert len(h._pool) == 5 and h._pool[-1].size >= 2000000 and h._pool[-1].size % 65536 == 0
  g_addr = g.ctypes.data
  del g
  g2 = _native.Context._recycled(h, (5, 410000), np.float32)
  assert g2.ctypes.data == g_addr and len(h._pool) == 5
  # over the byte budget only as many FREE buffers go as it takes, never one that is in use
  del f, g2
  monkeypatch.setenv("DALIGN_ROW_POOL_BYTES", str(10 << 20))
  big = _native.Context._recycled(h, (5, 600000), np.float32)     # 12 MB: over budget whatever is dropped; both free buffers go
  held = {b.ctypes.data, c.ctypes.data, d.ctypes.data}
  assert {q.ctypes.data for q in h._pool} == held | {big.ctypes.data}


def test_command_lines_equal_what_the_reference_compiles():
  """tests/golden/commands.json holds the argv the REFERENCE's own functions compile (describealign.py:149-153 decode, :443-449
  and :460-462 probes, :464-515 both mux graphs) -- recorded by make_golden.py through a recording restatement of ffmpeg-python
  0.2.0's compile rules (tests/golden/ffmpeg_python_double.py; neither the package nor the binaries exist in this image).
  The product builds its command lines directly; they must be the same lists, and the "FFmpeg command:" text of the report
  (subprocess.list2cmdline of that list, back-slashes turned, :513-514) the same string."""
  import json
  from describealign_amd import combine, media
  recs = {r["name"]: r for r in json.load(open(os.path.join(GOLD, "commands.json")))}
  assert len(recs) == 13
  for ch in (1, 2):
    r = recs[f"decode_{ch}ch"]
    assert media._ffmpeg_decode_command("ffmpeg", r["args"]["media_file"], ch) == r["argv"]
  seen = []
  real_run = subprocess.run

  class Done:
    returncode, stderr = 0, b""
    def __init__(self, out): self.stdout = out
  def fake_run(argv, **kw):
    seen.append(list(argv))
    return Done(b'{"frames": [{"pts_time": "0.000000"}, {"pts_time": "10.010000"}, {"pkt_pts_time": "3"}], "streams": [{"disposition": {"descriptions": 0, "visual_impaired": 1}}]}')
  subprocess.run = fake_run
  try:
    for name in ("key_frames_None", "key_frames_12.5", "key_frames_300.0"):
      r = recs[name]
      times = combine.get_key_frame_data(r["args"]["video_file"], r["args"]["time"], ffprobe="ffprobe")
      assert seen.pop() == r["argv"] and [float(t) for t in times] == r["returned"]
    r = recs["first_track_is_ad"]
    assert combine.is_first_video_track_ad(r["args"]["video_file"], ffprobe="ffprobe") is r["returned"] and seen.pop() == r["argv"]
  finally:
    subprocess.run = real_run
  for name in ("mux_mp3_late", "mux_wav_early", "mux_flac_rate", "mux_m4a_zero"):
    r = recs[name]; a = r["args"]
    argv = combine._mux_command("ffmpeg", a["video_file"], a["audio_desc_file"], a["output_filename"], a["setts_cmd"], a["video_offset"],
                                a["after_start_key_frame"], a["median_slope"])
    assert argv == r["argv"], name
    assert subprocess.list2cmdline(argv).replace("\\", "/") == r["logged"]
  for name in ("stretch_video_first_original", "stretch_video_first_ad", "stretch_audio_only"):
    r = recs[name]; a = r["args"]
    argv = combine._replaced_media_command("ffmpeg", a["output_filename"], a["video_file"], a["first_track_is_ad"])
    assert argv == r["argv"], name
    assert subprocess.list2cmdline(argv).replace("\\", "/") == r["logged"]


def test_pipeline_threads_take_quiet_cores_away_from_the_lp_workers():
  """align.aux_core_order: a host of 2 x 16 cores with SMT (siblings c, c + 32), 8 cores per L3 domain, 6 LP workers: the
  workers' cores AND their siblings are set aside; the pipeline's own threads take physical cores, the domains without a
  worker first, one per domain before a second of any."""
  from describealign_amd import align as A
  cpus = list(range(64))
  primary, secondary = cpus[:32], cpus[32:]
  domain = {c: 8 * ((c % 32) // 8) for c in cpus}
  sib = {c: {c % 32, c % 32 + 32} for c in cpus}
  workers = [0, 8, 16, 1, 9, 2]                       # domains 0 / 8 / 16 carry 3 / 2 / 1 workers, domain 24 none
  rest, dealt = A.aux_core_order(cpus, workers, topology=(primary, secondary, domain), siblings=sib)
  assert rest == set(cpus) - set(workers) - {w + 32 for w in workers}
  assert all(c in primary and c in rest for c in dealt) and len(set(dealt)) == len(dealt) == 32 - 6
  assert [domain[c] for c in dealt[:4]] == [24, 16, 8, 0] and dealt[0] == 24 and dealt[3] == 3
  assert [domain[c] for c in dealt[4:8]] == [24, 16, 8, 0]
  # the live host: whatever its topology, the result is consistent with it
  live = sorted(os.sched_getaffinity(0))
  order = A.cpu_order(live)
  rest, dealt = A.aux_core_order(live, order[:2])
  assert not (set(order[:2]) & rest) and set(dealt) <= rest


def test_streamed_wav_with_an_unknown_data_size_is_read_as_far_as_the_file_goes(tmp_path):
  """A WAV written to a pipe (`ffmpeg -f wav -`) declares 0xFFFFFFFF data bytes: the native reader must size its buffer by the
  file, not by the header (a 4 GiB -- in batch mode page-locked -- allocation for a 2 s clip otherwise)."""
  import struct
  from describealign_amd import media
  rng = np.random.default_rng(4)
  pcm = rng.integers(-2000, 2000, size=(2, 88200), dtype=np.int16)
  path = str(tmp_path / "piped.wav")
  media.write_wav(path, pcm)
  raw = bytearray(open(path, "rb").read())
  at = raw.index(b"data") + 4
  raw[at:at + 4] = struct.pack("<I", 0xFFFFFFFF)
  raw[4:8] = struct.pack("<I", 0xFFFFFFFF)
  open(path, "wb").write(bytes(raw) + b"\x01")          # and a ragged tail byte
  span = media._wav_pcm_span(path, 2)
  assert span is not None and span[1] == pcm.size * 2
  got = media.parse_audio_from_file(path, 2)
  assert got.shape == pcm.shape and np.array_equal(np.asarray(got, dtype=np.int16), pcm)
