"""Pins oracle/dalign_oracle.py to fixtures recorded from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import json
import os

import numpy as np
import pytest

import cases
from oracle import dalign_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))


@pytest.mark.parametrize("name", cases.FEATURE_CLIPS)
def test_features_match_reference(name):
  g = np.load(os.path.join(GOLD, "features.npz"))
  pcm = cases.feature_clip(name)
  assert cases.sha1_of(pcm) == INDEX["features"][name]["sha1"], "synthetic generator drifted"
  feats = O.features(pcm)
  for k, f in enumerate(feats):
    r = g[f"{name}.f{k}"]
    assert f.shape == r.shape and f.dtype == r.dtype
    # float32 pipelines differ only by summation order
    np.testing.assert_allclose(f, r, rtol=2e-6, atol=1e-6)


@pytest.fixture(scope="module")
def a40():
  g = np.load(os.path.join(GOLD, "align_a40.npz"))
  vf = [g[f"vf{k}"] for k in range(5)]
  af = [g[f"af{k}"] for k in range(5)]
  return g, vf, af


def test_stage12_prep_candidates_matches(a40):
  g, vf, af = a40
  ms_v = [O.mean_sub(f) for f in vf]; ms_a = [O.mean_sub(f) for f in af]
  nv = [O.window_norm(m) for m in ms_v]; na = [O.window_norm(m) for m in ms_a]
  for j in range(5):
    np.testing.assert_allclose(ms_v[j], g[f"ms_v{j}"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(na[j], g[f"nrm_a{j}"], rtol=1e-12, atol=0)
  ci, cv = O.candidates(ms_v, nv, ms_a, na, vf[0], af[0])
  assert np.array_equal(ci, g["cand_i"]) and np.array_equal(cv, g["cand_v"])
  _, keep, qual = O.verify(ci, cv, ms_v, nv, ms_a, na)
  assert np.array_equal(ci[keep], g["m_i"]) and np.array_equal(cv[keep], g["m_v"])
  np.testing.assert_allclose(qual[keep], g["m_q"], rtol=1e-9)


def test_stage2_chain(a40):
  g, _, _ = a40
  idx = O.chain(g["m_i"], g["m_v"], g["m_q"])
  assert np.array_equal(g["m_i"][idx], g["p1_x"]) and np.array_equal(g["m_v"][idx], g["p1_y"])


def test_stage3_filter_scale_lp(a40):
  g, vf, af = a40
  x, y = g["p1_x"], g["p1_y"]
  ok = O.continuity_err(x, y) < 3
  x, y = x[ok], y[ok]
  a_s, v_s = O.scale_features(vf, af, x, y)
  np.testing.assert_allclose(a_s, g["a_scaled"], rtol=1e-13)
  np.testing.assert_allclose(v_s, g["v_scaled"], rtol=1e-13)
  fx, fy = O.compress_path(x, y)
  assert np.array_equal(fx, g["lp_x"]) and np.array_equal(fy, g["lp_y"])
  c, A, b, _ = O.build_lp(fx, fy)
  A = A.tocsc(); A.sort_indices()
  assert np.array_equal(c, g["lp_c"]) and np.array_equal(b, g["lp_b"])
  assert np.array_equal(A.data, g["lp_A_data"]) and np.array_equal(A.indices, g["lp_A_indices"])
  assert np.array_equal(A.indptr, g["lp_A_indptr"])
  lp = O.solve_lp(fx, fy)
  np.testing.assert_allclose(lp["sol"], g["lp_sol"], atol=1e-9)
  np.testing.assert_allclose(lp["slopes"], g["slopes"], atol=1e-12)
  assert abs(lp["median_slope"] - g["median_slope"]) < 1e-12


def test_stage4_clusters_points_dp(a40):
  g, vf, af = a40
  sp = g["smooth_path"]
  cl = O.line_clusters(sp[:, 0], sp[:, 1], g["slopes"])
  assert len(cl) == len(g["cl_offset"])
  for k, (cx, off, sl) in enumerate(cl):
    assert cx[0] == g["cl_x0"][k] and cx[-1] == g["cl_x1"][k]
    assert abs(off - g["cl_offset"][k]) < 1e-9 and abs(sl - g["cl_slope"][k]) < 1e-12
  pts, _ = O.extend_clusters(cl, g["a_scaled"], g["v_scaled"])
  flat = [(i, j, c, q) for i, p in enumerate(pts) for (j, c, q) in p]
  pi, pj, pc, pq = map(np.array, zip(*flat))
  assert np.array_equal(pi, g["pt_i"]) and np.array_equal(pc, g["pt_c"])
  np.testing.assert_allclose(pj, g["pt_j"], atol=1e-9)
  np.testing.assert_allclose(pq, g["pt_q"], atol=1e-6)
  path = O.second_dp(pts, len(cl), len(g["v_scaled"]))
  nx, ny, sim, path = O.finish(path, len(g["a_scaled"]), len(g["v_scaled"]), len(af[0]), len(vf[0]))
  assert path.shape == g["path2"].shape
  np.testing.assert_allclose(path[:, :3], g["path2"][:, :3], atol=1e-9)
  np.testing.assert_allclose(path[:, 3:], g["path2"][:, 3:], atol=1e-3)
  np.testing.assert_allclose(nx, g["x"], atol=1e-9); np.testing.assert_allclose(ny, g["y"], atol=1e-9)
  assert abs(sim - g["sim"]) < 1e-9


E2E = ["e180", "e180s", "rate2", "e600", "rateneg600", "j600s"]


@pytest.mark.parametrize("name", E2E)
def test_end_to_end_from_pcm(name):
  """Oracle features + align from PCM vs the reference's nodes (tolerance: 1 ms; the bar for
  the product is +-23 ms)."""
  g = np.load(os.path.join(GOLD, f"align_{name}.npz"))
  pair = cases.align_case(name)
  assert pair.sha1() == INDEX["align"][name]["sha1"], "synthetic generator drifted"
  vf, af = O.features(pair.video), O.features(pair.audio)
  x, y, sim, path, med = O.align(vf, af, vf[0], af[0])
  assert len(x) == len(g["x"])
  np.testing.assert_allclose(x, g["x"], atol=1e-3); np.testing.assert_allclose(y, g["y"], atol=1e-3)
  assert abs(sim - float(g["sim"])) < 0.05 and abs(med - float(g["med"])) < 1e-6


def test_mismatched_pair_raises():
  pair = cases.align_case("mismatch")
  assert pair.sha1() == INDEX["align"]["mismatch"]["sha1"]
  assert INDEX["align"]["mismatch"]["error"] == O.MISMATCH_MSG
  vf, af = O.features(pair.video), O.features(pair.audio)
  with pytest.raises(RuntimeError, match="Alignment failed"):
    O.align(vf, af, vf[0], af[0])


# --------------------------------------------------------------------------- --stretch_audio path

from oracle import stretch_oracle as SO  # noqa: E402


@pytest.mark.parametrize("name", list(cases.STRETCH_CASES))
def test_stretch_oracle_matches_reference(name):
  """replace_aligned_segments (describealign.py:230-416): float16 bit patterns of every replaced
  interval and every jump schedule, as recorded from the reference."""
  g = np.load(os.path.join(GOLD, f"stretch_{name}.npz"))
  meta = INDEX["stretch"][name]
  v, a, x, y = cases.stretch_case_f16(name)
  assert cases.sha1_of(v.view(np.uint16), a.view(np.uint16)) == meta["sha1_inputs"], "synthetic generator drifted"
  before = v.copy()
  plan = SO.segment_plan(x, y, False)
  sched = SO.replace_aligned_segments(v, a, x, y, False)
  assert len(sched) == meta["n_schedules"]
  for k, s in enumerate(sched):
    assert np.array_equal(s, g[f"sched{k}"]), f"jump schedule {k}"
  for k, (kind, x0, x1, y0, y1) in enumerate(plan):
    if k in meta["replaced_intervals"]:
      assert kind != "skip"
      assert np.array_equal(v[:, y0:y1].view(np.uint16), g[f"seg{k}"]), f"interval {k} ({kind})"
    else:
      assert np.array_equal(v[:, y0:y1].view(np.uint16), before[:, y0:y1].view(np.uint16))
  assert cases.sha1_of(v.view(np.uint16)) == meta["sha1_output"]
  # no_pitch_correction: every kept interval goes through the resampler
  v2 = before.copy()
  assert SO.replace_aligned_segments(v2, a, x, y, True) == []
  assert cases.sha1_of(v2.view(np.uint16)) == bytes(g["npc_sha1"]).hex()


def test_stretch_chunk_plan_covers_every_window_once():
  """correlation_chunks (describealign.py:253-270): the windows handed out by successive chunks
  are consecutive, start at 0 and reach n // 512."""
  for n in (1535, 2048, 29286, 29287, 26112 + 25088, 100000, 529200, 3 * 25088 + 29286, 3 * 25088 + 29287):
    nxt = 0
    for begin, end, lo, hi in SO.correlation_chunks(n):
      assert begin % SO.WINDOW == 0 and end <= n
      assert begin // SO.WINDOW + lo == nxt
      nxt = begin // SO.WINDOW + hi
      assert end - begin >= 3 * SO.WINDOW - 1
    assert nxt == n // SO.WINDOW, n


@pytest.mark.parametrize("name", list(cases.COMBINE_STRETCH_CASES))
def test_whole_stretch_audio_block_matches_reference_combine(name):
  """The reference's combine(stretch_audio=True) was run with only its file I/O replaced
  (make_golden.py gen_combine_stretch), so the loudness matching and peak normalisation that are
  written inline in combine() (describealign.py:1135-1153) and the int16 serialisation (:136) are
  pinned too: sha1 of the scaled float16 tracks, of the array handed to the writer, and of its
  int16 frames."""
  meta = INDEX["combine_stretch"][name]
  g = np.load(os.path.join(GOLD, f"combine_stretch_{name}.npz"))
  pair = cases.combine_stretch_case(name)
  assert pair.sha1() == meta["sha1_inputs"], "synthetic generator drifted"
  vf, af = O.features(pair.video), O.features(pair.audio)
  x, y, sim, path, med = O.align(vf, af, vf[0], af[0])
  np.testing.assert_allclose(x, g["x"], atol=1e-9); np.testing.assert_allclose(y, g["y"], atol=1e-9)
  v, a = pair.video.astype(np.float16), pair.audio.astype(np.float16)
  SO.match_loudness(v, a)
  assert cases.sha1_of(v.view(np.uint16)) == meta["sha1_scaled_video"]
  assert cases.sha1_of(a.view(np.uint16)) == meta["sha1_scaled_audio"]
  SO.replace_aligned_segments(v, a, x, y, False)
  SO.normalise_peak(v)
  assert np.array_equal(v[:, ::997].view(np.uint16), g["media_every_997"])
  assert cases.sha1_of(v.view(np.uint16)) == meta["sha1_media_f16"]
  s16 = v.astype(np.int16).T
  assert np.array_equal(s16[::997], g["s16_every_997"])
  assert cases.sha1_of(s16) == meta["sha1_s16le"]
  assert [int(s16.min()), int(s16.max())] == meta["peak_int16"]
