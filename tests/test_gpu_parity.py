"""Parity of the HIP path (through the C ABI) against the oracle and the reference goldens.
Needs a real MI355X: run with `pytest -m gpu`."""
import json
import os
import sys

import numpy as np
import pytest

import cases
from oracle import dalign_oracle as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
INDEX = json.load(open(os.path.join(GOLD, "index.json")))
HOP_S = 0.023          # north_star tolerance on node times: +-1 "STFT hop" (~23 ms)


@pytest.fixture(scope="module")
def native():
  from describealign_amd import _native
  return _native


@pytest.fixture(scope="module")
def ctx(native):
  c = native.Context(0, native.PREC_F32)
  yield c
  c.close()


@pytest.fixture(scope="module")
def ctx_bf16(native):
  c = native.Context(0, native.PREC_BF16)
  yield c
  c.close()


@pytest.fixture(scope="module")
def a40():
  g = np.load(os.path.join(GOLD, "align_a40.npz"))
  vf = [g[f"vf{k}"] for k in range(5)]
  af = [g[f"af{k}"] for k in range(5)]
  return g, vf, af


# ------------------------------------------------------------------------------------ features
@pytest.mark.parametrize("name", cases.FEATURE_CLIPS)
def test_features_vs_reference_golden(ctx, name):
  g = np.load(os.path.join(GOLD, "features.npz"))
  pcm = cases.feature_clip(name)
  rows = ctx.features(pcm)
  for k, f in enumerate(rows):
    r = g[f"{name}.f{k}"]
    assert f.shape == r.shape, (name, k, f.shape, r.shape)
    # float32 arithmetic in a different summation order; tolerance 2e-5 relative + 2e-5 abs
    np.testing.assert_allclose(f, r.astype(np.float32), rtol=2e-5, atol=2e-5, err_msg=f"{name} row {k}")


@pytest.mark.parametrize("layout", ["planar", "interleaved"])
def test_features_stereo_layouts_agree_with_oracle(ctx, layout):
  pcm = cases.feature_clip("stereo")
  want = O.features(pcm)
  arg = pcm if layout == "planar" else np.ascontiguousarray(pcm.T)
  rows = ctx.features(arg)
  for f, r in zip(rows, want):
    np.testing.assert_allclose(f, np.asarray(r, dtype=np.float32), rtol=2e-5, atol=2e-5)


def test_features_long_clip_vs_oracle(ctx):
  """A clip spanning many workgroup chunks (halo exchange between chunks), odd length."""
  from describealign_amd import synth
  pcm = synth.programme(77, 95 * 44100 + 12345).astype(np.int16)[None, :]
  rows = ctx.features(pcm)
  want = O.features(pcm)
  for k, (f, r) in enumerate(zip(rows, want)):
    assert f.shape == r.shape
    np.testing.assert_allclose(f, np.asarray(r, dtype=np.float32), rtol=2e-5, atol=2e-5, err_msg=f"row {k}")


def test_features_empty_and_tiny(ctx):
  rows = ctx.features(np.zeros((1, 0), dtype=np.int16))
  assert [len(r) for r in rows] == [0, 0, 0, 0, 0]
  rows = ctx.features(np.ones((1, 100), dtype=np.int16))
  assert [len(r) for r in rows] == [0, 0, 0, 0, 0]
  # 19 frames.  (Below 15 frames numpy's convolve(..., 'same') makes the reference return 15 band
  # values for fewer frames -- such clips cannot be aligned anyway and are not reproduced.)
  pcm = cases.feature_clip("short")[:, :4000]
  rows = ctx.features(pcm)
  want = O.features(pcm)
  for f, r in zip(rows, want):
    np.testing.assert_allclose(f, np.asarray(r, dtype=np.float32), rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------ matching
def _match_sets(i, v):
  return set(zip(i.tolist(), v.tolist()))


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_match_equals_reference_matches(ctx, ctx_bf16, a40, prec):
  g, vf, af = a40
  c = ctx if prec == "f32" else ctx_bf16
  mi, mv, mq = c.match(vf, af)
  # sorted by (i, v)
  key = mi.astype(np.int64) * (1 << 32) + mv
  assert np.all(np.diff(key) > 0)
  got, want = _match_sets(mi, mv), _match_sets(g["m_i"], g["m_v"])
  # integer/index work: the verified match set must be identical (verification is float64)
  assert got == want, (len(got - want), len(want - got))
  np.testing.assert_allclose(mq, g["m_q"], rtol=1e-9)
  st = c.stats()
  assert st["survivors"] > 0 and st["gemm_pairs"] > 0


def test_dense_mode_is_superset_and_matches_oracle_threshold(ctx, a40):
  g, vf, af = a40
  from describealign_amd import _native
  mi, mv, mq = ctx.match(vf, af, mode=_native.MATCH_DENSE)
  dense = _match_sets(mi, mv)
  assert _match_sets(g["m_i"], g["m_v"]) <= dense
  ms_v = [O.mean_sub(f) for f in vf]; ms_a = [O.mean_sub(f) for f in af]
  nv = [O.window_norm(m) for m in ms_v]; na = [O.window_norm(m) for m in ms_a]
  vr, ar = O.video_rows(vf[0]), O.audio_rows(af[0])
  ii = np.repeat(ar, len(vr)); vv = np.tile(vr, len(ar))
  _, keep, qual = O.verify(ii, vv, ms_v, nv, ms_a, na)
  assert dense == _match_sets(ii[keep], vv[keep])


@pytest.mark.parametrize("prec,tol", [("f32", 1e-3), ("bf16", 2e-2)])
def test_similarity_values_vs_fp64(ctx, ctx_bf16, native, a40, prec, tol):
  """north_star: similarity values within 1e-3 relative (fp32 GEMM).  Checked on the matrix cores'
  own accumulators: da_match_dump_tile returns, for whole 32 x 32 tiles, what the acceptance test of
  k_match_f32 / k_match_bf16 sees -- 1 - corr_j (f32) or 1 - guard - corr_j (bf16), formed from the very
  fragment streams the last launch read, with its MFMA sequence -- against the float64 correlation of the
  oracle.  bf16 inputs are a prefilter only (everything is re-verified in float64); their tolerance is
  2e-2 absolute."""
  g, vf, af = a40
  c = ctx if prec == "f32" else ctx_bf16
  c.match(vf, af)
  ms_v = [O.mean_sub(f) for f in vf]; ms_a = [O.mean_sub(f) for f in af]
  nv = [O.window_norm(m) for m in ms_v]; na = [O.window_norm(m) for m in ms_a]
  vr, ar = O.video_rows(vf[0]), O.audio_rows(af[0])
  n_vt, n_at = (len(vr) + 31) // 32, (len(ar) + 31) // 32
  # tiles that hold reference matches (high correlations) and random tiles (everything else), last tiles included
  pos_v = {int(v): k for k, v in enumerate(vr)}; pos_a = {int(i): k for k, i in enumerate(ar)}
  tiles = {(pos_v[int(v)] // 32, pos_a[int(i)] // 32) for i, v in list(zip(g["m_i"], g["m_v"]))[::97]}
  rng = np.random.default_rng(0)
  tiles |= {(int(rng.integers(n_vt)), int(rng.integers(n_at))) for _ in range(24)} | {(n_vt - 1, n_at - 1), (0, 0)}
  checked = 0; hi = 0
  for vt, at in sorted(tiles):
    acc, vfr, afr = c.match_dump_tile(vt, at)
    assert np.array_equal(vfr[vfr >= 0], vr[vt * 32:vt * 32 + 32]) and np.array_equal(afr[afr >= 0], ar[at * 32:at * 32 + 32])
    rows = np.flatnonzero(vfr >= 0); cols = np.flatnonzero(afr >= 0)
    ii = np.repeat(afr[cols], len(rows)); vv = np.tile(vfr[rows], len(cols))
    corr64, _, _ = O.verify(ii, vv, ms_v, nv, ms_a, na)                 # [pairs][3]
    corr64 = corr64.reshape(len(cols), len(rows), 3)
    for j in range(3):
      got = 1.0 - acc[j][np.ix_(rows, cols)].astype(np.float64)                           # [rows][cols]
      want = corr64[:, :, j].T
      if prec == "f32":
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-5, err_msg=f"tile {vt},{at} feature {j}")
      else:
        # the bf16 norm slot carries 1 - guard: the accumulator is 1 - guard - corr + rounding, and the
        # guard is a proven bound on that rounding -- so the prefilter never over-estimates 1 - corr (superset property)
        assert np.all(got >= want - 1e-6), f"tile {vt},{at} feature {j}: bf16 accumulator above the exact value"
        np.testing.assert_allclose(got - native.BF16_GUARD, want, rtol=0, atol=tol, err_msg=f"tile {vt},{at} feature {j}")
      hi += int(np.sum(want > 0.9))
    checked += len(rows) * len(cols)
  assert checked > 20000 and hi > 50           # tens of thousands of pairs, matches among them
  # the scalar side kernel (explicit pairs) agrees with the tile dump
  acc, vfr, afr = c.match_dump_tile(0, 0)
  side = c.match_corr(np.repeat(afr[:4], 4), np.tile(vfr[:4], 4))
  for j in range(3):
    tile = 1.0 - acc[j][:4, :4]
    np.testing.assert_allclose(side[:, j].reshape(4, 4).T, tile, atol=2e-3 if prec == "f32" else 3e-2)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_stage2_matches_at_config_size_vs_reference_sample(ctx, ctx_bf16, prec):
  """Stage 2 against the REFERENCE at configs[1] size (22-minute pair, 2.07e6 matches): the reference's verified
  matches of every 64th audio frame and its total were recorded by tests/golden/make_golden.py matches:e1320
  (line hook at describealign.py:674).  Here the pair goes PCM -> GPU features -> GEMM -> k_verify.  The GPU's float32
  feature rows differ from the reference's by summation order (2e-6 relative), which moves a hash digit
  (floor(8 x + 3.5), :639-643) or a correlation across its threshold for about one match in 10^4: asserted is a
  symmetric difference below 2e-3 of the sampled set, the total within 2e-3, and equal qualities (1e-3) on the rest."""
  path = os.path.join(GOLD, "matches_e1320.npz")
  if not os.path.exists(path):
    pytest.skip("fixture matches_e1320.npz not recorded")
  g = np.load(path)
  pair = cases.align_case("e1320")
  assert pair.sha1() == INDEX["matches"]["e1320"]["sha1"]
  c = ctx if prec == "f32" else ctx_bf16
  vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
  mi, mv, mq = c.match(vf, af)
  total, every = int(g["total"]), int(g["every"])
  assert abs(len(mi) - total) <= 2e-3 * total, (len(mi), total)
  sel = (mi % every) == 0
  got = (mi[sel].astype(np.int64) << 32) | mv[sel]
  want = g["keys"]
  both = np.intersect1d(got, want)
  assert len(got) + len(want) - 2 * len(both) <= 2e-3 * len(want), (len(got), len(want), len(both))
  gq = mq[sel][np.searchsorted(got, both)]; wq = g["quals"][np.searchsorted(want, both)]
  assert np.mean(np.abs(gq - wq) <= 1e-3 * np.abs(wq)) >= 0.999


def test_chain_equals_reference_path(ctx, a40):
  g, _, _ = a40
  pi, pv = ctx.chain(g["m_i"], g["m_v"], g["m_q"])
  assert np.array_equal(pi, g["p1_x"]) and np.array_equal(pv, g["p1_y"])
  # and against the oracle on a tie-heavy random instance (capped qualities produce equal sums)
  rng = np.random.default_rng(5)
  n = 20000
  i = np.sort(rng.integers(0, 3000, n)); v = rng.integers(0, 800, n) * 4
  keys = np.unique(i.astype(np.int64) * 100000 + v)
  i, v = (keys // 100000).astype(np.int32), (keys % 100000).astype(np.int32)
  q = rng.choice([50.0, 50.0, 12.5, 3.25], len(i))
  idx = O.chain(i, v, q)
  pi, pv = ctx.chain(i, v, q)
  assert np.array_equal(pi, i[idx]) and np.array_equal(pv, v[idx])


def _random_chain_instance(rng, n, rows, cols, step=4, quals=(50.0, 50.0, 12.5, 3.25, 0.75)):
  i = np.sort(rng.integers(0, rows, n)); v = rng.integers(0, cols, n) * step
  keys = np.unique(i.astype(np.int64) * (1 << 32) + v)
  i, v = (keys >> 32).astype(np.int32), (keys & 0xffffffff).astype(np.int32)
  return i, v, rng.choice(quals, len(i))


@pytest.mark.parametrize("kernel", ["columns", "columns:1", "columns:37", "columns:700", "rows:1", "rows:4"])
@pytest.mark.parametrize("shape", ["wide_rows", "many_ranks", "sparse"])
def test_chain_kernel_equals_host_utility(ctx, native, shape, kernel, monkeypatch):
  """The device DP (da_chain with a context) against the host utility (NULL context) on instances
  that exercise what the goldens do not: rows with more than 64 / 256 points (several windows per
  row and column), more than 2^19 distinct video ranks and rows of one or two points.  Qualities are
  drawn from a few values, so equal sums abound.  The column pipeline runs with its own choice of
  columns and with 1, 37 and 700 forced (one workgroup; windows that straddle rows; far more columns
  than matches per row); the round-2 one-workgroup kernels (one / four wavefronts) stay as a cross-check."""
  kind, _, arg = kernel.partition(":")
  monkeypatch.setenv("DALIGN_CHAIN_KERNEL", kind)
  if kind == "rows": monkeypatch.setenv("DALIGN_CHAIN_WAVES", arg)
  elif arg: monkeypatch.setenv("DALIGN_CHAIN_COLS", arg)
  rng = np.random.default_rng({"wide_rows": 1, "many_ranks": 2, "sparse": 3}[shape])
  if shape == "wide_rows":
    i, v, q = _random_chain_instance(rng, 120000, 300, 5000)             # ~400 points per row: more than one 256-match super-step
  elif shape == "many_ranks":
    i, v, q = _random_chain_instance(rng, 1500000, 40000, 1200000, step=1)
  else:
    i, v, q = _random_chain_instance(rng, 20000, 15000, 30000)
  want_i, want_v = native.chain_host(i, v, q)
  got_i, got_v = ctx.chain(i, v, q)
  assert len(got_i) == len(want_i) and np.array_equal(got_i, want_i) and np.array_equal(got_v, want_v)
  assert ctx.stats()["chain_ms"] > 0


def test_chain_tiny_and_degenerate_instances(ctx, native):
  """Corner shapes of the column pipeline: one match; one row; one video frame; a diagonal where every
  match chains from the previous one; equal qualities everywhere (every comparison a tie); row counts
  around the 256-row batch and match counts around the 64-match window."""
  def check(i, v, q):
    i = np.asarray(i, np.int32); v = np.asarray(v, np.int32); q = np.asarray(q, np.float64)
    wi, wv = native.chain_host(i, v, q)
    gi, gv = ctx.chain(i, v, q)
    assert np.array_equal(gi, wi) and np.array_equal(gv, wv), (len(i), i[:8], v[:8])
  check([5], [7], [1.5])
  check([0, 0, 0], [0, 4, 8], [1.0, 1.0, 1.0])                               # one row: the whole row chains
  check([0, 1, 2, 3], [8, 8, 8, 8], [2.0, 2.0, 2.0, 2.0])                      # one video frame
  check([0, 1, 2, 3], [12, 8, 4, 0], [5.0, 1.0, 1.0, 1.0])                     # anti-diagonal: no chain longer than one
  for n in (63, 64, 65, 255, 256, 257, 511, 513, 1000):
    k = np.arange(n)
    check(k, 4 * k, np.full(n, 50.0))                                         # diagonal, all ties
    check(k, 4 * (k // 3), np.full(n, 12.5))                                  # three rows per video frame
    check(k // 5, 4 * (k % 5) + 20 * (k // 5), np.full(n, 3.25))              # five matches per row
  rng = np.random.default_rng(99)
  for rows in (1, 2, 255, 256, 257, 600):
    i, v, q = _random_chain_instance(rng, 4000, rows, 50, quals=(50.0,))
    check(i, v, q)


def test_chain_with_few_matches_per_video_frame_sizes_its_columns_for_the_lds_tree(ctx, native):
  """Many video frames, few matches each: the column count comes from the LDS limit on a column's width (a column may hold
  twice the average number of frames, + 2), not from the match count.  81 884 frames -- found by tests/gpu_stress_chain.py --
  sat exactly where the first formula for that minimum was one column short and the launch was refused."""
  rng = np.random.default_rng(7)
  for frames in (81884, 8189 * 2, 8190 * 3 + 1, 4094 * 21, 4094 * 21 + 1):
    n = 113005
    v = np.concatenate([np.arange(frames), rng.integers(0, frames, n - frames)]).astype(np.int64)
    i = rng.integers(0, 2337, n).astype(np.int64)
    key = np.unique((i << 32) | v)
    i = (key >> 32).astype(np.int32); v = (key & 0xffffffff).astype(np.int32)
    q = rng.choice([50.0, 12.5, 3.25, 0.75], len(i))
    wi, wv = native.chain_host(i, v, q)
    gi, gv = ctx.chain(i, v, q)
    assert np.array_equal(gi, wi) and np.array_equal(gv, wv), frames


def test_chain_handover_under_concurrent_gemm_load(ctx, native):
  """The columns of the chain DP hand their per-row records to each other through global memory while
  OTHER kernels keep every CU busy (in the batch pipeline: the similarity GEMM of the next pair).  Uneven
  load is where a stale read of a hand-over would show: the DP of one 10-minute match list is repeated
  while a second context runs f32 GEMMs back to back on another stream, with few and with many columns;
  every path must equal the host utility's."""
  import threading
  from describealign_amd import synth
  pair = synth.make_pair(23, 600.0, n_jumps=5, first_gap=60.0)
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  mi, mv, mq = ctx.match(vf, af)
  want = native.chain_host(mi, mv, mq)
  load = native.Context(0, native.PREC_F32)
  lvf = load.features(pair.video, 0); laf = load.features(pair.audio, 1)
  stop = threading.Event()
  def hammer():
    while not stop.is_set():
      load.match_begin(lvf, laf); load.match_finish()
  th = threading.Thread(target=hammer); th.start()
  try:
    for rep in range(24):
      os.environ["DALIGN_CHAIN_COLS"] = ("16", "64", "200", "800")[rep % 4]
      gi, gv = ctx.chain(mi, mv, mq)
      assert np.array_equal(gi, want[0]) and np.array_equal(gv, want[1]), rep
  finally:
    os.environ.pop("DALIGN_CHAIN_COLS", None)
    stop.set(); th.join(); load.close()


def test_chain_resident_equals_chain_of_fetched_matches(ctx, native):
  """da_chain_resident works on the match list left on the device by da_match; the same list copied
  out and run through the host utility gives the same path.  Two DPs may be in flight at once."""
  pairs = [cases.align_case("e180"), cases.align_case("a40")]
  want, tickets = [], []
  for p in pairs:
    vf = ctx.features(p.video, 0); af = ctx.features(p.audio, 1)
    mi, mv, mq = ctx.match(vf, af)
    want.append(native.chain_host(mi, mv, mq))
    tickets.append(ctx.chain_begin())                 # the next match starts while this DP runs
  for t, (wi, wv) in zip(tickets, want):
    gi, gv = ctx.chain_finish(t)
    assert np.array_equal(gi, wi) and np.array_equal(gv, wv) and len(gi) > 1000
  with pytest.raises(RuntimeError, match="unknown ticket"):
    ctx.chain_finish(tickets[0])


_FALLBACK_WORKER = r"""
import os, sys, json
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import cases
from describealign_amd import _native as native
c = native.Context(0, native.PREC_F32)
pair = cases.align_case("e180")
vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
mi, mv, mq = c.match(vf, af)
want = c.chain_resident()
cols = int(c.stats()["chain_columns"])
rows = len(np.unique(mi))
os.environ["DALIGN_CHAIN_HANDOVER_LIMIT"] = str(24 * (rows + 256) * 2)      # room for two columns
c.match_begin(vf, af); c.match_finish()
got = c.chain_resident()
print(json.dumps(dict(cols=cols, cols_limited=int(c.stats()["chain_columns"]),
                      same=bool(np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])))))
c.close()
"""


def test_chain_falls_back_to_fewer_columns_when_the_handover_buffer_does_not_fit(tmp_path):
  """The column DP sizes its hand-over records (24 B x rows x columns) by what is free on the device; when the allocation fails
  all the same it halves the column count until it fits.  Same path, fewer columns.  The failing allocation is provoked through
  a hook that only the diagnostic library carries (libdalign_dbg.so, -DDA_TEST_HOOKS): the shipped library reads no such
  variable -- checked here as well."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  script = tmp_path / "w.py"; script.write_text(_FALLBACK_WORKER)
  outs = {}
  for name in ("libdalign_dbg.so", "libdalign.so"):
    env = dict(os.environ, DALIGN_LIB=os.path.join(root, "describealign_amd", name))
    res = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stderr[-2000:]
    outs[name] = json.loads(res.stdout.strip().splitlines()[-1])
  d = outs["libdalign_dbg.so"]
  assert d["cols"] >= 4 and 1 <= d["cols_limited"] <= 2 and d["same"]
  s = outs["libdalign.so"]
  assert s["cols_limited"] == s["cols"] and s["same"]                    # the product ignores the variable


def test_chain_begin_refuses_the_next_pairs_row_list(ctx):
  """da_chain_begin ranks the resident matches with the video row list of the match that produced them; the
  next da_match_begin overwrites that list.  Calling chain_begin for pair k after match_begin of pair k+1 used to
  pair k's matches with k+1's rows; now it is a state error, and the documented order keeps working."""
  p1, p2 = cases.align_case("a40"), cases.align_case("e180")
  vf1 = ctx.features(p1.video, 0); af1 = ctx.features(p1.audio, 1)
  ctx.match_begin(vf1, af1); ctx.match_finish()
  vf2 = ctx.features(p2.video, 0); af2 = ctx.features(p2.audio, 1)
  ctx.match_begin(vf2, af2)                                  # pair 2's row lists replace pair 1's
  with pytest.raises(RuntimeError, match="da_match_begin has been called since"):
    ctx.chain_begin()
  ctx.match_finish()
  gi, gv = ctx.chain_resident()                              # pair 2 in the documented order: fine
  assert len(gi) > 1000


def test_chain_rejects_nonpositive_quality(ctx):
  i = np.arange(10, dtype=np.int32); v = np.arange(10, dtype=np.int32); q = np.ones(10); q[4] = 0.0
  with pytest.raises(RuntimeError, match="positive"):
    ctx.chain(i, v, q)


def test_chain_rejects_nan_and_infinite_quality_on_the_column_path(ctx):
  """The column kernel orders sums by their bit patterns (ds_max_u64): a NaN or an infinity must be refused by
  the validation kernel, which is compiled with -ffinite-math-only and therefore tests the bit pattern."""
  n = 5000
  i = np.arange(n, dtype=np.int32); v = np.arange(n, dtype=np.int32)
  for bad in (np.nan, np.inf, -1.0, -0.0, 1e301):
    q = np.ones(n); q[n // 2] = bad
    with pytest.raises(RuntimeError, match="positive"):
      ctx.chain(i, v, q)
  q = np.full(n, 5e-324)                                    # the smallest positive double is a legal quality
  gi, gv = ctx.chain(i, v, q)
  assert len(gi) == n


def test_imported_match_with_an_unlisted_video_frame_is_refused_cleanly(ctx):
  """da_match_import_device with a key whose video frame is not one of the matched rows: the rank map has no
  rank for it.  The column path used to turn rank 0 into column 65535 and write far past col_start[]; now the
  match is clamped to rank 1, the DP runs inside its arrays and the host reports DA_ERR_ARG.  The resident
  buffers of the context stay intact: the same list without the bad key gives the right path afterwards."""
  import torch
  pair = cases.align_case("e180")
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  mi, mv, mq = ctx.match(vf, af)
  n = len(mi)
  want = ctx.chain(mi, mv, mq)
  keys = (mi.astype(np.int64) << 32) | mv
  listed = set(O.video_rows(vf[0]).tolist())
  k = n // 2
  bad_v = next(x for x in range(int(mv[k]) + 1, int(mv[k]) + 8) if x not in listed)      # rows are every 4th non-quiet frame
  bad = keys.copy(); bad[k] = (int(mi[k]) << 32) | bad_v
  bad = np.sort(bad)
  for arr, ok in ((bad, False), (keys, True)):
    ctx.match(vf, af)                                       # the row list that ranks the imported matches
    tk = torch.from_numpy(arr).cuda(); tq = torch.from_numpy(mq).cuda()
    ctx.match_import_device(tk.data_ptr(), tq.data_ptr(), n)
    if ok:
      gi, gv = ctx.chain_resident()
      assert np.array_equal(gi, want[0]) and np.array_equal(gv, want[1])
    else:
      with pytest.raises(RuntimeError, match="outside the matched rows"):
        ctx.chain_resident()


def test_chain_mismatch_error(ctx):
  i = np.arange(10, dtype=np.int32); v = np.arange(10, dtype=np.int32); q = np.ones(10)
  with pytest.raises(RuntimeError, match="Alignment failed, are the input files mismatched"):
    ctx.chain(i, v, q, min_len=1050)


def test_refine_equals_reference_path(ctx, a40):
  g, vf, af = a40
  path, npts = ctx.refine(g["a_scaled"], g["v_scaled"], g["cl_x0"], g["cl_x1"], g["cl_offset"], g["cl_slope"])
  assert npts == len(g["pt_i"])
  want = g["path2"].copy(); want[:, :2] *= 210.0
  assert path.shape == want.shape
  np.testing.assert_allclose(path[:, :3], want[:, :3], atol=1e-6)
  np.testing.assert_allclose(path[:, 3:], want[:, 3:], atol=1e-3)


# ------------------------------------------------------------------------------------ end to end
def test_align_from_reference_features(ctx, a40):
  from describealign_amd import align as A
  g, vf, af = a40
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx)
  np.testing.assert_allclose(x, g["x"], atol=1e-6); np.testing.assert_allclose(y, g["y"], atol=1e-6)
  assert abs(sim - float(g["sim"])) < 1e-6 and abs(med - float(g["med"])) < 1e-9
  assert path.shape == g["path2"].shape


@pytest.mark.parametrize("name,prec", [("e180", "f32"), ("e180", "bf16"), ("e180s", "f32"), ("rate2", "f32"), ("rateneg600", "bf16"), ("j600s", "f32"),
                                       ("e600", "bf16"), ("e1320", "f32"), ("e1800", "bf16"), ("rate1800", "bf16"), ("j1800", "f32"), ("e3600", "bf16"), ("e7200s", "bf16")])
def test_end_to_end_from_pcm(ctx, ctx_bf16, name, prec):
  """PCM -> features -> align on the GPU vs the reference's recorded nodes: every node time
  within +-23 ms (north_star), similarity within 0.5 points.  e1320 is the configs[1] stand-in, e1800 is seed 0 of
  configs[3]'s batch of 32 half-hour pairs (the bench's `finite_batch_cfg3` runs seeds 0..31), rate1800 the same size with a 0.3 %
  rate difference between the files (median slope 1.003: the LP's rate terms at work), e7200s IS
  bench.py's configs[2] pair (2 h stereo, bf16 prefilter): its fixture took the reference ~40 minutes."""
  if name not in INDEX["align"]:
    pytest.skip(f"fixture align_{name}.npz not recorded")
  from describealign_amd import align as A
  c = ctx if prec == "f32" else ctx_bf16
  g = np.load(os.path.join(GOLD, f"align_{name}.npz"))
  pair = cases.align_case(name)
  assert pair.sha1() == INDEX["align"][name]["sha1"]
  vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=c)
  assert len(x) == len(g["x"])
  assert np.max(np.abs(x - g["x"])) < HOP_S and np.max(np.abs(y - g["y"])) < HOP_S
  assert abs(sim - float(g["sim"])) < 0.5 and abs(med - float(g["med"])) < 1e-4
  # the pass-2 path itself against the reference's (every 20th row was recorded, which is also what the reference
  # plots, :179).  The path is computed from THIS build's float32 feature rows, which differ from the reference's by
  # summation order (2e-6 relative).  What that leaves: the sub-frame refinement of a line's offset (:916-930) moves by up
  # to ~3e-3 frames (observed 1.3e-5 s at 1 h, 8e-5 s at 22 min), a handful of exact ties of the second DP may fall the
  # other way (observed: 2 rows of 255 145 on the 22-minute pair, 0 elsewhere), and a short cluster may be kept on one
  # side only, which renumbers the clusters after it (22-minute pair).  Asserted: the same number of rows to within 4;
  # at least 99.9 % of the recorded rows found again at the same audio frame with the video position within 2e-4 s
  # (0.04 frames; the north_star tolerance is 23 ms; j1800: see below); the path visits the same NUMBER of distinct clusters to within 1.
  # Qualities are -log10(1e-4 + |a - v|) of those rows, gated by clipped energy terms (:931-936): where the two sides
  # nearly agree a 2e-6 difference is a few per cent of |a - v|, and next to a gate's edge it switches part of the term
  # (observed: 0.5 % of the rows off by more than 0.1, at most 1.3); the running sum collects these as a random walk.
  path = np.asarray(path)
  want20 = g["path20"]
  assert abs(len(path) - int(g["path_rows"])) <= 4, (len(path), int(g["path_rows"]))
  frame = np.rint(path[:, 1] * 210.0).astype(np.int64)
  want_frame = np.rint(want20[:, 1] * 210.0).astype(np.int64)
  at = np.minimum(np.searchsorted(frame, want_frame), len(path) - 1)
  got = path[at]
  # (j1800, 25 jumps: the sub-frame offset of ONE of its 26 lines lands on the neighbouring step of the refinement's grid --
  # 0.042 frames = 0.2002 ms over the 21 s that line covers, 1.2 % of the rows, whichever GEMM precision and whichever way the LP
  # is solved (tests/gpu_probe_j1800.py); hence 5e-4 s here, with at least 98 % of the rows within the 2e-4 s the other cases keep)
  dv = np.abs(got[:, 0] - want20[:, 0])
  same = (frame[at] == want_frame) & (dv < 5e-4)
  assert same.mean() >= 0.999, f"{int((~same).sum())} of {len(same)} recorded path rows are not on the GPU path"
  tight = (frame[at] == want_frame) & (dv < 2e-4)
  assert tight.mean() >= (0.98 if name == "j1800" else 0.999), float(tight.mean())
  assert abs(len(np.unique(path[:, 2])) - len(np.unique(want20[:, 2]))) <= 1
  dq = np.abs(got[same, 3] - want20[same, 3])
  assert np.mean(dq < 0.1) >= 0.99 and np.median(dq) < 1e-2, (float(np.mean(dq < 0.1)), float(np.median(dq)))   # observed: >= 99.5 % within 0.1, median 2e-5 .. 2e-3
  np.testing.assert_allclose(got[same, 4], want20[same, 4], rtol=5e-3, atol=0.5)


def test_mismatched_pair_raises(ctx):
  from describealign_amd import align as A
  pair = cases.align_case("mismatch")
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  with pytest.raises(RuntimeError, match="Alignment failed, are the input files mismatched"):
    A.align(vf, af, vf[0], af[0], ctx=ctx)


def test_pipeline_equals_sequential(ctx):
  """The batch pipeline (LP in worker processes, next pair's GPU stages overlapped) returns
  exactly what align() returns, in submission order."""
  from describealign_amd import align as A
  pairs = [cases.align_case("a40"), cases.align_case("e180")]
  feats = [(ctx.features(p.video, 0), ctx.features(p.audio, 1)) for p in pairs]
  want = [A.align(vf, af, vf[0], af[0], ctx=ctx) for vf, af in feats]
  with A.AlignPipeline(ctx, lp_workers=2) as pipe:
    got = list(pipe.run(feats + feats))
  assert len(got) == 4
  for k, g in enumerate(got):
    w = want[k % 2]
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]) and g[2] == w[2] and g[4] == w[4]
    assert np.array_equal(g[3], w[3])


def test_pipeline_of_mixed_lengths_equals_sequential(ctx_bf16):
  """A batch whose pairs differ 12-fold in length, shortest and longest alternating: every chain DP finds different
  sizes in the slot it reuses (rank map, dense ranks of the matched frames, column plan) and takes a hand-over buffer from
  the context's pool that an earlier, larger or smaller DP returned -- results must be those of align() pair by pair."""
  from describealign_amd import align as A, synth
  lengths = [75.0, 900.0, 120.0, 610.0, 95.0, 333.0, 840.0, 60.0]
  pairs = [synth.make_pair(100 + k, sec, n_jumps=1 + k % 3, first_gap=20.0, channels=1 + k % 2) for k, sec in enumerate(lengths)]
  feats = [(ctx_bf16.features(p.video, 0), ctx_bf16.features(p.audio, 1)) for p in pairs]
  want = [A.align(vf, af, vf[0], af[0], ctx=ctx_bf16) for vf, af in feats]
  with A.AlignPipeline(ctx_bf16, lp_workers=3) as pipe:
    got = list(pipe.run(feats + feats[::-1]))
  assert len(got) == 2 * len(feats)
  for k, g in enumerate(got):
    w = want[k] if k < len(feats) else want[2 * len(feats) - 1 - k]
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]) and g[2] == w[2] and g[4] == w[4], k
    assert np.array_equal(g[3], w[3]), k


_TILED_WORKER = r"""
import os, sys, json
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))
import cases
from describealign_amd import _native, distrib, synth
from describealign_amd import align as A
backend = sys.argv[4] if len(sys.argv) > 4 else "gloo"   # gloo: all ranks share GPU 0 on the test box; RCCL needs one GPU per rank
import torch
g = distrib.Group(backend, init_single=True)
ctx = _native.Context(g.local_rank if (backend == "nccl" and g.world > 1) else 0, _native.PREC_F32)
name = sys.argv[3]
pair = cases.align_case(name) if name != "half_hour" else synth.make_pair(9, 1800.0, n_jumps=10, first_gap=120.0)
vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
tm = {}
x, y, sim, path, med = A.align_tiled(vf, af, vf[0], af[0], g, ctx=ctx, timings=tm)
np.savez(os.path.join(sys.argv[2], f"out{g.rank}.npz"), x=x, y=y, sim=sim, med=med, path=path)
print("rank", g.rank, "rows", tm["rows"], "matches", tm["n_matches"])
g.close(); ctx.close()
"""


def _run_tiled(tmp_path, world, name, port, backend="gloo"):
  import subprocess, sys
  script = tmp_path / "w.py"
  script.write_text(_TILED_WORKER)
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
  procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path), name, backend], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
  outs = [p.communicate(timeout=900)[0] for p in procs]
  assert all(p.returncode == 0 for p in procs), outs
  return [np.load(tmp_path / f"out{r}.npz") for r in range(world)]


def test_tiled_single_pair_two_ranks(tmp_path):
  """Config-5 style tiling: two ranks each match half of the audio rows, all-gather the match
  lists, and both arrive at the reference's nodes."""
  g = np.load(os.path.join(GOLD, "align_e180.npz"))
  for n in _run_tiled(tmp_path, 2, "e180", 29544):
    assert len(n["x"]) == len(g["x"])
    assert np.max(np.abs(n["x"] - g["x"])) < HOP_S and np.max(np.abs(n["y"] - g["y"])) < HOP_S


@pytest.mark.parametrize("world", [2, 8])
def test_tiled_half_hour_pair_equals_untiled_exactly(ctx, tmp_path, world):
  """The long-pair mode at a realistic size: an 1800 s pair (3.8e6 matches) matched in `world`
  contiguous audio-row blocks by as many processes (gloo; they share the one GPU here), gathered, and
  finished -- nodes, similarity, median slope and the whole path identical to the untiled align()."""
  from describealign_amd import align as A, synth
  pair = synth.make_pair(9, 1800.0, n_jumps=10, first_gap=120.0)
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx)
  for n in _run_tiled(tmp_path, world, "half_hour", 29560 + world):
    assert np.array_equal(n["x"], x) and np.array_equal(n["y"], y)
    assert float(n["sim"]) == sim and float(n["med"]) == med and np.array_equal(n["path"], path)


def test_tiled_over_rccl_one_rank_equals_untiled_exactly(ctx, tmp_path):
  """The RCCL branch of the long-pair mode on the one GPU there is: a ONE-rank "nccl" process group, so
  that align_tiled runs count all-gather -> da_match_export_device into the RCCL send buffers ->
  dist.gather -> torch.cat -> da_match_import_device -> device chain DP -> broadcast, all in device
  memory (distrib.Group.gather_matches_to_root, on_gpu path).  Result identical to align()."""
  from describealign_amd import align as A, synth
  pair = synth.make_pair(9, 1800.0, n_jumps=10, first_gap=120.0)
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx)
  (n,) = _run_tiled(tmp_path, 1, "half_hour", 29571, backend="nccl")
  assert np.array_equal(n["x"], x) and np.array_equal(n["y"], y)
  assert float(n["sim"]) == sim and float(n["med"]) == med and np.array_equal(n["path"], path)


def test_tiled_over_rccl_two_gpus_equals_untiled_exactly(ctx, tmp_path):
  """The one RCCL data path with MORE than one real rank (distrib.Group._gather_device: rank 0 reserves the gathered list in
  its context and posts the receives, rank 1 exports its block and sends it over xGMI): two ranks, one GPU each, on the
  1800 s pair -- nodes, similarity, slope and the whole path identical to the untiled align().  Skipped on a one-GPU box
  (RCCL refuses two ranks on one device); there the same code runs with one rank (test above) and under gloo."""
  import torch
  if torch.cuda.device_count() < 2:
    pytest.skip("needs two GPUs: RCCL refuses two ranks on one device")
  from describealign_amd import align as A, synth
  pair = synth.make_pair(9, 1800.0, n_jumps=10, first_gap=120.0)
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx)
  for n in _run_tiled(tmp_path, 2, "half_hour", 29573, backend="nccl"):
    assert np.array_equal(n["x"], x) and np.array_equal(n["y"], y)
    assert float(n["sim"]) == sim and float(n["med"]) == med and np.array_equal(n["path"], path)


@pytest.mark.slow
@pytest.mark.parametrize("seconds,min_matches", [(14400, 1e8), (28800, 5e8)])
def test_tiled_long_pair_eight_ranks_recovers_every_offset(seconds, min_matches):
  """configs[4] inside the test run, at half its stated length and AT its stated length: ONE 4 h / 8 h pair
  (2.9e8 / 1.1e9 matches), its matching stage tiled over 8 ranks (processes sharing the one GPU, gloo; the loop
  being tiled is describealign.py:658-682), gathered, chain DP on the device, LP, pass 2 -- every injected
  segment found, offsets within one hop of the truth."""
  import json, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  res = subprocess.run([sys.executable, os.path.join(root, "tests", "gpu_tiled_long_pair.py"), str(seconds), "8"],
                       capture_output=True, text=True, timeout=2400, env=dict(os.environ, DALIGN_DIST_BACKEND="gloo"))
  assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
  line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
  r = json.loads(line)
  assert r["world"] == 8 and r["seconds"] == float(seconds)
  assert r["nodes"] == 2 * r["segments_expected"], r            # two nodes per constant-offset segment
  assert r["max_offset_err_vs_injected_ms"] < 1e3 * HOP_S, r
  assert r["matches"] > min_matches and 50.0 < r["similarity"] <= 100.0, r


# ------------------------------------------------------------------------------------ properties at size
def _max_offset_error_ms(pair, x, y):
  """Every recovered segment's (audio - video) offset against the injected truth at its middle."""
  err = 0.0
  for k in range(0, len(x) - 1, 2):
    mid = 0.5 * (y[k] + y[k + 1])
    want = pair.true_offset_at(mid)
    err = max(err, abs((x[k] - y[k]) - want), abs((x[k + 1] - y[k + 1]) - want))
  return 1e3 * err


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_half_hour_pair_recovers_injected_offsets(ctx, ctx_bf16, prec):
  """Config-4 sized pair (1800 s, 10 jumps): no oracle run at this size in the test budget, so the
  size-independent property is checked instead: all injected offsets recovered within +-23 ms
  (match-set equality of the two GEMM precisions: test_bf16_prefilter_loses_no_match_at_size)."""
  from describealign_amd import align as A, synth
  c = ctx if prec == "f32" else ctx_bf16
  pair = synth.make_pair(9, 1800.0, n_jumps=10, first_gap=120.0)
  vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
  tm = {}
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=c, timings=tm)
  assert len(x) == 2 * (len(pair.jump_lengths))          # one segment per offset level
  assert _max_offset_error_ms(pair, x, y) < 23.0
  assert 60 < sim < 100 and abs(med - 1) < 1e-3


@pytest.mark.slow
def test_configs3_batch_of_32_distinct_half_hour_pairs(ctx_bf16):
  """BASELINE configs[3] at its stated shape: 32 DISTINCT synthetic 30-minute pairs (seeds 0 .. 31, 10 jumps each) through ONE
  AlignPipeline (the directory batch of describealign.py:1077-1078 as one GPU runs it).  Every pair: all injected offsets within
  +-23 ms, and the pipeline's result identical to sequential align() on the same features -- nodes, similarity, slope and the
  whole pass-2 path."""
  from describealign_amd import align as A, synth
  c = ctx_bf16
  pairs, feats = [], []
  for seed in range(32):
    pair = synth.make_pair(seed, 1800.0, n_jumps=10, first_gap=120.0)
    feats.append((c.features(pair.video, 0), c.features(pair.audio, 1)))
    pairs.append((pair.jump_video_times, pair.jump_lengths, pair))
    pair.video = pair.audio = None                      # 32 x 330 MB of PCM would not fit comfortably: the features are what is kept
  want = [A.align(vf, af, vf[0], af[0], ctx=c) for vf, af in feats]
  worst = 0.0
  with A.AlignPipeline(c, lp_workers=8) as pipe:
    got = list(pipe.run(feats, expected=len(feats)))
  assert len(got) == 32
  for k, (g, w) in enumerate(zip(got, want)):
    assert np.array_equal(g[0], w[0]) and np.array_equal(g[1], w[1]) and g[2] == w[2] and g[4] == w[4], f"pair {k}: nodes differ from sequential align()"
    assert np.array_equal(g[3], w[3]), f"pair {k}: pass-2 path differs from sequential align()"
    pair = pairs[k][2]
    assert len(g[0]) == 2 * len(pair.jump_lengths), f"pair {k}: {len(g[0]) // 2} segments for {len(pair.jump_lengths)} offset levels"
    worst = max(worst, _max_offset_error_ms(pair, g[0], g[1]))
  assert worst < 23.0, f"max offset error {worst:.2f} ms"


def _match_keys(c, vf, af):
  mi, mv, mq = c.match(vf, af)
  return (mi.astype(np.int64) << 32) | mv.astype(np.int64), mq


@pytest.mark.parametrize("seconds,channels", [(1800.0, 1), (7200.0, 1), (7200.0, 2)])
def test_bf16_prefilter_loses_no_match_at_size(ctx, ctx_bf16, seconds, channels):
  """The bf16 GEMM is a prefilter (rounding guard in the norm slot, csrc/dalign_common.h kBf16Guard); every
  survivor is re-verified in float64.  Its verified match SET (and every quality) must equal the f32 path's
  -- checked on whole pairs of config-3 and config-2 duration (3.8e6 / 7e7 matches), mono and the
  configs[2] stereo shape, not by count."""
  from describealign_amd import synth
  pair = synth.make_pair(9 if seconds < 3000 else 11, seconds, n_jumps=10, first_gap=120.0, channels=channels)
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  k32, q32 = _match_keys(ctx, vf, af)
  k16, q16 = _match_keys(ctx_bf16, [f.copy() for f in vf], [f.copy() for f in af])
  assert len(k32) > 1e6 * seconds / 1800
  assert np.all(np.diff(k32) > 0)                        # sorted by (i, v), no duplicates
  assert np.array_equal(k32, k16) and np.array_equal(q32, q16)


@pytest.mark.parametrize("seed,seconds", [(41, 150.0), (7, 333.3), (13, 61.7)])
def test_bf16_prefilter_equals_f32_with_gaps_in_the_row_lists(ctx, ctx_bf16, seed, seconds):
  """The bf16 GEMM streams 32-column tiles of NON-QUIET audio frames and keeps 192 video rows per
  wavefront: silences make the column frames non-consecutive (per-lane operand runs far apart) and odd
  durations leave partial tiles / partly empty row groups at every edge.  Verified set and qualities
  must equal the f32 path's."""
  from describealign_amd import synth
  pair = synth.make_pair(seed, seconds, n_jumps=3, first_gap=8.0)
  v = pair.video.copy(); a = pair.audio.copy()
  sr = synth.SAMPLE_RATE
  rng = np.random.default_rng(seed)
  for _ in range(6):                                      # silences of 0.2 .. 6 s on both sides
    for arr in (v, a):
      t0 = float(rng.uniform(1.0, seconds - 8.0)); d = float(rng.uniform(0.2, 6.0))
      arr[..., int(t0 * sr):int((t0 + d) * sr)] = 0
  vf = ctx.features(v, 0); af = ctx.features(a, 1)
  k32, q32 = _match_keys(ctx, vf, af)
  k16, q16 = _match_keys(ctx_bf16, [f.copy() for f in vf], [f.copy() for f in af])
  assert len(k32) > 1000
  assert np.array_equal(k32, k16) and np.array_equal(q32, q16)


def test_bf16_prefilter_repeats_itself_on_random_pairs(ctx, ctx_bf16):
  """k_match_bf16 places its MFMAs by hand (inline assembly), outside the compiler's hazard recogniser:
  a timing hazard would show as matches that come and go.  Random durations, jump counts and silences;
  every pair is matched three times and must equal the f32 path's set and qualities each time."""
  from describealign_amd import synth
  rng = np.random.default_rng(2024)
  sr = synth.SAMPLE_RATE
  for trial in range(10):
    secs = float(rng.uniform(40, 600)); seed = int(rng.integers(1, 10000))
    pair = synth.make_pair(seed, secs, n_jumps=int(rng.integers(1, 6)), first_gap=float(rng.uniform(2, 20)))
    v = pair.video.copy(); a = pair.audio.copy()
    for _ in range(int(rng.integers(0, 5))):
      for arr in (v, a):
        t0 = float(rng.uniform(1.0, max(2.0, secs - 8.0))); d = float(rng.uniform(0.1, 5.0))
        arr[..., int(t0 * sr):int((t0 + d) * sr)] = 0
    vf = ctx.features(v, 0); af = ctx.features(a, 1)
    k32, q32 = _match_keys(ctx, vf, af)
    for rep in range(3):
      k16, q16 = _match_keys(ctx_bf16, [f.copy() for f in vf], [f.copy() for f in af])
      assert np.array_equal(k32, k16) and np.array_equal(q32, q16), (trial, rep, secs, seed, len(k32), len(k16))


def test_silence_heavy_pair_vs_oracle(ctx):
  """Long stretches of digital silence on both sides (quiet frames are excluded from matching,
  :629-630, :657-658): GPU path vs the oracle, end to end."""
  from describealign_amd import align as A, synth
  pair = synth.make_pair(41, 150.0, jumps=([0.0, 70.0], [6.0, 2.5]))
  v = pair.video.copy(); a = pair.audio.copy()
  sr = synth.SAMPLE_RATE
  for t0, t1 in ((20, 32), (95, 101), (120, 124)):
    v[:, t0 * sr:t1 * sr] = 0
  a[:, 40 * sr:47 * sr] = 0
  vf = ctx.features(v, 0); af = ctx.features(a, 1)
  assert np.mean(vf[0] <= 0.5) > 0.1                      # a real share of quiet frames
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx)
  ovf, oaf = O.features(v), O.features(a)
  ox, oy, osim, opath, omed = O.align(ovf, oaf, ovf[0], oaf[0])
  assert len(x) == len(ox)
  assert np.max(np.abs(x - ox)) < HOP_S and np.max(np.abs(y - oy)) < HOP_S
  assert abs(sim - osim) < 0.5


def test_interleaved_stereo_end_to_end(ctx):
  """s16le frames as ffmpeg emits them (N, 2) give the same nodes as the planar (2, N) layout."""
  from describealign_amd import align as A
  pair = cases.align_case("e180s")
  g = np.load(os.path.join(GOLD, "align_e180s.npz"))
  vf = ctx.features(np.ascontiguousarray(pair.video.T), 0)
  af = ctx.features(np.ascontiguousarray(pair.audio.T), 1)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=ctx)
  assert len(x) == len(g["x"]) and np.max(np.abs(x - g["x"])) < HOP_S and np.max(np.abs(y - g["y"])) < HOP_S


def test_c_abi_argument_errors(ctx, native):
  import ctypes as C
  lib = native.load()
  h = ctx._h
  assert lib.da_pcm_upload(h, 5, None, 0, 1, 1) == -1                      # bad side / null
  pcm = np.zeros((1, 1000), dtype=np.int16)
  assert lib.da_pcm_upload(h, 0, pcm.ctypes.data_as(C.c_void_p), 1000, 3, 1) == -1     # 3 channels
  assert b"bad argument" in lib.da_last_error(h)
  with pytest.raises(ValueError):
    ctx.match([np.zeros(10, np.float32)] * 5, [np.zeros(12, np.float32)] + [np.zeros(10, np.float32)] * 4)
  # all-quiet input: no rows to match -> empty match list, and the chain raises the reference's error
  vf = [np.zeros(500, np.float32)] * 5
  mi, mv, mq = ctx.match(vf, vf)
  assert len(mi) == 0
  with pytest.raises(RuntimeError, match="Alignment failed"):
    ctx.chain(mi, mv, mq, min_len=1050)
  n = C.c_int64(0)
  assert lib.da_match_fetch(h, None, None, None, 5) == -1                  # more than resident


# ------------------------------------------------------------------------------------ randomized edges
@pytest.mark.parametrize("seed", range(8))
def test_features_random_lengths_and_extremes_vs_oracle(ctx, seed):
  """Ragged lengths in every residue class of 105/210, mono and stereo, full int16 range
  (including -32768 and runs of equal samples), against the oracle."""
  from describealign_amd import synth
  rng = np.random.default_rng(seed)
  n = int(rng.integers(4000, 400000))
  n += [0, 1, 104, 105, 106, 209, 211, 315][seed]
  c = 1 + (seed % 2)
  base = synth.programme(100 + seed, n).astype(np.int64) * int(rng.integers(1, 4))
  pcm = np.clip(base, -32768, 32767).astype(np.int16)
  pcm[rng.integers(0, n, 50)] = -32768
  pcm[rng.integers(0, n, 50)] = 32767
  k = int(rng.integers(0, n - 3000)); pcm[k:k + 2500] = 0                 # a silent stretch (sign stays +)
  k = int(rng.integers(0, n - 3000)); pcm[k:k + 700] = -3                  # constant negative run
  arr = pcm[None, :] if c == 1 else np.stack([pcm, np.roll(pcm, 37) // 2 * -1])
  arr = np.ascontiguousarray(arr.astype(np.int16))
  rows = ctx.features(arr)
  want = O.features(arr)
  for kk, (f, r) in enumerate(zip(rows, want)):
    assert f.shape == r.shape, (kk, f.shape, r.shape)
    np.testing.assert_allclose(f, np.asarray(r, dtype=np.float32), rtol=3e-5, atol=3e-5, err_msg=f"row {kk} n={n} c={c}")


def test_refine_vs_oracle_on_ten_minute_pair(ctx):
  """da_refine (banded extension kernels + second DP) against the oracle's restatement, fed with
  the oracle's own pass-1 / LP / cluster intermediates for the 600 s pair (16+ clusters,
  ~3e5 banded points) -- a much larger instance than the a40 golden."""
  pair = cases.align_case("e600")
  vf, af = O.features(pair.video), O.features(pair.audio)
  st = {}
  ox, oy, osim, opath, omed = O.align(vf, af, vf[0], af[0], stages=st)
  cl = st["clusters"]
  x0 = np.array([c[0][0] for c in cl]); x1 = np.array([c[0][-1] for c in cl])
  off = np.array([c[1] for c in cl]); slo = np.array([c[2] for c in cl])
  path, npts = ctx.refine(st["a_scaled"], st["v_scaled"], x0, x1, off, slo)
  assert npts == sum(len(p) for p in st["points"])
  want = opath.copy(); want[:, :2] *= 210.0
  assert path.shape == want.shape
  np.testing.assert_allclose(path[:, :3], want[:, :3], atol=1e-6)
  np.testing.assert_allclose(path[:, 3:], want[:, 3:], atol=2e-3)


def test_directory_batch_is_pipelined_and_identical_to_sequential(ctx, tmp_path):
  """combine.process_batch (what `combine` uses for a directory on one GPU) writes the same report
  for every pair as process_pair does one pair at a time."""
  from describealign_amd import combine, media, synth
  todo = []
  for k in range(4):
    pair = synth.make_pair(seed=60 + k, video_seconds=60.0 + 3 * k, jumps=([0.0, 25.0 + k], [4.0 + k, 1.5]))
    v, a = str(tmp_path / f"ep{k}.wav"), str(tmp_path / f"ep{k}_ad.wav")
    media.write_wav(v, pair.video); media.write_wav(a, pair.audio)
    todo.append((v, a, False))
  seq_dir, bat_dir = tmp_path / "seq", tmp_path / "bat"
  for d in (seq_dir, bat_dir):
    os.makedirs(d / "out"); os.makedirs(d / "plots")
  seq = [combine.process_pair(v, a, alt, ctx, output_dir=str(seq_dir / "out"), alignment_dir=str(seq_dir / "plots"))
         for v, a, alt in todo]
  # the shipped batch takes the fused device stage (da_pair_stage: one native call per pair), not the five separate calls
  from describealign_amd import _native
  calls = {"pair_stage": 0, "match_begin": 0}
  real_stage, real_begin = _native.Context.pair_stage, _native.Context.match_begin

  def counted_stage(self, *a, **k):
    calls["pair_stage"] += 1
    return real_stage(self, *a, **k)

  def counted_begin(self, *a, **k):
    calls["match_begin"] += 1
    return real_begin(self, *a, **k)

  _native.Context.pair_stage, _native.Context.match_begin = counted_stage, counted_begin
  try:
    bat = combine.process_batch(todo, ctx, output_dir=str(bat_dir / "out"), alignment_dir=str(bat_dir / "plots"), lp_workers=2)
    st_dir = tmp_path / "st"
    os.makedirs(st_dir / "out"); os.makedirs(st_dir / "plots")
    stretched = combine.process_batch([(v, a, True) for v, a, _ in todo[:3]], ctx, output_dir=str(st_dir / "out"), alignment_dir=str(st_dir / "plots"), lp_workers=2,
                                      stretch_audio=True)
  finally:
    _native.Context.pair_stage, _native.Context.match_begin = real_stage, real_begin
  assert calls == {"pair_stage": 7, "match_begin": 0}, calls
  assert len(stretched) == 3
  for s_, b_ in zip(seq, stretched):               # --stretch_audio aligns the stereo decode of the same files: same jumps found
    assert len(s_["audio_desc_times"]) == len(b_["audio_desc_times"])
    assert np.max(np.abs(np.asarray(s_["video_times"]) - np.asarray(b_["video_times"]))) < 0.023
  assert len(bat) == len(seq) == 4
  for s_, b_ in zip(seq, bat):
    assert np.array_equal(s_["audio_desc_times"], b_["audio_desc_times"]) and np.array_equal(s_["video_times"], b_["video_times"])
    assert s_["setts"] == b_["setts"]
    assert open(s_["report"]).read() == open(b_["report"]).read()
    assert os.path.getsize(b_["report"][:-4] + ".png") > 10000


def test_combine_two_gpu_workers_share_one_device(ctx, tmp_path, monkeypatch):
  """combine(gpus=2) -- BASELINE config 4's path: the directory is sharded round-robin over two worker
  processes, one per GPU.  With one GPU here both workers are pointed at device 0
  (DALIGN_DEVICE_OVERRIDE); every pair's report must equal the one-pair-at-a-time result."""
  from describealign_amd import combine, media, synth
  vids, auds = tmp_path / "v", tmp_path / "a"
  os.makedirs(vids); os.makedirs(auds)
  todo = []
  for k in range(6):
    pair = synth.make_pair(seed=80 + k, video_seconds=50.0 + 2 * k, jumps=([0.0, 20.0 + k], [3.0 + k, 1.5]))
    v, a = str(vids / f"ep{k}.wav"), str(auds / f"ep{k}.wav")
    media.write_wav(v, pair.video); media.write_wav(a, pair.audio)
    todo.append((v, a))
  seq_dir = tmp_path / "seq"; par_dir = tmp_path / "par"
  for d in (seq_dir, par_dir):
    os.makedirs(d / "out"); os.makedirs(d / "plots")
  # audio-only inputs need --stretch_audio (:1092); it also exercises the replacement stage in the workers
  seq = [combine.process_pair(v, a, True, ctx, stretch_audio=True, output_dir=str(seq_dir / "out"), alignment_dir=str(seq_dir / "plots"))
         for v, a in todo]
  monkeypatch.setenv("DALIGN_DEVICE_OVERRIDE", "0")
  combine.combine(str(vids), str(auds), stretch_audio=True, yes=True, output_dir=str(par_dir / "out"),
                  alignment_dir=str(par_dir / "plots"), gpus=2)
  for k, s_ in enumerate(seq):
    want = open(s_["report"]).read()
    got = open(par_dir / "plots" / f"ep{k}.txt").read()
    strip = lambda t: "\n".join(l for l in t.splitlines() if not l.startswith("(no ffmpeg") and "FFmpeg command" not in l and str(tmp_path) not in l)
    assert strip(got) == strip(want), k
    assert os.path.getsize(par_dir / "plots" / f"ep{k}.png") > 10000


def test_two_hour_stereo_pair_recovers_injected_offsets(ctx_bf16):
  """BASELINE config 3 at full size (7200 s stereo, 10 injected offset jumps, bf16 similarity
  GEMM): every injected offset recovered within +-23 ms, in far less than 1 % of real time, and the
  --stretch_audio track built from those nodes has the right length."""
  import time
  from describealign_amd import align as A, synth
  pair = synth.make_pair(11, 7200.0, n_jumps=10, first_gap=200.0, channels=2)
  c = ctx_bf16
  t0 = time.perf_counter()
  vf = c.features(pair.video, 0); af = c.features(pair.audio, 1)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=c)
  elapsed = time.perf_counter() - t0
  assert len(x) == 2 * len(pair.jump_lengths)
  assert _max_offset_error_ms(pair, x, y) < 23.0
  assert abs(med - 1) < 1e-3
  assert elapsed < 72.0, f"{elapsed:.1f} s is slower than 100x real time"
  track, fac = c.stretch_resident(x, y, False)
  assert track.shape == (pair.video.shape[1], 2) and np.all(np.isfinite(fac))
  assert int(np.abs(track.astype(np.int32)).max()) >= 32000            # peak-normalised


def test_resident_rows_fast_path_equals_uploaded_rows(ctx, native):
  """Rows that features_resident() just produced are matched from their device copies (no re-pack,
  no upload); copies of the same rows take the upload path: identical matches either way."""
  pair = cases.align_case("a40")
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  assert ctx._resident_rows(0, vf) is not None and ctx._resident_rows(1, af) is not None
  fast = ctx.match(vf, af)
  vf2, af2 = [f.copy() for f in vf], [f.copy() for f in af]
  assert ctx._resident_rows(0, vf2) is None
  slow = ctx.match(vf2, af2)
  for a_, b_ in zip(fast, slow):
    assert np.array_equal(a_, b_)
  assert len(fast[0]) > 1000
  # a second upload for one side invalidates its resident rows
  ctx.pcm_upload(0, pair.video)
  assert ctx._resident_rows(0, vf) is None


def test_match_list_device_export_import(ctx, native):
  """The device-to-device hand-over used by the multi-GPU long-pair mode (RCCL buffers): the resident
  match list exported into torch CUDA tensors equals the fetched one; imported back in two pieces
  glued together, the device chain DP gives the path of the whole list."""
  import torch
  pair = cases.align_case("e180")
  vf = ctx.features(pair.video, 0); af = ctx.features(pair.audio, 1)
  mi, mv, mq = ctx.match(vf, af)
  want = native.chain_host(mi, mv, mq)
  n = len(mi)
  keys = torch.zeros(n, dtype=torch.int64, device="cuda"); qual = torch.zeros(n, dtype=torch.float64, device="cuda")
  ctx.match_export_device(keys.data_ptr(), qual.data_ptr(), n)
  assert np.array_equal(keys.cpu().numpy(), (mi.astype(np.int64) << 32) | mv) and np.array_equal(qual.cpu().numpy(), mq)
  # two "ranks": the audio-row halves matched separately, gathered on the device, imported
  la = len(af[0]) - 41
  parts_k, parts_q = [], []
  for rows in ((0, la // 2), (la // 2, la)):
    k = ctx.match_begin(vf, af, rows=rows); m = ctx.match_finish()
    pk = torch.zeros(max(m, 1), dtype=torch.int64, device="cuda"); pq = torch.zeros(max(m, 1), dtype=torch.float64, device="cuda")
    ctx.match_export_device(pk.data_ptr(), pq.data_ptr(), m)
    parts_k.append(pk[:m]); parts_q.append(pq[:m])
  all_k = torch.cat(parts_k); all_q = torch.cat(parts_q)
  torch.cuda.synchronize()
  assert torch.equal(all_k, keys) and torch.equal(all_q, qual)
  ctx.match_import_device(all_k.data_ptr(), all_q.data_ptr(), int(all_k.numel()))
  ctx.trim()                                            # scratch given back: the resident list must survive it
  fi, fv, fq = ctx.match_fetch(n)                      # the imported list is "the resident match" in every respect
  assert np.array_equal(fi, mi) and np.array_equal(fv, mv) and np.array_equal(fq, mq)
  gi, gv = ctx.chain_resident()                         # (collecting the DP releases the list)
  assert np.array_equal(gi, want[0]) and np.array_equal(gv, want[1])


def test_async_pinned_upload_equals_blocking_upload(ctx, native):
  """da_host_alloc + da_pcm_upload_async (page-locked source, copy on the copy stream, the feature
  kernel waits on the device) gives the same rows as the blocking da_pcm_upload, for both layouts;
  the copy's duration is reported."""
  pair = cases.align_case("e180s")
  want_v = [f.copy() for f in ctx.features(pair.video, 0)]
  want_a = [f.copy() for f in ctx.features(pair.audio, 1)]
  pv = native.pinned_empty(pair.video.shape); pv[...] = pair.video
  pa = native.pinned_empty((pair.audio.shape[1], 2)); pa[...] = pair.audio.T            # interleaved frames
  ctx.pcm_upload_async(0, pv); ctx.pcm_upload_async(1, pa)
  got_v = ctx.features_resident(0)
  assert ctx.stats()["h2d_ms"] > 0
  got_a = ctx.features_resident(1)
  for g_, w_ in zip(got_v + got_a, want_v + want_a):
    assert np.array_equal(g_, w_)
  del pv, pa                                            # frees the page-locked memory with the last view


def test_resident_clips_rotate_through_one_context_without_a_copy(native):
  """da_pcm_exchange (bench.py's stream of distinct pairs): three clips of different lengths, one in the context and two in
  streams, rotate through the context side for three rounds -- each time the feature rows equal those of a plain upload of that
  clip, and a stream that received the context's PCM knows its length.  A planar upload cannot be exchanged (like swaps with like)."""
  c = native.Context(0, native.PREC_F32)
  clips = [cases.feature_clip(n) for n in ("stereo", "stereo_anti")] + [np.ascontiguousarray(cases.feature_clip("stereo")[:, 5000:150000])]
  want = [[r.copy() for r in c.features(p, 0)] for p in clips]

  def as_stream(p):
    st = native.PcmStream(0, 2, p.shape[1])
    st.piece(np.ascontiguousarray(p.T)); st.sync()
    return st

  c.pcm_upload(0, clips[0])                                  # planar: cannot take part
  probe = as_stream(clips[1])
  with pytest.raises(RuntimeError, match="interleaved"):
    c.pcm_exchange(0, probe)
  probe.close()
  first = as_stream(clips[0])
  c.pcm_adopt(0, first); first.close()
  slots = [as_stream(clips[1]), as_stream(clips[2])]
  held = {"ctx": 0, "slots": [1, 2]}
  for step in range(1, 10):
    p = step % 3
    if held["ctx"] != p:
      k = held["slots"].index(p)
      n_before = slots[k].frames
      assert n_before == clips[p].shape[1]
      c.pcm_exchange(0, slots[k])
      held["ctx"], held["slots"][k] = p, held["ctx"]
      assert slots[k].frames == clips[held["slots"][k]].shape[1]          # the stream now holds the clip the context held
    got = c.features_resident(0)
    assert all(np.array_equal(got[r], want[p][r]) for r in range(5)), (step, p)
  for st in slots:
    st.close()
  c.close()


def test_streaming_ingest_pipe_to_hbm_equals_blocking_upload(ctx, native, tmp_path, monkeypatch):
  """SURVEY section 8(f) item 1 / describealign.py:149-157: decoder pipe -> ring of page-locked pieces -> HBM
  (media.stream_file_to_device + da_pcm_stream_* + da_pcm_adopt).  The feature rows of the adopted buffer equal those
  of da_pcm_upload of the same PCM -- through the decoder process (a test double, tests/doubles/fake_decoder.py) and
  for a natively read WAV, with pieces small enough that the ring is reused many times and the device buffer has to
  grow (the pipe's length is unknown) -- and the host side allocates nothing beyond the ring."""
  import tracemalloc
  from describealign_amd import media
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "doubles"))
  import fake_decoder
  bindir = tmp_path / "bin"; bindir.mkdir()
  fake_decoder.install(bindir)
  monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ.get("PATH", ""))
  rng = np.random.default_rng(3)
  n = 44100 * 240 + 1234                                              # 4 minutes: 42 MB stereo
  env = (np.sin(np.arange(n) / 5000.0) * 0.5 + 0.5)
  st = (rng.integers(-20000, 20000, size=(2, n)) * env).astype(np.int16)
  media.write_wav(str(tmp_path / "clip.wav"), st)
  os.link(tmp_path / "clip.wav", tmp_path / "clip.mka")               # same bytes under a name only the decoder takes
  want = {c: ctx.features(st if c == 2 else ((st[0].astype(np.int32) + st[1] + 1) >> 1).astype(np.int16)[None, :], 0) for c in (1, 2)}
  piece = 1 << 20
  ring = [native.pinned_empty((piece // 2,), np.int16) for _ in range(3)]
  for name, channels in (("clip.mka", 2), ("clip.mka", 1), ("clip.wav", 2)):
    stream = native.PcmStream(ctx.device, channels)                   # no length hint: grows on the device
    tracemalloc.start()
    frames = media.stream_file_to_device(stream, str(tmp_path / name), channels, ring, piece_bytes=piece)
    _, peak = tracemalloc.get_traced_memory(); tracemalloc.stop()
    assert frames == n and stream.frames == n
    assert peak < (4 << 20), f"streaming {name} allocated {peak} bytes on the host (file: {st.nbytes})"
    ctx.pcm_adopt(1, stream)                                          # side 1 this time
    assert stream.frames == 0                                         # empty again, reusable
    got = ctx.features_resident(1)
    for k in range(5):
      assert np.array_equal(got[k], want[channels][k]), (name, channels, k)
    # the same stream object takes the next file (its buffer is now the side's previous one)
    frames = media.stream_file_to_device(stream, str(tmp_path / "clip.wav"), 2, ring, piece_bytes=piece) if channels == 2 else 0
    if frames:
      ctx.pcm_adopt(0, stream)
      again = ctx.features_resident(0)
      assert all(np.array_equal(again[k], want[2][k]) for k in range(5))
    stream.close()
  # a stream of another channel count than its frames, or torn input, is refused by the binding
  stream = native.PcmStream(ctx.device, 2)
  with pytest.raises(ValueError):
    stream.piece(np.zeros(5, dtype=np.int16))
  stream.close()


def test_combine_default_path_through_decoder_probe_and_mux_doubles(ctx, tmp_path, monkeypatch):
  """The reference's default mode (describealign.py:1162-1169: no --stretch_audio) end to end with the external
  binaries replaced by test doubles (tests/doubles/fake_decoder.py; this image has no ffmpeg): both inputs are decoded
  through the pipe into HBM, aligned, the key frames around the audio start are asked of ffprobe
  (get_closest_key_frame_time :451-458), and ffmpeg is run with the re-timing command line of :489-510 -- the argv it
  RECEIVED is compared with the one built from the alignment; a second call skips the finished output (:1087-1089)."""
  import json
  from describealign_amd import combine, media, report
  sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "doubles"))
  import fake_decoder
  bindir = tmp_path / "bin"; bindir.mkdir()
  fake_decoder.install(bindir)
  monkeypatch.setenv("PATH", str(bindir) + os.pathsep + os.environ.get("PATH", ""))
  pair = cases.align_case("e180")
  media.write_wav(str(tmp_path / "show.wav"), pair.video[None, :] if pair.video.ndim == 1 else pair.video)
  media.write_wav(str(tmp_path / "ad.wav"), pair.audio[None, :] if pair.audio.ndim == 1 else pair.audio)
  os.rename(tmp_path / "show.wav", tmp_path / "show.mkv"); os.rename(tmp_path / "ad.wav", tmp_path / "ad.mka")
  out_dir, plot_dir = str(tmp_path / "out"), str(tmp_path / "plots")
  os.makedirs(out_dir); os.makedirs(plot_dir)
  res = combine.process_pair(str(tmp_path / "show.mkv"), str(tmp_path / "ad.mka"), False, ctx, output_dir=out_dir, alignment_dir=plot_dir)
  g = np.load(os.path.join(GOLD, "align_e180.npz"))
  assert np.max(np.abs(res["audio_desc_times"] - g["x"])) < HOP_S and np.max(np.abs(res["video_times"] - g["y"])) < HOP_S
  got = json.loads(open(os.path.join(out_dir, "ad_show.mkv")).read().split("\n")[0])["argv"]
  x, y = res["audio_desc_times"], res["video_times"]
  offset = y[0] - x[0]
  keys = np.arange(0.0, max(60, offset + 40) + 1e-9, fake_decoder.KEY_FRAME_STEP)
  after = combine.closest_key_frame_time(keys, offset)
  want = combine._mux_command(str(bindir / "ffmpeg"), str(tmp_path / "show.mkv"), str(tmp_path / "ad.mka"), os.path.join(out_dir, "ad_show.mkv"),
                              report.encode_fit_as_ffmpeg_expr(x, y, offset), offset, after, res["median_slope"])
  assert got == want[1:], (got, want[1:])
  assert "-acodec" in got and got[got.index("-acodec") + 1] == "copy"          # not a .wav description: stream copy (:499)
  assert res["setts"] in " ".join(got) and os.path.exists(res["report"])
  assert "FFmpeg command:" in open(res["report"]).read()
  # skip-existing rule
  assert combine.process_pair(str(tmp_path / "show.mkv"), str(tmp_path / "ad.mka"), False, ctx, output_dir=out_dir, alignment_dir=plot_dir) is None
  # --stretch_audio mux: the track goes in through the pipe; a video whose first audio track already is a description
  os.rename(tmp_path / "show.mkv", tmp_path / "described_show.mkv")
  res2 = combine.process_pair(str(tmp_path / "described_show.mkv"), str(tmp_path / "ad.mka"), False, ctx, stretch_audio=True,
                              output_dir=out_dir, alignment_dir=plot_dir)
  rec = json.loads(open(os.path.join(out_dir, "ad_described_show.mkv")).read().split("\n")[0])
  assert rec["argv"][:10] == ["-f", "s16le", "-ac", "2", "-acodec", "pcm_s16le", "-ar", "44100", "-i", "pipe:"]
  assert "visual_impaired+descriptions" in rec["argv"][rec["argv"].index("-disposition:a:1") + 1]
  assert rec["stdin_bytes"] == 4 * len(pair.video if pair.video.ndim == 1 else pair.video[0])    # stereo s16le frames of the video's length
  assert res2 is not None


def test_bench_launch_contract_two_ranks(tmp_path):
  """The driver launches bench.py under torch.distributed.run, one rank per GPU.  With only one
  GPU here both ranks share it (gloo instead of RCCL, which refuses two ranks on one device): rank 0
  must print exactly one JSON line with the contract's keys."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, DALIGN_DIST_BACKEND="gloo", DALIGN_BENCH_DEVICE="0")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1",
         "--pipeline", "2", "--no-cpu-baseline", "--workload", "cfg-small"]
  res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
  assert res.returncode == 0, res.stderr[-2000:]
  lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, res.stdout[-2000:]
  d = json.loads(lines[0])
  for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
    assert key in d, key
  assert d["n_gpus"] == 2 and d["steps"] == 4 and d["scaling"] == "weak" and d["value"] > 0
  assert d["roofline"]["bound"] == "mfma" and 0 < d["roofline"]["frac"] < 1
  # the line says what bounds it and what the untimed lead-in really was
  assert d["bound"] in ("host_lp", "gpu") and d["lead_in_pairs_actual"] >= d["warmup"] and d["whole_stream_value"] > 0
  assert d["lp_solves_per_s_host"] == pytest.approx(2 * d["lp_solves_per_s_rank"], rel=1e-3) and d["gpu_stage_pairs_per_s"] > 0


def test_plain_bench_command_starts_its_own_ranks():
  """`python bench.py --gpus 2` WITHOUT a launcher (how a scaling run may be started): the process must start two ranks itself
  (torch.distributed.run as a child, before it touches the GPU) and relay rank 0's line; n_gpus is the process group's own size.
  Both ranks share the one GPU here (gloo; RCCL refuses two ranks on one device).  With RCCL, asking for more ranks than there
  are GPUs fails loudly instead of measuring one GPU."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
  cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--pipeline", "2",
         "--no-cpu-baseline", "--workload", "cfg-small"]
  res = subprocess.run(cmd, capture_output=True, text=True, timeout=600,
                       env=dict(env, DALIGN_DIST_BACKEND="gloo", DALIGN_BENCH_DEVICE="0", DALIGN_BENCH_TILED_SECONDS="600"), cwd=root)
  assert res.returncode == 0, res.stderr[-2000:]
  lines = [l for l in res.stdout.splitlines() if l.strip()]
  assert len(lines) == 1 and lines[0].startswith("{"), res.stdout[-2000:]
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 == d["ranks_in_group"] and d["steps"] == 4 and d["value"] > 0
  # with more than one rank the line also carries the one workload that has an exchange step: a single pair tiled over the ranks
  t = d["secondary_tiled"]
  assert "error" not in t, t
  assert t["ranks"] == 2 and t["backend"] == "gloo" and t["video_seconds"] == 600.0 and t["matches"] > 1e5
  assert t["nodes"] == 2 * t["segments_expected"] and t["max_offset_err_vs_injected_ms"] < 23.0 and t["gathered_bytes"] > 0
  import torch
  if torch.cuda.device_count() < 8:
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--workload", "cfg-small"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert res.returncode != 0 and "needs 8 visible GPUs" in res.stderr and not res.stdout.strip()


def test_one_call_pair_stage_equals_the_five_calls(ctx_bf16, native):
  """da_pair_stage (features of both resident sides + match_begin + match_finish + chain_begin in one native call, what the
  GPU-feeding thread of a batch uses) against the separate calls: identical rows, matches, path; and a batch fed through it
  equals sequential align()."""
  from describealign_amd import align as A, synth
  c = ctx_bf16
  pair = cases.align_case("e600")
  c.pcm_upload(0, pair.video); c.pcm_upload(1, pair.audio)
  vf = [r.copy() for r in c.features_resident(0)]; af = [r.copy() for r in c.features_resident(1)]
  mi, mv, mq = c.match(vf, af)
  pi, pv = c.chain_resident()
  vf2, af2, n, ticket = c.pair_stage()
  assert n == len(mi)
  for a_, b_ in zip(vf + af, list(vf2) + list(af2)):
    assert a_.shape == b_.shape and np.array_equal(a_, b_)
  gi, gv, gq = c.match_fetch(n)
  assert np.array_equal(gi, mi) and np.array_equal(gv, mv) and np.array_equal(gq, mq)
  qi, qv = c.chain_finish(ticket)
  assert np.array_equal(qi, pi) and np.array_equal(qv, pv)
  st = c.stats()
  assert st["features_ms"] > 0 and st["features_bytes"] == pytest.approx(2.0 * (pair.video.size + pair.audio.size) + 20.0 * (len(vf[1]) + len(af[1])))
  want = A.align(vf, af, vf[0], af[0], ctx=c)
  with A.AlignPipeline(c, lp_workers=2) as pipe:
    got = list(pipe.run([lambda ctx_: A.RESIDENT_PCM] * 3, expected=3))
  for g in got:
    assert np.array_equal(g[0], want[0]) and np.array_equal(g[1], want[1]) and g[2] == want[2] and np.array_equal(g[3], want[3])


def test_pair_stage_edge_cases(native):
  """da_pair_stage on inputs the batch loop can meet: no PCM uploaded (error, not a crash), clips shorter than the 41-frame
  window (no rows, no matches: an empty DP whose collection reports the reference's mismatch error once a minimum length is
  asked for), and silence (every frame quiet: empty row lists)."""
  c = native.Context(0, native.PREC_BF16)
  try:
    with pytest.raises(RuntimeError, match="no PCM"):
      c.pair_stage()
    rng = np.random.default_rng(3)
    short = rng.integers(-3000, 3000, size=(1, 4410), dtype=np.int16)          # 0.1 s: 21 frames
    c.pcm_upload(0, short); c.pcm_upload(1, short)
    vf, af, n, t = c.pair_stage()
    assert n == 0 and len(vf[0]) == 21 and len(af[1]) == 21
    pi, pv = c.chain_finish(t)
    assert len(pi) == 0 and len(pv) == 0
    vf, af, n, t = c.pair_stage()
    with pytest.raises(RuntimeError, match="mismatched"):
      c.chain_finish(t, min_len=1050.0)
    quiet = np.zeros((2, 44100 * 20), dtype=np.int16)
    c.pcm_upload(0, quiet); c.pcm_upload(1, quiet)
    vf, af, n, t = c.pair_stage()
    assert n == 0 and float(np.max(vf[0])) == 0.0
    assert len(c.chain_finish(t)[0]) == 0
    # and a normal pair right after, on the same context
    pair = cases.align_case("a40")
    c.pcm_upload(0, pair.video); c.pcm_upload(1, pair.audio)
    vf, af, n, t = c.pair_stage()
    g = np.load(os.path.join(GOLD, "align_a40.npz"))
    pi, pv = c.chain_finish(t)
    assert n > 0 and len(pi) > 100
  finally:
    c.close()


_SWITCH_WORKER = r"""
import os, sys, json
import numpy as np
root = sys.argv[1]
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests", "golden"))

def main():
  import cases
  from describealign_amd import _native as native, align as A
  c = native.Context(0, native.PREC_BF16)
  pair = cases.align_case("e180")
  c.pcm_upload(0, pair.video); c.pcm_upload(1, pair.audio)
  vf, af, n, t = c.pair_stage()
  pi, pv = c.chain_finish(t)
  x, y, sim, path, med = A.align(vf, af, vf[0], af[0], ctx=c)
  with A.AlignPipeline(c, lp_workers=2) as pipe:
    got = list(pipe.run([lambda ctx_: A.RESIDENT_PCM] * 2, expected=2))
  same = all(np.array_equal(g[0], x) and np.array_equal(g[1], y) and np.array_equal(g[3], path) for g in got)
  print(json.dumps(dict(n=int(n), path=int(len(pi)), x=[float(v) for v in x], y=[float(v) for v in y], sim=float(sim), rows=int(path.shape[0]), pipeline_same=bool(same),
                        cols=int(c.stats()["chain_columns"]))))
  c.close()

if __name__ == "__main__":          # the pipeline's worker processes are spawned: they import this file
  main()
"""


def test_runtime_switches_do_not_change_results(tmp_path):
  """The round-5 mechanisms are about WHEN things run, never about what they compute: without the chain DP's CU mask
  (DALIGN_CHAIN_CUS=0), with other masks, with the runtime's blocking wait instead of polling (DALIGN_BLOCKING_SYNC=1) and with
  unpinned pipeline threads the same pair gives the same matches, path, nodes and pass-2 rows -- and the reference's nodes."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  script = tmp_path / "w.py"; script.write_text(_SWITCH_WORKER)
  outs = []
  for extra in ({}, {"DALIGN_CHAIN_CUS": "0"}, {"DALIGN_CHAIN_CUS": "2"}, {"DALIGN_BLOCKING_SYNC": "1"}, {"DALIGN_PIN_THREADS": "0", "DALIGN_PIN_WORKERS": "0"}):
    env = dict(os.environ, **extra)
    res = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, (extra, res.stderr[-2000:])
    outs.append(json.loads(res.stdout.strip().splitlines()[-1]))
  for o in outs[1:]:
    assert o == outs[0], (o, outs[0])
  assert outs[0]["pipeline_same"] and outs[0]["n"] > 0
  g = np.load(os.path.join(GOLD, "align_e180.npz"))
  assert np.max(np.abs(np.array(outs[0]["x"]) - g["x"])) < HOP_S and np.max(np.abs(np.array(outs[0]["y"]) - g["y"])) < HOP_S
