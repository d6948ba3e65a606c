#!/usr/bin/env python3
"""Generate golden fixtures by RUNNING the reference (julbean/describealign v2.0.8).

Runs only in the build container, where the reference is mounted read-only at
/root/reference; nothing from the reference is copied - it is imported, driven on
deterministic synthetic PCM (tests/golden/cases.py) and its inputs/outputs/intermediate
values are recorded.  Intermediates are read from the live `align` frame with a
`sys.settrace` line hook keyed on v2.0.8 line numbers (describealign.py:635, :674, :726,
:860, :946 and the return event).

  python tests/golden/make_golden.py [case ...]

`matches:<case>` records the stage-2 matches of every 64th audio frame of a config-sized pair (MatchSampleProbe).
`stretch:<case>` records replace_aligned_segments (describealign.py:230-416) on the
STRETCH_CASES inputs: the float16 bits of every replaced interval and each jump schedule.

Outputs tests/golden/*.npz (+ report text for e180) and tests/golden/index.json.
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
import time
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

for _m in ("ffmpeg", "static_ffmpeg", "natsort"):   # I/O-only imports of the reference, absent here
  sys.modules.setdefault(_m, types.ModuleType(_m))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
import describealign as ref  # noqa: E402

import cases  # noqa: E402

assert ref.__version__ == "2.0.8", ref.__version__
REF_FILE = os.path.abspath(ref.__file__)


def ref_features(pcm_i16: np.ndarray):
  arr = pcm_i16.astype(np.float16)                  # describealign.py:156
  return [ref.get_energy(arr), ref.get_zero_crossings(arr)] + ref.get_freq_bands(arr)


def gen_features():
  out = {}
  meta = {}
  for name in cases.FEATURE_CLIPS:
    pcm = cases.feature_clip(name)
    feats = ref_features(pcm)
    for k, f in enumerate(feats):
      out[f"{name}.f{k}"] = np.asarray(f)            # keep the reference's dtype (f32 / f64 for band 2)
    meta[name] = dict(sha1=cases.sha1_of(pcm), shape=list(pcm.shape),
                      lengths=[int(len(f)) for f in feats], dtypes=[str(f.dtype) for f in feats])
  np.savez_compressed(os.path.join(HERE, "features.npz"), **out)
  return meta


class AlignProbe:
  """Line hook on the reference's align() frame (v2.0.8 line numbers)."""

  def __init__(self):
    self.cap = {}
    self.cand_i, self.cand_v = [], []
    self.m_i, self.m_v, self.m_q = [], [], []

  def _global(self, frame, event, arg):
    if event == "call" and frame.f_code.co_name == "align" and frame.f_code.co_filename == REF_FILE:
      return self._local
    return None

  def _local(self, frame, event, arg):
    L = frame.f_locals
    if event == "line":
      ln = frame.f_lineno
      if ln == 674:                                   # candidates voted + verified for audio frame i
        i = L["i"]
        for v in sorted(L["common"]):
          self.cand_i.append(i); self.cand_v.append(v)
        for v, q in sorted(L["match_points"]):
          self.m_i.append(i); self.m_v.append(v); self.m_q.append(q)
      elif ln == 635 and "ms_v" not in self.cap:
        self.cap["ms_v"] = np.array(L["video_features_mean_sub"][0:5], dtype=object)
        self.cap["ms_v"] = [np.array(a, dtype=np.float64) for a in L["video_features_mean_sub"]]
        self.cap["ms_a"] = [np.array(a, dtype=np.float64) for a in L["audio_desc_features_mean_sub"]]
        self.cap["nrm_v"] = [np.array(a, dtype=np.float64) for a in L["video_uniform_norms"]]
        self.cap["nrm_a"] = [np.array(a, dtype=np.float64) for a in L["audio_desc_uniform_norms"]]
      elif ln == 726 and "p1_x" not in self.cap:
        self.cap["p1_x"] = np.array(L["x"]); self.cap["p1_y"] = np.array(L["y"])
      elif ln == 860 and "lp_x" not in self.cap:
        self.cap["lp_x"] = np.array(L["x"], dtype=np.float64)
        self.cap["lp_y"] = np.array(L["y"], dtype=np.float64)
        self.cap["lp_sol"] = np.array(L["fit"].x)
        self.cap["lp_fun"] = float(L["fit"].fun)
        self.cap["lp_c"] = np.array(L["c"])
        A = L["A_eq"].tocsc(); A.sort_indices()
        self.cap["lp_A_data"], self.cap["lp_A_indices"], self.cap["lp_A_indptr"] = A.data, A.indices, A.indptr
        self.cap["lp_A_shape"] = np.array(A.shape)
        self.cap["lp_b"] = np.array(L["b_eq"])
        self.cap["lp_cont_err"] = np.array(L["continuity_err"])
        self.cap["fit_err"] = np.array(L["fit_err"])
        self.cap["slopes"] = np.array(L["slopes"])
        self.cap["median_slope"] = float(L["median_slope"])
        self.cap["smooth_path"] = np.array(L["smooth_path"], dtype=np.float64)
        self.cap["a_scaled"] = np.array(L["audio_desc_features_scaled"], dtype=np.float64)
        self.cap["v_scaled"] = np.array(L["video_features_scaled"], dtype=np.float64)
      elif ln == 946 and "cl_offset" not in self.cap:
        lc = L["line_clusters"]
        self.cap["cl_x0"] = np.array([c[0][0] for c in lc], dtype=np.float64)
        self.cap["cl_x1"] = np.array([c[0][-1] for c in lc], dtype=np.float64)
        self.cap["cl_offset"] = np.array([c[1] for c in lc], dtype=np.float64)
        self.cap["cl_slope"] = np.array([c[2] for c in lc], dtype=np.float64)
        pi, pj, pc, pq = [], [], [], []
        for i, pts in enumerate(L["points"]):
          for (j, c, q) in pts:
            pi.append(i); pj.append(j); pc.append(c); pq.append(q)
        self.cap["pt_i"] = np.array(pi, dtype=np.int64); self.cap["pt_j"] = np.array(pj, dtype=np.float64)
        self.cap["pt_c"] = np.array(pc, dtype=np.int64); self.cap["pt_q"] = np.array(pq, dtype=np.float64)
    elif event == "return":
      if "path" in L and isinstance(L["path"], np.ndarray):
        self.cap["path2"] = np.array(L["path"])       # already divided by 210 (describealign.py:1026)
    return self._local

  def run(self, vf, af):
    sys.settrace(self._global)
    try:
      res = ref.align(vf, af, vf[0], af[0])
    finally:
      sys.settrace(None)
    c = self.cap
    for k in ("ms_v", "ms_a", "nrm_v", "nrm_a"):
      for j, a in enumerate(c.pop(k)):
        c[f"{k}{j}"] = a
    c["cand_i"] = np.array(self.cand_i, dtype=np.int32); c["cand_v"] = np.array(self.cand_v, dtype=np.int32)
    c["m_i"] = np.array(self.m_i, dtype=np.int32); c["m_v"] = np.array(self.m_v, dtype=np.int32)
    c["m_q"] = np.array(self.m_q, dtype=np.float64)
    return res, c


def gen_align(name: str, deep: bool):
  t0 = time.time()
  if name == "mismatch":
    pair = cases.align_case(name)
    vf, af = ref_features(pair.video), ref_features(pair.audio)
    try:
      ref.align(vf, af, vf[0], af[0])
      err = None
    except RuntimeError as e:
      err = str(e)
    print(f"[{name}] error={err!r}")
    return dict(sha1=pair.sha1(), error=err), None
  pair = cases.align_case(name)
  vf, af = ref_features(pair.video), ref_features(pair.audio)
  cap = {}
  if deep:
    (x, y, sim, path, med), cap = AlignProbe().run(vf, af)
  else:
    x, y, sim, path, med = ref.align(vf, af, vf[0], af[0])
  cap.update(dict(x=x, y=y, sim=np.float64(sim), med=np.float64(med),
                  path_rows=np.int64(len(path)), path20=np.array(path[::20])))
  if deep:      # the reference's own feature rows, so align stages can be pinned independently of features
    for k, (fv, fa) in enumerate(zip(vf, af)):
      cap[f"vf{k}"] = np.asarray(fv, dtype=np.float32 if k < 4 else np.float64)
      cap[f"af{k}"] = np.asarray(fa, dtype=np.float32 if k < 4 else np.float64)
  meta = dict(sha1=pair.sha1(), video_shape=list(pair.video.shape), audio_shape=list(pair.audio.shape),
              jump_video_times=pair.jump_video_times, jump_lengths=pair.jump_lengths,
              rate_change=pair.rate_change, n_nodes=int(len(x)), sim=float(sim), med=float(med),
              seconds_reference_align=None)
  if name == "e180":
    # text report + setts expression (describealign.py:159-227, :419-435)
    with tempfile.TemporaryDirectory() as td:
      stem = os.path.join(td, "e180")
      video_offset = y[0] - x[0]
      setts = ref.encode_fit_as_ffmpeg_expr(x, y, video_offset)
      ref.plot_alignment(stem, path.copy(), x, y, sim, med, False, False, "<ffmpeg command>")
      with open(stem + ".txt") as f:
        meta["report_txt"] = f.read()
      meta["setts"] = setts
      meta["png_bytes"] = os.path.getsize(stem + ".png")
  np.savez_compressed(os.path.join(HERE, f"align_{name}.npz"), **cap)
  meta["seconds_reference_total"] = round(time.time() - t0, 2)
  print(f"[{name}] nodes x={np.round(x, 4)} y={np.round(y, 4)} sim={sim:.3f} med={med:.6f} "
        f"rows={len(path)} ({meta['seconds_reference_total']} s)")
  return meta, cap


class MatchSampleProbe:
  """Stage-2 matches of every `every`-th audio frame of the reference's align() (line hook at describealign.py:674, where
  `match_points` of audio frame i is complete), plus the total number of matches: a config-sized stage-2 fixture that stays
  small (the full list of a 22-minute pair is 2e6 matches)."""

  def __init__(self, every):
    self.every = every
    self.keys, self.quals, self.total, self.rows = [], [], 0, 0

  def _global(self, frame, event, arg):
    if event == "call" and frame.f_code.co_name == "align" and frame.f_code.co_filename == REF_FILE:
      return self._local
    return None

  def _local(self, frame, event, arg):
    if event == "line" and frame.f_lineno == 674:
      L = frame.f_locals
      i = L["i"]; mp = L["match_points"]
      self.total += len(mp); self.rows += 1
      if i % self.every == 0:
        for v, q in sorted(mp):
          self.keys.append((int(i) << 32) | int(v)); self.quals.append(float(q))
    return self._local

  def run(self, vf, af):
    sys.settrace(self._global)
    try:
      res = ref.align(vf, af, vf[0], af[0])
    finally:
      sys.settrace(None)
    return res


def gen_match_sample(name: str, every: int = 64):
  """`matches:<case>`: sampled stage-2 matches of a config-sized pair (see MatchSampleProbe)."""
  t0 = time.time()
  pair = cases.align_case(name)
  vf, af = ref_features(pair.video), ref_features(pair.audio)
  probe = MatchSampleProbe(every)
  x, y, sim, path, med = probe.run(vf, af)
  np.savez_compressed(os.path.join(HERE, f"matches_{name}.npz"), keys=np.array(probe.keys, dtype=np.int64),
                      quals=np.array(probe.quals, dtype=np.float64), total=np.int64(probe.total), rows=np.int64(probe.rows),
                      every=np.int64(every), x=x, y=y)
  print(f"[matches:{name}] {probe.total} matches over {probe.rows} audio frames, {len(probe.keys)} recorded (every {every}th frame), "
        f"{time.time() - t0:.0f} s")
  return dict(sha1=pair.sha1(), total=int(probe.total), recorded=len(probe.keys), every=every)


class StretchProbe:
  """Records the jump schedule of every call of the reference's nested `stretch`
  (describealign.py:298-385) from its frame at return."""

  def __init__(self):
    self.schedules = []

  def _global(self, frame, event, arg):
    if event == "call" and frame.f_code.co_name == "stretch" and frame.f_code.co_filename == REF_FILE:
      return self._local
    return None

  def _local(self, frame, event, arg):
    if event == "return":
      L = frame.f_locals
      self.schedules.append(np.stack([np.asarray(L["jump_input_indices"]), np.asarray(L["jump_distances"])], axis=1)
                            .astype(np.int64))
    return self._local


def gen_stretch(name):
  """replace_aligned_segments (describealign.py:230-416) on a STRETCH_CASES input: records the
  float16 bit patterns of every replaced interval and the jump schedules."""
  t0 = time.time()
  v, a, x, y = cases.stretch_case_f16(name)
  before = v.copy()
  probe = StretchProbe()
  sys.settrace(probe._global)
  try:
    ref.replace_aligned_segments(v, a, x, y, False)
  finally:
    sys.settrace(None)
  print()
  ys = (y * ref.AUDIO_SAMPLE_RATE).astype(int)
  out = {}
  kept = []
  for k in range(len(ys) - 1):
    sl = slice(int(ys[k]), int(ys[k + 1]))
    if not np.array_equal(v[:, sl].view(np.uint16), before[:, sl].view(np.uint16)):
      out[f"seg{k}"] = v[:, sl].view(np.uint16).copy()
      kept.append(k)
  for k, s in enumerate(probe.schedules):
    out[f"sched{k}"] = s
  untouched = np.ones(v.shape[1], dtype=bool)
  for k in kept:
    untouched[int(ys[k]):int(ys[k + 1])] = False
  assert np.array_equal(v[:, untouched].view(np.uint16), before[:, untouched].view(np.uint16))
  # the same input with no_pitch_correction=True (every kept interval is resampled)
  v2 = before.copy()
  ref.replace_aligned_segments(v2, a, x, y, True)
  print()
  out["npc_sha1"] = np.frombuffer(bytes.fromhex(cases.sha1_of(v2.view(np.uint16))), dtype=np.uint8)
  np.savez_compressed(os.path.join(HERE, f"stretch_{name}.npz"), **out)
  meta = dict(sha1_inputs=cases.sha1_of(before.view(np.uint16), a.view(np.uint16)), replaced_intervals=kept,
              n_schedules=len(probe.schedules), jumps_per_schedule=[int(len(s)) for s in probe.schedules],
              sha1_output=cases.sha1_of(v.view(np.uint16)), seconds_reference=round(time.time() - t0, 2))
  print(f"[stretch {name}] replaced={kept} schedules={meta['jumps_per_schedule']} ({meta['seconds_reference']} s)")
  return meta


def gen_combine_stretch(name):
  """The reference's combine(..., stretch_audio=True) itself (describealign.py:1031-1175) on a
  synthetic stereo pair: only its file I/O is replaced -- get_sorted_filenames,
  parse_audio_from_file (returns the synthetic PCM as float16, what :156 produces) and
  write_replaced_media_to_disk (records the array it is handed) -- so the loudness matching and peak
  normalisation written inline in combine() (:1135-1153) run unmodified together with
  get_energy/.../align/replace_aligned_segments.  Records sha1 of the float16 array handed to the
  writer and of its int16 serialisation (:136), a sparse sample of both, and the nodes."""
  t0 = time.time()
  pair = cases.combine_stretch_case(name)
  pcm = {"video.wav": pair.video, "ad.wav": pair.audio}
  captured = {}
  saved = {k: getattr(ref, k) for k in ("get_sorted_filenames", "parse_audio_from_file", "write_replaced_media_to_disk",
                                        "is_ffmpeg_installed", "replace_aligned_segments")}
  def fake_sorted(path, extensions, alt_extensions=set([])):
    return [path], [os.path.splitext(path)[1][1:] in alt_extensions]
  def fake_parse(media_file, num_channels=2):
    arr = pcm[os.path.basename(media_file)]
    assert arr.shape[0] == num_channels
    return arr.astype(np.float16)
  def fake_write(output_filename, media_arr, video_file=None, **kw):
    captured["media"] = media_arr.copy(); captured["video_file"] = video_file
    return "<ffmpeg command>"
  def spy_replace(video_arr, audio_desc_arr, audio_desc_times, video_times, no_pitch_correction):
    captured["x"] = np.array(audio_desc_times); captured["y"] = np.array(video_times)
    captured["scaled_video_sha1"] = cases.sha1_of(video_arr.view(np.uint16)); captured["scaled_audio_sha1"] = cases.sha1_of(audio_desc_arr.view(np.uint16))
    return saved["replace_aligned_segments"](video_arr, audio_desc_arr, audio_desc_times, video_times, no_pitch_correction)
  ref.get_sorted_filenames = fake_sorted; ref.parse_audio_from_file = fake_parse
  ref.write_replaced_media_to_disk = fake_write; ref.is_ffmpeg_installed = lambda *a, **k: True
  ref.replace_aligned_segments = spy_replace
  try:
    with tempfile.TemporaryDirectory() as td:
      ref.combine(os.path.join(td, "video.wav"), os.path.join(td, "ad.wav"), stretch_audio=True, yes=True,
                  output_dir=os.path.join(td, "out"), alignment_dir=os.path.join(td, "plots"))
      report = open(os.path.join(td, "plots", "video.txt")).read()
  finally:
    for k, v in saved.items():
      setattr(ref, k, v)
  print()
  media = captured["media"]
  s16 = media.astype(np.int16).T                        # describealign.py:136
  out = dict(x=captured["x"], y=captured["y"], media_every_997=media[:, ::997].view(np.uint16).copy(),
             s16_every_997=np.ascontiguousarray(s16[::997]))
  np.savez_compressed(os.path.join(HERE, f"combine_stretch_{name}.npz"), **out)
  meta = dict(sha1_inputs=pair.sha1(), media_shape=list(media.shape), media_dtype=str(media.dtype),
              sha1_media_f16=cases.sha1_of(media.view(np.uint16)), sha1_s16le=cases.sha1_of(s16),
              sha1_scaled_video=captured["scaled_video_sha1"], sha1_scaled_audio=captured["scaled_audio_sha1"],
              peak_int16=[int(s16.min()), int(s16.max())], report_first_lines=report.split("\n")[:8],
              seconds_reference=round(time.time() - t0, 2))
  print(f"[combine_stretch {name}] nodes {len(captured['x'])} peak {meta['peak_int16']} ({meta['seconds_reference']} s)")
  return meta


def gen_commands():
  """The command lines the reference hands to ffmpeg / ffprobe (describealign.py:149-153 decode, :443-449 and :460-462 probes,
  :464-515 default mux and --stretch_audio mux), recorded by running the reference's own functions against
  tests/golden/ffmpeg_python_double.py (ffmpeg-python is not installable here; the double restates its compile rules).
  -> tests/golden/commands.json: [{name, args, argv | probe_argv | logged}]"""
  import ffmpeg_python_double as dbl
  saved = {k: getattr(ref, k) for k in ("ffmpeg", "get_ffmpeg", "get_ffprobe")}
  ref.ffmpeg = dbl; ref.get_ffmpeg = lambda: "ffmpeg"; ref.get_ffprobe = lambda: "ffprobe"
  out = []
  try:
    def take():
      rec = list(dbl.RECORDED); dbl.RECORDED.clear(); return rec
    # decode (:149-157): the double returns no bytes, the reshape of an empty stream is fine
    for ch in (1, 2):
      take()
      ref.parse_audio_from_file("/media/show.mkv", ch)
      out.append(dict(name=f"decode_{ch}ch", args=dict(media_file="/media/show.mkv", num_channels=ch), argv=take()[0][1]))
    # probes (:443-449, :460-462)
    dbl.PROBE_RESULT = {"frames": [{"pts_time": "0.000000"}, {"pts_time": "10.010000"}, {"pkt_pts_time": "3"}]}
    for t in (None, 12.5, 300.0):
      take()
      times = ref.get_key_frame_data("/media/show.mkv", t)
      out.append(dict(name=f"key_frames_{t}", args=dict(video_file="/media/show.mkv", time=t), argv=take()[0][1], returned=[float(x) for x in times]))
    dbl.PROBE_RESULT = {"streams": [{"disposition": {"descriptions": 0, "visual_impaired": 1}}]}
    take()
    ad = ref.is_first_video_track_ad("/media/show.mkv")
    out.append(dict(name="first_track_is_ad", args=dict(video_file="/media/show.mkv"), argv=take()[0][1], returned=bool(ad)))
    # default mux (:489-515): extensions decide codec / strictness; offsets of either sign; a rate change
    setts = "if(lt(PTS,10),PTS*1.000000+0.500000,PTS*0.990000+1.250000)"
    for name, ad_file, off, key, slope in (("mux_mp3_late", "/media/ad.mp3", 3.25, 1.5, 1.0), ("mux_wav_early", "/media/ad.wav", 0.75, 2.0, 1.0),
                                           ("mux_flac_rate", "/media/ad.flac", 12.0, 11.123456, 1.0004), ("mux_m4a_zero", "/media/ad.m4a", 0.0, 0.0, 0.98)):
      take()
      logged = ref.write_replaced_media_to_disk("/out/ad_show.mkv", None, "/media/show.mkv", ad_file, setts, off, key, slope)
      out.append(dict(name=name, args=dict(output_filename="/out/ad_show.mkv", video_file="/media/show.mkv", audio_desc_file=ad_file, setts_cmd=setts,
                                           video_offset=off, after_start_key_frame=key, median_slope=slope), argv=take()[0][1], logged=logged))
    # --stretch_audio mux (:468-488): with a video (first track AD or not) and audio only
    media_arr = np.zeros((2, 441), dtype=np.float16)
    for name, video, first_ad in (("stretch_video_first_original", "/media/show.mkv", 0), ("stretch_video_first_ad", "/media/described_show.mkv", 1),
                                  ("stretch_audio_only", None, 0)):
      dbl.PROBE_RESULT = {"streams": [{"disposition": {"descriptions": first_ad, "visual_impaired": 0}}]}
      take()
      logged = ref.write_replaced_media_to_disk("/out/ad_show.mkv", media_arr, video)
      rec = take()
      out.append(dict(name=name, args=dict(output_filename="/out/ad_show.mkv", video_file=video, first_track_is_ad=bool(first_ad)),
                      argv=[a for k, a in rec if k == "run_async"][0], probe_argv=[a for k, a in rec if k == "probe"],
                      stdin_bytes=[a for k, a in rec if k == "stdin_bytes"][0], logged=logged))
  finally:
    for k, v in saved.items():
      setattr(ref, k, v)
  json.dump(out, open(os.path.join(HERE, "commands.json"), "w"), indent=1)
  print(f"[commands] {len(out)} command lines recorded")
  return dict(count=len(out), how="reference functions run against tests/golden/ffmpeg_python_double.py (ffmpeg-python 0.2.0's compile rules restated)")


def main(argv):
  idx_path = os.path.join(HERE, "index.json")
  index = json.load(open(idx_path)) if os.path.exists(idx_path) else {}
  want = argv or (["features", "a40", "e180", "e180s", "e600", "rate2", "rateneg600", "j600s", "mismatch", "e1320", "e1800", "rate1800", "j1800"] +
                  ["stretch:" + n for n in cases.STRETCH_CASES] +
                  ["combine_stretch:" + n for n in cases.COMBINE_STRETCH_CASES])
  index["reference_version"] = ref.__version__
  index["numpy"] = np.__version__
  import scipy
  index["scipy"] = scipy.__version__
  for name in want:
    if name == "features":
      index["features"] = gen_features()
      print("[features] done")
    elif name == "commands":
      index["commands"] = gen_commands()
    elif name.startswith("combine_stretch:"):
      index.setdefault("combine_stretch", {})[name[16:]] = gen_combine_stretch(name[16:])
    elif name.startswith("stretch:"):
      index.setdefault("stretch", {})[name[8:]] = gen_stretch(name[8:])
    elif name.startswith("matches:"):
      index.setdefault("matches", {})[name[8:]] = gen_match_sample(name[8:])
    else:
      meta, _ = gen_align(name, deep=(name == "a40"))
      index.setdefault("align", {})[name] = meta
    json.dump(index, open(idx_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
  main(sys.argv[1:])
