"""Alignment plot + piecewise-offset text report, and the ffmpeg `setts` expression.

Host-side re-implementation of the reference's output rendering (describealign.py:159-227,
:419-435): same file names, same text lines, same expression format, so downstream tooling
that reads describealign's alignment reports keeps working.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

from . import REFERENCE_VERSION, __version__

TIMESTEP_SIZE_SECONDS = 0.1          # 1 / TIMESTEPS_PER_SECOND (:29-30)
MAX_RATE_RATIO_DIFF_ALIGN = 0.1      # (:33)


def get_version_hash(filename):
  """First 8 hex digits of the file's sha1, or "None" (:1762-1769)."""
  try:
    with open(filename, "rb") as f:
      return hashlib.sha1(f.read()).hexdigest()[:8]
  except Exception:
    return "None"


def _hms(seconds):
  minutes, seconds = divmod(seconds, 60)
  hours, minutes = divmod(minutes, 60)
  return f"{hours:2.0f}:{minutes:02.0f}:{seconds:06.3f}"


def report_lines(audio_times, video_times, similarity_percent, median_slope, stretch_audio,
                 no_pitch_correction, ffmpeg_command):
  """The lines of the .txt report (:205-227)."""
  params = {'stretch_audio': stretch_audio, 'no_pitch_correction': no_pitch_correction}
  here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "align.py")
  lines = [f"Parameters: {params}",
           f"Version: {REFERENCE_VERSION}",
           f"Script Hash: {get_version_hash(here)}",
           f"Input file similarity: {similarity_percent:.2f}%",
           "Main changes needed to video to align it to audio input:",
           f"Start Offset: {-(video_times[0] - audio_times[0]):.2f} seconds",
           f"Median Rate Change: {(median_slope - 1.) * 100:.2f}%"]
  for k in range(len(video_times) - 1):
    rate = (video_times[k + 1] - video_times[k]) / (audio_times[k + 1] - audio_times[k])
    lines.append(f"Rate change of {(rate - 1.) * 100:8.1f}% from {_hms(video_times[k])} to "
                 f"{_hms(video_times[k + 1])} aligning with audio from "
                 f"{_hms(audio_times[k])} to {_hms(audio_times[k + 1])}")
  lines += ["", "FFmpeg command:", str(ffmpeg_command)]
  return lines


def plot_alignment(plot_filename_no_ext, path, audio_times, video_times, similarity_percent,
                   median_slope, stretch_audio, no_pitch_correction, ffmpeg_command):
  """Scatter of every 20th match (opacity from its quality) with the fitted piecewise line,
  saved as <name>.png, plus the text report <name>.txt (:159-227)."""
  import matplotlib
  matplotlib.use("Agg", force=False)
  import matplotlib.pyplot as plt
  plt.switch_backend("Agg")
  sub = np.asarray(path)[::20]
  v_full, a_full, quals = sub[:, 0], sub[:, 1], sub[:, 3]
  colour = [.2, .4, .8]
  rgba = np.zeros((len(quals), 4))
  rgba[:, :3] = colour
  rgba[:, 3] = np.clip(quals * 400. / len(quals), 0, 1)
  plt.scatter(v_full / 60., a_full - v_full, s=3, c=rgba, label='Matches')
  offsets = audio_times - video_times

  def widen(lo, hi, ratio=.01):
    mid, half = (hi + lo) / 2., (hi - lo) / 2. * (1 + ratio)
    return mid - half, mid + half

  plt.xlim(widen(0, np.max(video_times) / 60.))
  plt.ylim(widen(np.min(offsets) - 10 * TIMESTEP_SIZE_SECONDS, np.max(offsets) + 10 * TIMESTEP_SIZE_SECONDS, .05))
  if stretch_audio:
    plt.plot(video_times / 60., offsets, 'r-', lw=.5, label='Replaced Audio')
    vt, at = [], []
    for k in range(len(video_times) - 1):
      slope = (audio_times[k + 1] - audio_times[k]) / (video_times[k + 1] - video_times[k])
      if abs(1 - slope) > MAX_RATE_RATIO_DIFF_ALIGN:
        vt += [video_times[k], video_times[k + 1], video_times[k + 1]]
        at += [audio_times[k], audio_times[k + 1], np.nan]
    if vt:
      vt, at = np.array(vt), np.array(at)
      plt.plot(vt / 60., at - vt, 'c-', lw=1, label='Original Audio')
  else:
    plt.plot(video_times / 60., offsets, 'r-', lw=1, label='Combined Media')
  plt.xlabel('Original Video Time (minutes)')
  plt.ylabel('Original Audio Description Offset (seconds behind video)')
  plt.title(f"Alignment - Media Similarity {similarity_percent:.2f}%")
  plt.legend().legend_handles[0].set_color(colour)
  plt.tight_layout()
  plt.savefig(plot_filename_no_ext + '.png', dpi=400)
  plt.clf()
  with open(plot_filename_no_ext + '.txt', 'w') as f:
    for line in report_lines(audio_times, video_times, similarity_percent, median_slope, stretch_audio,
                             no_pitch_correction, ffmpeg_command):
      print(line, file=f)


def encode_fit_as_ffmpeg_expr(audio_desc_times, video_times, video_offset):
  """Piecewise-linear fit as an ffmpeg `setts` timestamp expression: one clip() term per
  segment (:419-435)."""
  x = np.asarray(audio_desc_times); y = np.asarray(video_times)
  dx, dy = np.diff(x), np.diff(y)
  ratio = dx / dy
  terms = [f'+clip(TS-{y[k] - video_offset:.4f}/TB,0,{max(0, dy[k]):.4f}/TB)*{ratio[k] - 1:.9f}'
           for k in range(len(x) - 1)]
  return 'TS+(0' + ''.join(terms) + ')'
