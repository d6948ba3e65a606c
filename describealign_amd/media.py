"""Minimal PCM ingest for the alignment path.

The reference decodes everything through an ffmpeg child process (describealign.py:149-157).
Media I/O is outside the accelerated path: this module reads
44.1 kHz 16-bit WAV / raw s16le directly, and shells out to an `ffmpeg` binary with the
reference's decode arguments when one is on PATH.  It returns int16 (C, N) -- the integer
values the reference then holds as float16.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import wave

import numpy as np

AUDIO_SAMPLE_RATE = 44100


def find_ffmpeg():
  return shutil.which("ffmpeg")


def find_ffprobe():
  exe = shutil.which("ffprobe")
  if exe is None and find_ffmpeg() is not None:          # static builds ship the two side by side
    cand = os.path.join(os.path.dirname(find_ffmpeg()), "ffprobe")
    exe = cand if os.path.isfile(cand) else None
  return exe


def _deliver(planar_view, alloc):
  """The decoded (C, N) int16 array, C-contiguous: written into `alloc((C, N))` when the caller provides
  an allocator (page-locked memory for an asynchronous upload), else a fresh numpy array."""
  if alloc is None:
    return np.ascontiguousarray(planar_view)
  out = alloc(planar_view.shape)
  out[...] = planar_view
  return out


def _read_native(media_file, ext, num_channels, alloc=None):
  """44.1 kHz 16-bit PCM WAV / raw s16le without ffmpeg.  Returns None when the file is not
  exactly what ffmpeg would hand back untouched (other rate / width / channel count, extensible
  or float WAV headers, ragged raw files): those go through ffmpeg, whose resampler and downmix
  the reference's results depend on (:149-157)."""
  try:
    if ext == ".wav":
      with wave.open(media_file, "rb") as w:
        if w.getframerate() != AUDIO_SAMPLE_RATE or w.getsampwidth() != 2 or w.getnchannels() != num_channels:
          return None
        raw = w.readframes(w.getnframes())
      return _deliver(np.frombuffer(raw, dtype="<i2").reshape(-1, num_channels).T, alloc)
    pcm = np.fromfile(media_file, dtype="<i2")        # raw files carry no header: taken as num_channels s16le
    if len(pcm) % num_channels:
      return None
    return _deliver(pcm.reshape(-1, num_channels).T, alloc)
  except (wave.Error, EOFError, OSError, ValueError):
    return None


def _downmix_like_swresample(pcm: np.ndarray, num_channels: int) -> np.ndarray:
  """Channel-count change for natively read PCM when no ffmpeg binary exists (test media only;
  with ffmpeg on PATH the file is decoded by ffmpeg itself).  Stereo -> mono is the average with
  the half rounded up, (L + R + 1) >> 1, which is what swresample's int16 path produces for its
  0.5 / 0.5 mix matrix; mono -> stereo duplicates the channel."""
  c = pcm.shape[0]
  if c == num_channels:
    return pcm
  if num_channels == 1 and c == 2:
    return ((pcm[0].astype(np.int32) + pcm[1].astype(np.int32) + 1) >> 1).astype(np.int16)[None, :]
  if num_channels == 2 and c == 1:
    return np.repeat(pcm, 2, axis=0)
  return None


def _read_wav_any_channels(media_file):
  try:
    with wave.open(media_file, "rb") as w:
      if w.getframerate() != AUDIO_SAMPLE_RATE or w.getsampwidth() != 2:
        return None
      raw = w.readframes(w.getnframes())
      return np.ascontiguousarray(np.frombuffer(raw, dtype="<i2").reshape(-1, w.getnchannels()).T)
  except (wave.Error, EOFError, OSError, ValueError):
    return None


def parse_audio_from_file(media_file, num_channels=2, alloc=None) -> np.ndarray:
  """describealign.parse_audio_from_file (:149-157) returning the int16 values as int16 (C, N).
  alloc: optional allocator shape -> int16 array the result is written into (combine hands out
  page-locked buffers, so the upload of this file overlaps the kernels of the previous pair)."""
  ext = os.path.splitext(media_file)[1].lower()
  if ext in (".wav", ".raw", ".s16le", ".pcm"):
    pcm = _read_native(media_file, ext, num_channels, alloc)
    if pcm is not None:
      return pcm
  exe = find_ffmpeg()
  if exe is None:
    if ext == ".wav":              # no decoder at all: 1 <-> 2 channel WAVs are still usable
      pcm = _read_wav_any_channels(media_file)
      pcm = None if pcm is None else _downmix_like_swresample(pcm, num_channels)
      if pcm is not None:
        return _deliver(pcm, alloc)
    raise RuntimeError(f"cannot decode {media_file}: no ffmpeg binary on PATH "
                       "(only 44.1 kHz 16-bit .wav and raw s16le are read natively)")
  cmd = [exe, "-i", media_file, "-f", "s16le", "-acodec", "pcm_s16le", "-af", "aresample=async=1:first_pts=0",
         "-map", "0:a:0", "-ac", str(num_channels), "-ar", str(AUDIO_SAMPLE_RATE), "-loglevel", "error", "-"]
  res = subprocess.run(cmd, capture_output=True)
  if res.returncode != 0:
    print("  ERROR: ffmpeg failed to parse audio from input file: " + media_file)
    print("FFmpeg error:")
    print(res.stderr.decode("utf-8", "replace"))
    raise RuntimeError("FFmpeg error.")
  return _deliver(np.frombuffer(res.stdout, np.int16).reshape((-1, num_channels)).T, alloc)


def write_wav(path, pcm: np.ndarray):
  pcm = np.asarray(pcm, dtype=np.int16)
  with wave.open(path, "wb") as w:
    w.setnchannels(pcm.shape[0]); w.setsampwidth(2); w.setframerate(AUDIO_SAMPLE_RATE)
    w.writeframes(np.ascontiguousarray(pcm.T).tobytes())
