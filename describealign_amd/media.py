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


def _downmix(pcm: np.ndarray, num_channels: int) -> np.ndarray:
  c = pcm.shape[0]
  if c == num_channels:
    return pcm
  if num_channels == 1:
    return np.clip(np.round(pcm.astype(np.float32).mean(axis=0)), -32768, 32767).astype(np.int16)[None, :]
  if c == 1:
    return np.repeat(pcm, 2, axis=0)
  return pcm[:num_channels]


def parse_audio_from_file(media_file, num_channels=2) -> np.ndarray:
  ext = os.path.splitext(media_file)[1].lower()
  if ext == ".wav":
    with wave.open(media_file, "rb") as w:
      if w.getframerate() == AUDIO_SAMPLE_RATE and w.getsampwidth() == 2:
        raw = w.readframes(w.getnframes())
        pcm = np.frombuffer(raw, dtype="<i2").reshape(-1, w.getnchannels()).T
        return np.ascontiguousarray(_downmix(pcm, num_channels))
  if ext in (".raw", ".s16le", ".pcm"):
    pcm = np.fromfile(media_file, dtype="<i2")
    return np.ascontiguousarray(pcm.reshape(1, -1) if num_channels == 1 else pcm.reshape(-1, 2).T)
  exe = find_ffmpeg()
  if exe is None:
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
  return np.ascontiguousarray(np.frombuffer(res.stdout, np.int16).reshape((-1, num_channels)).T)


def write_wav(path, pcm: np.ndarray):
  pcm = np.asarray(pcm, dtype=np.int16)
  with wave.open(path, "wb") as w:
    w.setnchannels(pcm.shape[0]); w.setsampwidth(2); w.setframerate(AUDIO_SAMPLE_RATE)
    w.writeframes(np.ascontiguousarray(pcm.T).tobytes())
