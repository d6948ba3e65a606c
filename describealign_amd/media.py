"""Minimal PCM ingest for the alignment path.

The reference decodes everything through an ffmpeg child process (describealign.py:149-157).
SURVEY section 8(f) item 1: this module reads 44.1 kHz 16-bit WAV / raw s16le directly and runs an `ffmpeg`
binary with the reference's decode arguments for everything else, reading its pipe with readinto() into
page-locked memory: either a ring of pieces whose host->device copies are enqueued as they fill
(stream_file_to_device: no whole-file host buffer), or one whole-file buffer (parse_audio_from_file, the
reference's interface: int16 (C, N) -- the integer values the reference then holds as float16).
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


def _ffmpeg_decode_command(exe, media_file, num_channels):
  """The reference's decode (describealign.py:149-153): first audio stream, s16le at 44.1 kHz,
  aresample=async=1:first_pts=0, downmix to num_channels."""
  # options in the order the reference's ffmpeg-python graph compiles them (sorted by name; tests/golden/commands.json)
  return [exe, "-i", media_file, "-f", "s16le", "-ac", str(num_channels), "-acodec", "pcm_s16le", "-af", "aresample=async=1:first_pts=0",
          "-ar", str(AUDIO_SAMPLE_RATE), "-loglevel", "error", "-map", "0:a:0", "-"]


def _wav_pcm_span(media_file, num_channels):
  """(byte offset, byte length) of the sample data of a 44.1 kHz 16-bit PCM WAV with num_channels channels, or None
  when the file is anything else (those go through ffmpeg, see _read_native)."""
  try:
    with wave.open(media_file, "rb") as w:
      if w.getframerate() != AUDIO_SAMPLE_RATE or w.getsampwidth() != 2 or w.getnchannels() != num_channels or w.getcomptype() != "NONE":
        return None
      nbytes = w.getnframes() * 2 * num_channels
      w.setpos(0)
      # the wave module has parsed the chunks and stands at the first frame: the underlying file's position is the data offset
      offset = w.getfp().file.tell() if hasattr(w.getfp(), "file") else None
      if offset is not None:
        # a streamed WAV (`ffmpeg -f wav -`) declares a data size of 0xFFFFFFFF: never more than the file holds, whole frames
        have = max(0, os.path.getsize(media_file) - offset)
        nbytes = min(nbytes, have - have % (2 * num_channels))
      return offset, nbytes
  except (wave.Error, EOFError, OSError, ValueError, AttributeError):
    return None


class PcmSource:
  """Interleaved s16le frames of a media file as a byte stream read with readinto() -- straight from the decoder's
  pipe (or from the file itself for 44.1 kHz 16-bit WAV / raw s16le with the wanted channel count), never through
  an intermediate `bytes`.  frames_hint: the number of frames when it is known up front (0 otherwise).
  close() raises the reference's "FFmpeg error." when the decoder failed."""

  def __init__(self, media_file, num_channels):
    self.media_file, self.num_channels = media_file, num_channels
    self.frames_hint = 0
    self._proc = self._file = self._errlog = None
    self._left = None
    ext = os.path.splitext(media_file)[1].lower()
    if ext == ".wav":
      span = _wav_pcm_span(media_file, num_channels)
      if span is not None and span[0] is not None:
        self._file = open(media_file, "rb", buffering=0)
        self._file.seek(span[0]); self._left = span[1]
        self.frames_hint = span[1] // (2 * num_channels)
        return
    elif ext in (".raw", ".s16le", ".pcm"):
      size = os.path.getsize(media_file)
      if size % (2 * num_channels) == 0:
        self._file = open(media_file, "rb", buffering=0); self._left = size
        self.frames_hint = size // (2 * num_channels)
        return
    exe = find_ffmpeg()
    if exe is None:
      raise RuntimeError(f"cannot decode {media_file}: no ffmpeg binary on PATH "
                         "(only 44.1 kHz 16-bit .wav and raw s16le are read natively)")
    import tempfile
    self._errlog = tempfile.TemporaryFile()          # stderr to a file: a full pipe nobody reads would stall the decoder
    self._proc = subprocess.Popen(_ffmpeg_decode_command(exe, media_file, num_channels), stdout=subprocess.PIPE,
                                  stderr=self._errlog, stdin=subprocess.DEVNULL, bufsize=0)

  def readinto(self, view) -> int:
    """Fill `view` (a writable bytes-like object) as far as the stream goes; returns the bytes written (short only at the end)."""
    mv = memoryview(view).cast("B")
    got = 0
    src = self._proc.stdout if self._proc is not None else self._file
    while got < len(mv):
      want = len(mv) - got
      if self._left is not None:
        want = min(want, self._left)
        if want == 0:
          break
      k = src.readinto(mv[got:got + want])
      if not k:
        break
      got += k
      if self._left is not None:
        self._left -= k
    return got

  def close(self):
    if self._file is not None:
      self._file.close(); self._file = None
    if self._proc is not None:
      proc, self._proc = self._proc, None
      proc.stdout.close()
      rc = proc.wait()
      self._errlog.seek(0); err = self._errlog.read(); self._errlog.close()
      if rc != 0:
        print("  ERROR: ffmpeg failed to parse audio from input file: " + self.media_file)
        print("FFmpeg error:")
        print(err.decode("utf-8", "replace"))
        raise RuntimeError("FFmpeg error.")

  def __enter__(self):
    return self

  def __exit__(self, et, ev, tb):
    if et is None:
      self.close()
    else:                                            # already failing: do not mask the first error
      try:
        self.close()
      except RuntimeError:
        pass


PIECE_BYTES = 64 << 20


def stream_file_to_device(stream, media_file, num_channels, ring=None, piece_bytes=PIECE_BYTES) -> int:
  """parse_audio_from_file (:149-157) without the array: decoder pipe -> a ring of page-locked pieces -> HBM
  (`stream`: _native.PcmStream).  Each piece's host->device copy is enqueued as soon as the piece is full, so decode
  and transfer of one file overlap and the host never holds more than the ring (3 x 64 MiB by default, against
  1.27 GB for a 2 h stereo side).  Returns the frames uploaded; hand the stream to Context.pcm_adopt."""
  from . import _native
  fb = 2 * num_channels
  piece_bytes = max(fb, piece_bytes // fb * fb)
  if ring is None:
    ring = [_native.pinned_empty((piece_bytes // 2,), np.int16) for _ in range(3)]
  k = 0
  ext = os.path.splitext(media_file)[1].lower()
  if find_ffmpeg() is None and ext == ".wav" and _wav_pcm_span(media_file, num_channels) is None:
    # no decoder at all and the WAV has the other channel count: mixed on the host (parse_audio_from_file), then streamed
    pcm = parse_audio_from_file(media_file, num_channels)
    frames = np.ascontiguousarray(pcm.T)
    step = piece_bytes // fb
    for at in range(0, len(frames), step):
      stream.piece(frames[at:at + step])
    stream.sync()
    return stream.frames
  with PcmSource(media_file, num_channels) as src:
    while True:
      buf = ring[k % len(ring)]
      if k >= len(ring):
        stream.sync()                                # the copies enqueued so far have left the ring
      got = src.readinto(buf)
      got -= got % fb                                # a torn last frame is dropped
      if got:
        stream.piece(buf[:got // 2].reshape(-1, num_channels))
      if got < buf.nbytes:
        break
      k += 1
  stream.sync()
  return stream.frames


def parse_audio_from_file(media_file, num_channels=2, alloc=None) -> np.ndarray:
  """describealign.parse_audio_from_file (:149-157) returning the int16 values as int16 (C, N).
  alloc: optional allocator shape -> int16 array the result is written into (combine hands out
  page-locked buffers, so the upload of this file overlaps the kernels of the previous pair).
  The frames are read from the decoder's pipe straight into that buffer (readinto, no intermediate bytes
  object) and stay interleaved: the (C, N) array returned is the transposed VIEW of the (N, C) frames, which
  the upload calls recognise and copy as they are."""
  ext = os.path.splitext(media_file)[1].lower()
  if find_ffmpeg() is None and ext == ".wav" and _wav_pcm_span(media_file, num_channels) is None:
    # no decoder at all: 1 <-> 2 channel WAVs are still usable
    pcm = _read_wav_any_channels(media_file)
    pcm = None if pcm is None else _downmix_like_swresample(pcm, num_channels)
    if pcm is not None:
      return _deliver(pcm, alloc)
  fb = 2 * num_channels
  make = alloc if alloc is not None else (lambda shape: np.empty(shape, dtype=np.int16))
  with PcmSource(media_file, num_channels) as src:
    if src.frames_hint:                              # length known: one buffer, one pass
      out = make((src.frames_hint, num_channels))
      got = src.readinto(out)
      return out[:got // fb].T
    # length unknown (a decoder pipe): pieces as they come, then one gathering pass into a buffer of the final size
    pieces, total = [], 0
    while True:
      buf = np.empty((PIECE_BYTES // fb, num_channels), dtype=np.int16)
      got = src.readinto(buf)
      got -= got % fb
      if got:
        pieces.append(buf[:got // fb]); total += got // fb
      if got < buf.nbytes:
        break
  out = make((total, num_channels))
  at = 0
  while pieces:
    p = pieces.pop(0)
    out[at:at + len(p)] = p; at += len(p)
    del p                                            # a piece is given back as soon as it has been copied: peak = total + one piece
  return out.T


def write_wav(path, pcm: np.ndarray):
  pcm = np.asarray(pcm, dtype=np.int16)
  with wave.open(path, "wb") as w:
    w.setnchannels(pcm.shape[0]); w.setsampwidth(2); w.setframerate(AUDIO_SAMPLE_RATE)
    w.writeframes(np.ascontiguousarray(pcm.T).tobytes())
