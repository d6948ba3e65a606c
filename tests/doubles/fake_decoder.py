"""Test double for the `ffmpeg` binary on the decode path (describealign.py:149-153 runs
`ffmpeg -i <file> -f s16le -acodec pcm_s16le -af ... -map 0:a:0 -ac <C> -ar 44100 -loglevel error -`).
It understands exactly that command line: reads the 16-bit WAV container behind -i (whatever the file is called),
mixes to the requested channel count the way swresample's int16 path does ((L + R + 1) >> 1 / duplicate) and writes
the interleaved s16le frames to stdout in small writes, like a decoder pipe.  A file whose name contains "broken"
makes it fail with a message on stderr.  Not a media decoder; it exists so that the pipe -> page-locked ring -> HBM
path can run in an image without ffmpeg.  tests install it as <tmp>/ffmpeg (install() below)."""
import os
import stat
import sys
import wave


def install(directory) -> str:
  """<directory>/ffmpeg: an executable wrapper running this file with the current interpreter."""
  path = os.path.join(str(directory), "ffmpeg")
  with open(path, "w") as f:
    f.write(f"#!/bin/sh\nexec {sys.executable} {os.path.abspath(__file__)} \"$@\"\n")
  os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
  return path


def main(argv):
  src = argv[argv.index("-i") + 1]
  want = int(argv[argv.index("-ac") + 1])
  assert argv[argv.index("-f") + 1] == "s16le" and argv[argv.index("-ar") + 1] == "44100" and argv[-1] == "-", argv
  if "broken" in os.path.basename(src):
    sys.stderr.write(f"{src}: Invalid data found when processing input\n")
    return 1
  import numpy as np
  out = sys.stdout.buffer
  with wave.open(src, "rb") as w:
    assert w.getsampwidth() == 2 and w.getframerate() == 44100
    have = w.getnchannels()
    left = w.getnframes()
    while left > 0:
      n = min(left, 37813)                       # odd-sized writes: pieces never line up with the reader's buffers
      fr = np.frombuffer(w.readframes(n), dtype="<i2").reshape(-1, have)
      if want == 1 and have == 2:
        fr = ((fr[:, 0].astype(np.int32) + fr[:, 1].astype(np.int32) + 1) >> 1).astype("<i2")[:, None]
      elif want == 2 and have == 1:
        fr = np.repeat(fr, 2, axis=1)
      out.write(np.ascontiguousarray(fr).tobytes())
      left -= n
  out.flush()
  return 0


if __name__ == "__main__":
  sys.exit(main(sys.argv[1:]))
