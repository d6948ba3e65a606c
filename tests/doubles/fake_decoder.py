"""Test doubles for the `ffmpeg` / `ffprobe` binaries (this image has neither; there is no network to fetch them).

ffmpeg, decode role -- describealign.py:149-153 runs
  ffmpeg -i <file> -f s16le -acodec pcm_s16le -af ... -map 0:a:0 -ac <C> -ar 44100 -loglevel error -
The double understands exactly that command line: it reads the 16-bit WAV container behind -i (whatever the file is
called), mixes to the requested channel count the way swresample's int16 path does ((L + R + 1) >> 1 / duplicate) and
writes the interleaved s16le frames to stdout in small odd-sized writes, like a decoder pipe.  A file whose name
contains "broken" makes it fail with a message on stderr.

ffmpeg, mux role (the command lines of describealign.py:468-510: any invocation whose last arguments are
`<output> -y`) -- it consumes stdin when the input is `pipe:`, and writes a JSON record of its argv (and of the bytes
received on stdin: count and sha1) into the output file, padded beyond the 100 kB the skip-existing rule (:1087-1089)
looks for.

ffprobe -- prints the JSON the callers parse: `-select_streams V -show_frames -skip_frame nokey` -> key frames every
2.5 s with pts_time entries (inside the -read_intervals window); `-select_streams a` -> one audio stream whose
disposition marks an audio description when the file name contains "described".

Not media tools: they exist so that the pipe -> page-locked ring -> HBM path, the key-frame probe and the mux calls run
for real (subprocess, argv, stdin, files) in an image without ffmpeg.  tests install them as <tmp>/ffmpeg and
<tmp>/ffprobe (install() below)."""
import hashlib
import json
import os
import stat
import sys
import wave

KEY_FRAME_STEP = 2.5


def install(directory) -> str:
  """<directory>/ffmpeg and <directory>/ffprobe: executable wrappers running this file with the current interpreter."""
  for role in ("ffmpeg", "ffprobe"):
    path = os.path.join(str(directory), role)
    with open(path, "w") as f:
      f.write(f"#!/bin/sh\nexec {sys.executable} {os.path.abspath(__file__)} --role={role} \"$@\"\n")
    os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
  return os.path.join(str(directory), "ffmpeg")


def decode(argv):
  src = argv[argv.index("-i") + 1]
  want = int(argv[argv.index("-ac") + 1])
  assert argv[argv.index("-f") + 1] == "s16le" and argv[argv.index("-ar") + 1] == "44100", argv
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


def mux(argv):
  assert argv[-1] == "-y", argv
  output = argv[-2]
  record = {"argv": argv}
  if "pipe:" in argv:
    data = sys.stdin.buffer.read()
    record["stdin_bytes"] = len(data); record["stdin_sha1"] = hashlib.sha1(data).hexdigest()
  blob = json.dumps(record)
  with open(output, "w") as f:
    f.write(blob + "\n" + " " * 120000)           # > 1e5 bytes: a finished output for the skip-existing rule
  return 0


def probe(argv):
  target = argv[-1]
  sel = argv[argv.index("-select_streams") + 1]
  if sel == "V":
    iv = argv[argv.index("-read_intervals") + 1]
    end = float(iv[2:]) if iv.startswith("%+") else 1e4
    frames, t = [], 0.0
    while t <= end:
      frames.append({"pts_time": f"{t:.6f}"}); t += KEY_FRAME_STEP
    print(json.dumps({"frames": frames, "streams": [{"codec_type": "video"}], "format": {"filename": target}}))
  else:
    ad = "described" in os.path.basename(target)
    print(json.dumps({"streams": [{"codec_type": "audio", "disposition": {"descriptions": int(ad), "visual_impaired": int(ad), "original": int(not ad)}}],
                      "format": {"filename": target}}))
  return 0


def main(argv):
  role = argv[0].split("=", 1)[1]
  argv = argv[1:]
  if role == "ffprobe":
    return probe(argv)
  if argv[-1] == "-":
    return decode(argv)
  return mux(argv)


if __name__ == "__main__":
  sys.exit(main(sys.argv[1:]))
