"""Recording stand-in for the `ffmpeg` module (ffmpeg-python 0.2.0, requirements.txt:1 of the reference), used ONLY by
make_golden.py to capture the command lines the reference builds: the package is not in this image (no network) and
neither is an ffmpeg binary, so the reference's calls are driven against this double and what it compiles is recorded
as a fixture (tests/golden/commands.json).

It restates the published compile rules of ffmpeg-python 0.2.0 for the graph shapes the reference builds -- plain
inputs into one output, no filters (describealign.py:152-153, :470-487, :491-509) -- and `probe` (:445-447, :461):

  * `ffmpeg/_utils.py  convert_kwargs_to_cmd_line_args`: options in SORTED key order, `-key value`, a None value gives a
    bare `-key`, a list value repeats the key;
  * `ffmpeg/_run.py  _get_input_args`: `-f <format>` first, then the remaining options, then `-i <filename>`;
  * `ffmpeg/_run.py  _get_output_args`: `-map <input index>` per incoming stream unless the output has the single input 0,
    then `-f <format>`, `-b:v`, `-b:a`, then the remaining options, then the file name;
  * `ffmpeg/_run.py  get_args / compile`: [cmd] + inputs in edge order + outputs + global arguments
    (`overwrite_output()` is the global argument `-y`);
  * `ffmpeg/_probe.py  probe`: [cmd, -show_format, -show_streams, -of, json] + options + [filename].

Nothing here is shipped or imported by the product; the product builds its argv directly (combine.py, media.py) and
tests/test_host_cpu.py compares that with the recorded fixture."""
from __future__ import annotations


class Error(Exception):
  def __init__(self, cmd="ffmpeg", stdout=b"", stderr=b""):
    super().__init__(f"{cmd} error (see stderr output for detail)")
    self.stdout, self.stderr = stdout, stderr


def convert_kwargs_to_cmd_line_args(kwargs):
  args = []
  for k in sorted(kwargs.keys()):
    v = kwargs[k]
    if isinstance(v, (list, tuple)):
      for value in v:
        args.append(f"-{k}")
        if value is not None:
          args.append(f"{value}")
      continue
    args.append(f"-{k}")
    if v is not None:
      args.append(f"{v}")
  return args


RECORDED = []          # every argv that was compiled for running / probing, in call order: (kind, argv)
PROBE_RESULT = {}      # what probe() returns (set by the caller)


class _Input:
  def __init__(self, filename, kwargs):
    self.filename, self.kwargs = filename, dict(kwargs)

  def output(self, *streams_and_filename, **kwargs):
    return output(self, *streams_and_filename, **kwargs)

  def args(self):
    kw = dict(self.kwargs)
    fmt = kw.pop("format", None)
    args = []
    if fmt:
      args += ["-f", fmt]
    args += convert_kwargs_to_cmd_line_args(kw)
    return args + ["-i", self.filename]


class _Output:
  def __init__(self, inputs, filename, kwargs, global_args=()):
    self.inputs, self.filename, self.kwargs, self.global_args = list(inputs), filename, dict(kwargs), list(global_args)

  def overwrite_output(self):
    return _Output(self.inputs, self.filename, self.kwargs, self.global_args + ["-y"])

  def global_args_(self, *args):
    return _Output(self.inputs, self.filename, self.kwargs, self.global_args + list(args))

  def get_args(self):
    args = []
    for node in self.inputs:
      args += node.args()
    if len(self.inputs) > 1:
      for k in range(len(self.inputs)):
        args += ["-map", str(k)]
    kw = dict(self.kwargs)
    if "format" in kw:
      args += ["-f", kw.pop("format")]
    if "video_bitrate" in kw:
      args += ["-b:v", str(kw.pop("video_bitrate"))]
    if "audio_bitrate" in kw:
      args += ["-b:a", str(kw.pop("audio_bitrate"))]
    args += convert_kwargs_to_cmd_line_args(kw)
    args += [self.filename]
    return args + self.global_args

  def compile(self, cmd="ffmpeg", overwrite_output=False):
    return [cmd] + self.get_args() + (["-y"] if overwrite_output else [])

  def run(self, cmd="ffmpeg", capture_stdout=False, capture_stderr=False, input=None, quiet=False, overwrite_output=False):
    RECORDED.append(("run", self.compile(cmd, overwrite_output)))
    return b"", b""

  def run_async(self, cmd="ffmpeg", pipe_stdin=False, pipe_stdout=False, pipe_stderr=False, quiet=False, overwrite_output=False):
    RECORDED.append(("run_async", self.compile(cmd, overwrite_output)))

    class _Proc:
      def communicate(self, data=None):
        RECORDED.append(("stdin_bytes", len(data) if data is not None else 0))
        return b"", b""
    return _Proc()


def input(filename, **kwargs):      # noqa: A001 -- the package's own name
  return _Input(filename, kwargs)


def output(*streams_and_filename, **kwargs):
  *streams, filename = streams_and_filename
  return _Output(streams, filename, kwargs)


def compile(stream_spec, cmd="ffmpeg", overwrite_output=False):      # noqa: A001
  return stream_spec.compile(cmd, overwrite_output)


def probe(filename, cmd="ffprobe", timeout=None, **kwargs):
  RECORDED.append(("probe", [cmd, "-show_format", "-show_streams", "-of", "json"] + convert_kwargs_to_cmd_line_args(kwargs) + [filename]))
  return PROBE_RESULT
