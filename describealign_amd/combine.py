"""combine() and the command line: the entry points of describealign kept by this build
(reference describealign.py:1031-1175, :1773-1849), with the compute section running on
MI355X.  For every (video, audio description) pair: decode PCM, extract features, align, warn
on suspicious similarity, write the alignment plot + text report, and -- when an ffmpeg binary
is available -- mux the described audio onto the re-timed video with the same `setts`
bitstream-filter command line the reference issues.

A directory batch shards across GPUs with no collectives: pair k goes to GPU k % gpus, one
worker process per GPU (`--gpus`).

--stretch_audio (replace_aligned_segments, :230-416, and the loudness / peak handling around it,
:1135-1153) runs on the GPU as well: the PCM uploaded for the feature kernels stays resident and
only the finished int16 track is copied back.  The key-frame probe of the default mux (:443-458) and the
"is the first audio track already an audio description" probe of the --stretch_audio mux (:460-462) go
through the ffprobe binary next to ffmpeg.  Out of scope: the GUI.
"""
from __future__ import annotations

import argparse
import glob
import os
import re
import subprocess
import sys

import numpy as np

from . import REFERENCE_VERSION, __version__
from . import media, report

VIDEO_EXTENSIONS = set(['mp4', 'mkv', 'avi', 'mov', 'webm', 'm4v', 'flv', 'vob'])
AUDIO_EXTENSIONS = set(['mp3', 'm4a', 'opus', 'wav', 'aac', 'flac', 'ac3', 'mka'])
default_output_dir = os.path.expanduser('~') + '/videos_with_ad'
default_alignment_dir = os.path.expanduser('~') + '/alignment_plots'


def _natural_key(path):
  return [int(tok) if tok.isdigit() else tok.lower() for tok in re.split(r'(\d+)', os.path.basename(path))]


def ensure_folders_exist(dirs):
  for d in dirs:
    if not os.path.isdir(d):
      print(f"Directory not found, creating it: {d}")
      os.makedirs(d)


def _candidate_files(path):
  """The files an input path stands for: the entries of a list, the children of a directory,
  or the file itself.  Missing inputs raise the reference's messages (:97-110)."""
  if isinstance(path, list):
    listed = [os.path.abspath(p) for p in path]
    missing = next((p for p in listed if not os.path.isfile(p)), None)
    if missing is not None:
      raise RuntimeError(f"No file found at input path:\n  {missing}")
    return listed, path
  where = os.path.abspath(path)
  if os.path.isfile(where):
    return [where], where
  if not os.path.isdir(where):
    raise RuntimeError(f"No file or directory found at input path:\n  {where}")
  children = glob.glob(os.path.join(glob.escape(where), "*"))
  if not children:
    raise RuntimeError(f"Empty input directory:\n  {where}")
  return children, where


def get_sorted_filenames(path, extensions, alt_extensions=set([])):
  """describealign.get_sorted_filenames (:94-121): the media files behind `path` in natural
  order, plus per file 0 when its extension is one of `extensions` and 1 when it only matched
  `alt_extensions`."""
  candidates, shown = _candidate_files(path)
  kind = {}
  for name in candidates:
    ext = os.path.splitext(name)[1][1:]
    if ext in extensions:
      kind[name] = 0
    elif ext in alt_extensions:
      kind[name] = 1
  if not kind:
    raise RuntimeError(f"No files with valid extensions found at input path:\n  {shown}\n"
                       "Did you accidentally put the audio filepath before the video filepath?\n"
                       "The video path should be the first positional input, audio second.\n"
                       "Or maybe you need to add a new extension to this script's regex?\n"
                       f"valid extensions for this input are:\n  {extensions}")
  ordered = sorted(kind, key=_natural_key)
  return ordered, [kind[name] for name in ordered]


from .distrib import shard_pairs  # noqa: E402  (round-robin pair -> GPU assignment)


def parse_key_frame_times(ffprobe_json, entry="pts_time"):
  """Key-frame timestamps out of `ffprobe -show_frames -skip_frame nokey -of json` output
  (what ffmpeg.probe hands get_key_frame_data, :443-449): frames lacking the entry are skipped."""
  import json
  frames = json.loads(ffprobe_json).get("frames", [])
  return np.array([float(f[entry]) for f in frames if entry in f], dtype=np.float64)


def get_key_frame_data(video_file, time=None, entry="pts_time", ffprobe=None):
  """describealign.get_key_frame_data (:443-449) through the ffprobe binary directly."""
  ffprobe = ffprobe or media.find_ffprobe()
  if ffprobe is None:
    raise RuntimeError("no ffprobe binary on PATH")
  interval = f"%+{max(60, time + 40)}" if time is not None else "%"
  argv = ([ffprobe, "-show_format", "-show_streams", "-of", "json"] +
          _options(select_streams='V', show_frames=None, skip_frame='nokey', read_intervals=interval, show_entries='frame=' + entry) + [video_file])
  res = subprocess.run(argv, capture_output=True)
  if res.returncode != 0:
    raise RuntimeError("ffprobe error: " + res.stderr.decode("utf-8", "replace"))
  return parse_key_frame_times(res.stdout.decode("utf-8", "replace"), entry)


def closest_key_frame_time(key_frame_times, time):
  """Midpoint between the key frames on either side of `time` (:451-458): ffmpeg's -ss then cuts
  at the last key frame before the described audio starts."""
  key_frame_times = np.asarray(key_frame_times, dtype=np.float64)
  if len(key_frame_times) == 0:
    key_frame_times = np.array([0.0])
  later = key_frame_times[key_frame_times > time]
  earlier = key_frame_times[key_frame_times <= time]
  nxt = np.min(later) if len(later) > 0 else time
  prv = np.max(earlier) if len(earlier) > 0 else nxt
  return (prv + nxt) / 2.


def get_closest_key_frame_time(video_file, time, ffprobe=None):
  return closest_key_frame_time(get_key_frame_data(video_file, time, ffprobe=ffprobe), time)


def _options(**kwargs):
  """Options as the reference's ffmpeg-python graph compiles them (ffmpeg-python 0.2.0, `convert_kwargs_to_cmd_line_args`): in
  sorted key order, `-key value`, a None value a bare `-key`.  Keeping that order keeps the "FFmpeg command:" line of the
  alignment report identical to the reference's (tests/golden/commands.json holds what the reference compiles)."""
  args = []
  for k in sorted(kwargs):
    args.append(f"-{k}")
    if kwargs[k] is not None:
      args.append(f"{kwargs[k]}")
  return args


def _mux_command(ffmpeg, video_file, audio_desc_file, output_filename, setts_cmd, video_offset,
                 after_start_key_frame, median_slope):
  """ffmpeg argv for the default (video re-timing) output: what the graph of :489-510 compiles to."""
  start_offset = video_offset - after_start_key_frame
  audio_codec = 'copy' if os.path.splitext(audio_desc_file)[1] != '.wav' else 'aac'
  standards = 'normal' if os.path.splitext(audio_desc_file)[1] != '.flac' else 'experimental'
  sub_stretch = f":duration='DURATION*{1. / median_slope:.6f}'"
  return ([ffmpeg] + _options(itsoffset=f"{max(0, start_offset):.6f}") + ["-i", audio_desc_file] +
          _options(an=None, ss=f"{after_start_key_frame:.6f}", itsoffset=f"{max(0, -start_offset):.6f}", dn=None) + ["-i", video_file] +
          ["-map", "0", "-map", "1"] +
          _options(acodec=audio_codec, vcodec='copy', scodec='copy', max_interleave_delta='0', loglevel='error',
                   strict=standards, movflags='frag_keyframe',
                   **{'bsf:v': f"setts=pts='{setts_cmd}':dts='{setts_cmd}'", 'bsf:s': f"setts=ts='{setts_cmd}'" + sub_stretch,
                      "disposition:a:0": "default+visual_impaired+descriptions", "metadata:s:a:0": "title=AD"}) +
          [output_filename, "-y"])


def parse_first_audio_track_is_ad(ffprobe_json) -> bool:
  """What is_first_video_track_ad (:460-462) reads out of `ffprobe -show_streams -select_streams a -of json`:
  the first audio stream's `descriptions` or `visual_impaired` disposition."""
  import json
  streams = json.loads(ffprobe_json).get("streams", [])
  if not streams:
    return False
  disp = streams[0].get("disposition", {})
  return bool(disp.get("descriptions") or disp.get("visual_impaired"))


def is_first_video_track_ad(video_file, ffprobe=None) -> bool:
  """describealign.is_first_video_track_ad (:460-462) through the ffprobe binary directly."""
  ffprobe = ffprobe or media.find_ffprobe()
  if ffprobe is None:
    raise RuntimeError("no ffprobe binary on PATH")
  res = subprocess.run([ffprobe, "-show_format", "-show_streams", "-of", "json", "-select_streams", "a", video_file], capture_output=True)
  if res.returncode != 0:
    raise RuntimeError("ffprobe error: " + res.stderr.decode("utf-8", "replace"))
  return parse_first_audio_track_is_ad(res.stdout.decode("utf-8", "replace"))


def _replaced_media_command(ffmpeg, output_filename, video_file, first_track_is_ad=False):
  """ffmpeg argv for the --stretch_audio output (what the graphs of :468-487 compile to): the new stereo track is
  piped in as s16le and either stored on its own (audio-only input) or muxed in front of the
  original streams.  The video's own first audio track becomes "original" unless it already is an
  audio description (the output of a previous run, :478-480)."""
  head = [ffmpeg, "-f", "s16le"] + _options(acodec='pcm_s16le', ac=2, ar=media.AUDIO_SAMPLE_RATE) + ["-i", "pipe:"]
  if video_file is None:
    return head + _options(loglevel='error') + [output_filename, "-y"]
  kwargs = {"c:a:0": "aac", "disposition:a:0": "default+visual_impaired+descriptions",
            "metadata:s:a:0": "title=AD", "disposition:a:1": "visual_impaired+descriptions"}
  if not first_track_is_ad:
    kwargs.update({"disposition:a:1": "original", "metadata:s:a:1": "title=original"})
  return (head + _options(dn=None) + ["-i", video_file, "-map", "0", "-map", "1"] +
          _options(acodec='copy', vcodec='copy', scodec='copy', max_interleave_delta='0', loglevel='error', **kwargs) +
          [output_filename, "-y"])


def _write_replaced_media(ffmpeg, output_filename, frames, video_file):
  """write_replaced_media_to_disk with a media array (:468-487).  Without an ffmpeg binary the
  track is written as a .wav next to where the output would have gone."""
  if frames.shape[1] == 1:
    frames = np.repeat(frames, 2, axis=1)
  if ffmpeg is None:
    wav = os.path.splitext(output_filename)[0] + ".wav"
    media.write_wav(wav, np.ascontiguousarray(frames.T))
    return f"(no ffmpeg binary on PATH; replaced audio track written to {wav})"
  first_is_ad = False
  if video_file is not None:
    try:
      first_is_ad = is_first_video_track_ad(video_file)
    except (RuntimeError, OSError, ValueError, KeyError) as e:
      print(f"  WARNING: could not probe the video's first audio track ({e}); labelling it \"original\"")
  argv = _replaced_media_command(ffmpeg, output_filename, video_file, first_is_ad)
  res = subprocess.run(argv, input=np.ascontiguousarray(frames).tobytes(), capture_output=True)
  if res.returncode != 0 or len(res.stderr) > 0:
    print("  ERROR: ffmpeg failed to write output file: " + output_filename)
    print("FFmpeg error:")
    print(res.stderr.decode("utf-8", "replace"))
    raise RuntimeError("FFmpeg error.")
  return subprocess.list2cmdline(argv).replace('\\', '/')


def _output_name(video_file, prepend, output_dir):
  return os.path.join(output_dir, prepend + os.path.split(video_file)[1])


def _already_done(output_filename):
  return os.path.exists(output_filename) and os.path.getsize(output_filename) > 1e5        # (:1087-1089)


def _finish_pair(outputs, video_file, audio_desc_file, has_audio_extension, ctx, output_filename, stretch_audio,
                 no_pitch_correction, alignment_dir):
  """Everything after align() for one pair (:1123-1174): warnings, the setts expression, the muxed
  or stretched output, the plot and the text report."""
  audio_desc_times, video_times, similarity_percent, path, median_slope = outputs
  if similarity_percent < 20:
    print(f"  WARNING: similarity {similarity_percent:.1f}%, likely mismatched files")
  if similarity_percent > 90:
    print(f"  WARNING: similarity {similarity_percent:.1f}%, likely undescribed media")
  if (median_slope < .1) or (median_slope > 10):
    print("  WARNING: median slope estimation failed, output subtitles may be misaligned")
    median_slope = 1.
  video_offset = video_times[0] - audio_desc_times[0]
  setts_cmd = report.encode_fit_as_ffmpeg_expr(audio_desc_times, video_times, video_offset)
  ffmpeg_command = ""
  ffmpeg = media.find_ffmpeg()
  if stretch_audio:
    # :1135-1159 -- loudness matching, replace_aligned_segments, peak normalisation and the int16
    # interleave all run on the PCM already resident on the GPU; only the finished track comes back
    print("  stretching audio...                         \r", end='')
    frames, _ = ctx.stretch_resident(audio_desc_times, video_times, no_pitch_correction)
    print("  processing output file...                   \r", end='')
    ffmpeg_command = _write_replaced_media(ffmpeg, output_filename, frames,
                                           None if has_audio_extension else video_file)
  elif ffmpeg is not None and not has_audio_extension:
    print("  processing output file...                   \r", end='')
    # to make ffmpeg cut at the last key frame before the audio starts, use a timestamp after it (:1162-1164)
    try:
      after_start_key_frame = get_closest_key_frame_time(video_file, video_offset)
    except (RuntimeError, OSError, ValueError) as e:
      # no ffprobe next to ffmpeg, or it could not read the file: cut at the offset itself instead of
      # losing the pair after all the alignment work is done
      print(f"  WARNING: key frame lookup failed ({e}); cutting at the audio start instead")
      after_start_key_frame = max(0., video_offset)
    argv = _mux_command(ffmpeg, video_file, audio_desc_file, output_filename, setts_cmd, video_offset,
                        after_start_key_frame, median_slope)
    res = subprocess.run(argv, capture_output=True)
    if res.returncode != 0:
      print("  ERROR: ffmpeg failed to write output file: " + output_filename)
      print(res.stderr.decode("utf-8", "replace"))
      raise RuntimeError("FFmpeg error.")
    ffmpeg_command = subprocess.list2cmdline(argv).replace('\\', '/')
  else:
    ffmpeg_command = f"(no ffmpeg binary on PATH; output not muxed) setts expression: {setts_cmd}"
  stem = os.path.join(alignment_dir, os.path.splitext(os.path.split(video_file)[1])[0])
  report.plot_alignment(stem, path, audio_desc_times, video_times, similarity_percent, median_slope,
                        stretch_audio, no_pitch_correction, ffmpeg_command)
  return dict(audio_desc_times=audio_desc_times, video_times=video_times, similarity_percent=similarity_percent,
              median_slope=median_slope, setts=setts_cmd, report=stem + ".txt")


def process_pair(video_file, audio_desc_file, has_audio_extension, ctx, stretch_audio=False, prepend="ad_",
                 no_pitch_correction=False, output_dir=default_output_dir, alignment_dir=default_alignment_dir):
  """One iteration of the reference's per-pair loop (:1077-1174)."""
  from . import _native
  from .align import align
  output_filename = _output_name(video_file, prepend, output_dir)
  print(f" {output_filename}")
  if (not stretch_audio) & has_audio_extension:
    raise RuntimeError("Argument --stretch_audio is required when both inputs are audio files.")
  if _already_done(output_filename):
    print("   output file already exists, skipping...")
    return None
  num_channels = 2 if stretch_audio else 1
  # decoder pipe -> ring of page-locked pieces -> HBM: every piece's copy is enqueued while the next is being decoded,
  # and the host never holds a whole file (parse_audio_from_file + the float16 array of :149-157)
  stream = _native.PcmStream(ctx.device, num_channels)
  ring = [_native.pinned_empty((media.PIECE_BYTES // 2,), np.int16) for _ in range(3)]
  try:
    print("  reading video file...\r", end='')
    media.stream_file_to_device(stream, video_file, num_channels, ring)
    ctx.pcm_adopt(_native.SIDE_VIDEO, stream)                       # the PCM stays resident on the GPU
    print("  computing video features... \r", end='')
    video_features = ctx.features_resident(_native.SIDE_VIDEO)
    print("  reading audio file...       \r", end='')
    media.stream_file_to_device(stream, audio_desc_file, num_channels, ring)
    ctx.pcm_adopt(_native.SIDE_AUDIO, stream)
    print("  computing audio features...\r", end='')
    audio_desc_features = ctx.features_resident(_native.SIDE_AUDIO)
  finally:
    stream.close()
    del ring
  outputs = align(video_features, audio_desc_features, video_features[0], audio_desc_features[0], ctx=ctx)
  return _finish_pair(outputs, video_file, audio_desc_file, has_audio_extension, ctx, output_filename, stretch_audio,
                      no_pitch_correction, alignment_dir)


class _PinnedPool:
  """Page-locked PCM buffers for the decoder threads of a directory batch (da_host_alloc): a decoded
  file lands where the GPU can fetch it by DMA, so its upload (da_pcm_upload_async) overlaps the
  kernels of the pair before.  Buffers are recycled: locking pages costs more than filling them."""

  def __init__(self):
    import threading
    self._free, self._lock = [], threading.Lock()

  def alloc(self, shape):
    from . import _native
    need = int(np.prod(shape))
    with self._lock:
      fit = [b for b in self._free if b.size >= need]
      flat = min(fit, key=lambda b: b.size) if fit else None
      if flat is not None:
        self._free = [b for b in self._free if b is not flat]
    if flat is None:
      flat = _native.pinned_empty((need + need // 8 + 64,), np.int16)
    view = flat[:need].reshape(shape)
    return view

  def release(self, view):
    flat = view
    while isinstance(flat, np.ndarray) and isinstance(flat.base, np.ndarray):
      flat = flat.base                                  # back to the whole page-locked array
    with self._lock:
      if all(b is not flat for b in self._free):
        self._free.append(flat)


def process_batch(todo, ctx, prepend="ad_", no_pitch_correction=False, output_dir=default_output_dir,
                  alignment_dir=default_alignment_dir, lp_workers=None, decode_ahead=3, stretch_audio=False):
  """A directory batch on one GPU: same results and files as calling process_pair for every
  (video, audio description) in `todo`, but pipelined -- decoding of the next pairs (a small thread
  pool, `decode_ahead` pairs in flight), the GPU stages of pair k+1 and the host-side LP / DP stages
  of pair k overlap (align.AlignPipeline).  With stretch_audio the decoded PCM of the pairs in flight
  is kept on the host and uploaded again (to a second context on the same GPU) when a pair's nodes
  arrive; a semaphore admits at most `workers + 2` (<= 6) such pairs between decode and the finished
  track (2 h stereo = 2.5 GB of PCM per pair), and the pipeline window is narrowed to match.
  Returns the per-pair results."""
  import concurrent.futures as cf
  import contextlib
  import io
  import threading
  from . import _native
  from .align import AlignPipeline, default_worker_count
  work = []
  for video_file, audio_desc_file, has_audio_extension in todo:
    if has_audio_extension and not stretch_audio:
      raise RuntimeError("Argument --stretch_audio is required when both inputs are audio files.")
    out = _output_name(video_file, prepend, output_dir)
    if _already_done(out):
      print(f" {out}\n   output file already exists, skipping...")
      continue
    work.append((video_file, audio_desc_file, out, has_audio_extension))
  num_channels = 2 if stretch_audio else 1
  results = []
  if not work:
    return results
  decoders = cf.ThreadPoolExecutor(max_workers=2)
  decoded = {}
  kept = {}                  # stretch_audio: PCM of the pairs in flight (bounded by `held`)
  stretch_ctx = _native.Context(ctx.device, ctx.precision) if stretch_audio else None
  local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
  workers = lp_workers or min(default_worker_count(local_world), max(2, len(work)))
  if stretch_audio:
    workers = min(workers, 4)
  max_held = workers + 2
  held = threading.BoundedSemaphore(max_held)

  pool = _PinnedPool()
  # without --stretch_audio nothing needs the PCM on the host: each decoder thread streams its file through a ring of
  # page-locked pieces into a device buffer of its own (PcmStream) and the GPU thread adopts that buffer -- no
  # whole-file host array, no second pass.  (With --stretch_audio the second context uploads the PCM again from the host.)
  streams, streams_lock, tls = [], threading.Lock(), threading.local()

  def decode_to_device(path):
    with streams_lock:
      st = streams.pop() if streams else None
    if st is None:
      st = _native.PcmStream(ctx.device, num_channels)
    if not hasattr(tls, "ring"):
      tls.ring = [_native.pinned_empty((media.PIECE_BYTES // 2,), np.int16) for _ in range(3)]
    try:
      media.stream_file_to_device(st, path, num_channels, tls.ring)
    except BaseException:
      st.close()
      raise
    return st

  def request(k):
    if k < len(work) and k not in decoded:
      if stretch_audio:
        decoded[k] = (decoders.submit(media.parse_audio_from_file, work[k][0], num_channels, pool.alloc),
                      decoders.submit(media.parse_audio_from_file, work[k][1], num_channels, pool.alloc))
      else:
        decoded[k] = (decoders.submit(decode_to_device, work[k][0]), decoders.submit(decode_to_device, work[k][1]))

  def make_job(k):
    def job(c):
      for ahead in range(k, k + 1 + decode_ahead):
        request(ahead)
      if stretch_audio:
        held.acquire()         # released when pair k's track has been written (below)
      fv, fa = decoded.pop(k)
      from .align import RESIDENT_PCM
      if not stretch_audio:
        sv, sa = fv.result(), fa.result()
        c.pcm_adopt(_native.SIDE_VIDEO, sv)
        c.pcm_adopt(_native.SIDE_AUDIO, sa)
        with streams_lock:
          streams.extend((sv, sa))                 # they now hold the buffers of the pair before: the next files' targets
        return RESIDENT_PCM                        # features + matching + chain enqueue: ONE native call (da_pair_stage)
      video_arr, audio_desc_arr = fv.result(), fa.result()
      kept[k] = (video_arr, audio_desc_arr)        # uploaded again to the second context when the pair's nodes arrive
      # both uploads are enqueued before the first feature kernel: the second copy runs under it
      c.pcm_upload_async(_native.SIDE_VIDEO, video_arr)
      c.pcm_upload_async(_native.SIDE_AUDIO, audio_desc_arr)
      return RESIDENT_PCM
    return job

  quiet = contextlib.redirect_stdout(io.StringIO())          # align()'s progress lines would interleave
  with AlignPipeline(ctx, lp_workers=workers) as pipe:
    if len(work) >= 8:
      pipe.warm()
    it = pipe.run((make_job(k) for k in range(len(work))), window=(max_held if stretch_audio else None), expected=len(work))
    for k in range(len(work)):
      with quiet:
        outputs = next(it)
      video_file, audio_desc_file, out, has_audio_extension = work[k]
      print(f" {out}")
      if stretch_audio:
        video_arr, audio_desc_arr = kept.pop(k)
        stretch_ctx.pcm_upload(_native.SIDE_VIDEO, video_arr)
        stretch_ctx.pcm_upload(_native.SIDE_AUDIO, audio_desc_arr)
        pool.release(video_arr); pool.release(audio_desc_arr)
        del video_arr, audio_desc_arr
      try:
        results.append(_finish_pair(outputs, video_file, audio_desc_file, has_audio_extension,
                                    stretch_ctx if stretch_audio else ctx, out, stretch_audio, no_pitch_correction,
                                    alignment_dir))
      finally:
        if stretch_audio:
          held.release()
  decoders.shutdown(wait=True)
  for pending in decoded.values():                 # decoded ahead but never consumed (an exception above)
    for f in pending:
      if not stretch_audio and f.exception() is None:
        f.result().close()
  for st in streams:
    st.close()
  if stretch_ctx is not None:
    stretch_ctx.close()
  return results


def _worker(gpu, indices, pairs, kwargs, precision, n_workers=1):
  from . import _native
  if n_workers > 1:
    # one process per GPU on one host: tell the batch pipeline, so that the LP worker pools of the
    # processes split the host's cores between them instead of all pinning to the same ones
    os.environ["LOCAL_WORLD_SIZE"] = str(n_workers)
    os.environ["LOCAL_RANK"] = str(gpu)
  # DALIGN_DEVICE_OVERRIDE=<id>: every worker uses that device (exercising the sharded path on a box
  # with fewer GPUs than workers)
  dev = int(os.environ["DALIGN_DEVICE_OVERRIDE"]) if os.environ.get("DALIGN_DEVICE_OVERRIDE") else gpu
  ctx = _native.Context(dev, precision)
  todo = [pairs[k] for k in indices]
  if len(todo) >= 3:
    process_batch(todo, ctx, prepend=kwargs["prepend"], no_pitch_correction=kwargs["no_pitch_correction"],
                  output_dir=kwargs["output_dir"], alignment_dir=kwargs["alignment_dir"],
                  stretch_audio=bool(kwargs.get("stretch_audio")))
  else:
    for v, a, alt in todo:
      process_pair(v, a, alt, ctx, **kwargs)
  ctx.close()


_ABORT_HINT = "If not, press ctrl+c to kill this script."


def _checkpoint(question, go_on, lead_blank=False):
  """One of the reference's interactive stops (:1035-1040, :1058-1062): the question, how to abort, Enter to go on."""
  lines = ([""] if lead_blank else []) + [question, _ABORT_HINT]
  for line in lines:
    print(line)
  input(go_on)
  print("")


def _paired_inputs(video, audio, interactive):
  """Both input paths resolved and paired (:1033-1046): [(video file, audio description file, video side is an audio file)].
  Audio files on the video side are legal (--stretch_audio) but worth a question; unequal counts are an error."""
  video_files, is_audio = get_sorted_filenames(video, VIDEO_EXTENSIONS, AUDIO_EXTENSIONS)
  if interactive and any(is_audio):
    _checkpoint("One or more audio files found in video input. Was this intentional?",
                "If this was intended, press Enter to continue...", lead_blank=True)
  audio_desc_files, _ = get_sorted_filenames(audio, AUDIO_EXTENSIONS)
  counts = (len(video_files), len(audio_desc_files))
  if counts[0] != counts[1]:
    raise RuntimeError("Number of valid files in input paths are not the same.\n"
                       f"The video path has {counts[0]} files\nThe audio path has {counts[1]} files")
  return list(zip(video_files, audio_desc_files, is_audio))


def _list_pairs(pairs, interactive):
  """The pairing shown to the user, file names only, and the second stop (:1053-1062)."""
  print("")
  for video_file, audio_desc_file, _ in pairs:
    print("\n".join(os.path.split(f)[1] for f in (video_file, audio_desc_file)) + "\n")
  if interactive:
    _checkpoint("Are the above input file pairings correct?", "If they are correct, press Enter to continue...")


def combine(video, audio, stretch_audio=False, yes=False, prepend="ad_", no_pitch_correction=False,
            output_dir=default_output_dir, alignment_dir=default_alignment_dir, gpus=1, precision="f32", device=0):
  """Same signature as the reference's combine() (:1031-1032) plus gpus/precision/device."""
  from . import _native
  pairs = _paired_inputs(video, audio, interactive=not yes)
  print("")
  ensure_folders_exist([output_dir, alignment_dir])
  _list_pairs(pairs, interactive=not yes)
  print(f"Processing files with describealign_amd v{__version__} (reproducing describealign v{REFERENCE_VERSION}):")
  prec = _native.PREC_F32 if precision == "f32" else _native.PREC_BF16
  kwargs = dict(stretch_audio=stretch_audio, prepend=prepend, no_pitch_correction=no_pitch_correction,
                output_dir=output_dir, alignment_dir=alignment_dir)
  if gpus <= 1 or len(pairs) <= 1:
    _worker(device, range(len(pairs)), pairs, kwargs, prec)
  else:
    import multiprocessing as mp
    mpctx = mp.get_context("spawn")
    procs = [mpctx.Process(target=_worker, args=(g, idx, pairs, kwargs, prec, gpus))
             for g, idx in enumerate(shard_pairs(len(pairs), gpus)) if idx]
    for p in procs:
      p.start()
    for p in procs:
      p.join()
    if any(p.exitcode != 0 for p in procs):
      raise RuntimeError("one or more GPU workers failed")
  print("All files processed.       ")


def command_line_interface(argv=None):
  """Same flags as the reference CLI (:1791-1817) plus --gpus / --precision / --device."""
  parser = argparse.ArgumentParser(description="Replaces a video's sound with an audio description.",
                                   usage="describealign_amd video_file.mp4 audio_file.mp3")
  parser.add_argument("video", help='A video file or directory containing video files.', nargs='?', default=None)
  parser.add_argument("audio", help='An audio file or directory containing audio files.', nargs='?', default=None)
  parser.add_argument('--stretch_audio', action='store_true',
                      help='Stretches the input audio to fit the input video (runs on the GPU).')
  parser.add_argument('--yes', action='store_true', help='Auto-skips user prompts asking to verify information.')
  parser.add_argument("--prepend", default="ad_", help='Output file name prepend text. Default is "ad_"')
  parser.add_argument('--no_pitch_correction', action='store_true',
                      help='Skips pitch correction step when stretching audio.')
  parser.add_argument("--output_dir", default=default_output_dir,
                      help='Directory combined output media is saved to. Default is "videos_with_ad"')
  parser.add_argument("--alignment_dir", default=default_alignment_dir,
                      help='Directory alignment data and plots are saved to. Default is "alignment_plots"')
  parser.add_argument('--version', action='store_true', help='Prints the installed version.')
  parser.add_argument('--gpus', type=int, default=1, help='Shard a directory batch over this many GPUs.')
  parser.add_argument('--device', type=int, default=0, help='GPU index for single-GPU runs.')
  parser.add_argument('--precision', choices=['f32', 'bf16'], default='f32', help='Similarity GEMM input precision.')
  args = parser.parse_args(argv)
  if args.version:
    print(f"version: {__version__} (describealign {REFERENCE_VERSION} results)")
  elif args.video and args.audio:
    combine(args.video, args.audio, args.stretch_audio, args.yes, args.prepend, args.no_pitch_correction,
            args.output_dir, args.alignment_dir, gpus=args.gpus, precision=args.precision, device=args.device)
  else:
    parser.print_usage()


if __name__ == "__main__":
  command_line_interface()
