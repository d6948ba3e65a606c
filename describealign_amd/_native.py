"""ctypes binding of libdalign.so (C ABI: include/dalign.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is usable,
creating a Context raises.  Build the library with `python -c "import __graft_entry__ as g;
g.build()"` or `make -C describealign_amd/csrc`.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DALIGN_LIB") or os.path.join(_HERE, "libdalign.so")   # DALIGN_LIB: diagnostic builds
ABI_VERSION = 5
BF16_GUARD = 2.0 ** -7 + 2.0 ** -14   # csrc/dalign_common.h kBf16Guard: subtracted from the norm slot of the bf16 GEMM

PREC_F32, PREC_BF16 = 0, 1
SIDE_VIDEO, SIDE_AUDIO = 0, 1
MATCH_HASHED, MATCH_DENSE = 0, 1
MATCH_RESIDENT_ROWS = 0x100

ERR_CAPACITY = -3
ERR_MISMATCH = -4

EXPORTS = ["da_create", "da_destroy", "da_last_error", "da_abi_version", "da_pcm_upload", "da_pcm_upload_async", "da_host_alloc", "da_host_free",
           "da_pcm_stream_open", "da_pcm_stream_piece", "da_pcm_stream_sync", "da_pcm_stream_frames", "da_pcm_stream_error", "da_pcm_adopt", "da_pcm_exchange",
           "da_pcm_stream_close",
           "da_features_resident", "da_features", "da_match", "da_match_begin", "da_match_finish", "da_match_fetch",
           "da_match_corr", "da_trim", "da_match_dump_tile", "da_match_export_device", "da_match_import_device", "da_match_import_reserve", "da_match_import_commit", "da_chain", "da_chain_begin", "da_chain_begin_exclusive", "da_pair_stage", "da_chain_finish", "da_chain_resident", "da_chain_poll", "da_chain_masked",
           "da_refine", "da_stats", "da_replace_segments", "da_stretch_resident", "da_stretch_schedule"]


class Stats(C.Structure):
  _fields_ = [(n, C.c_double) for n in (
      "features_ms", "features_bytes", "prep_ms", "gemm_ms", "gemm_pairs", "gemm_flops", "verify_ms",
      "survivors", "matches", "chain_ms", "refine_kernel_ms", "refine_dp_ms", "refine_points", "h2d_ms",
      "resample_ms", "resample_points", "resample_bytes", "correlate_ms", "correlate_windows", "viterbi_ms",
      "splice_ms", "splice_points", "stretch_prepare_ms", "stretch_finish_ms", "chain_columns", "chain_column_width", "verify_kernel_ms")]

  def as_dict(self):
    return {n: getattr(self, n) for n, _ in self._fields_}


def build(verbose: bool = False) -> str:
  """Compile libdalign.so for gfx950 with hipcc (cross-compiles without a GPU)."""
  cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "all", "dbg"]      # dbg: the diagnostic / test-hook library the GPU tests load by name
  res = subprocess.run(cmd, capture_output=True, text=True)
  if verbose or res.returncode != 0:
    print(res.stdout[-4000:])
    print(res.stderr[-4000:])
  if res.returncode != 0 or not os.path.exists(LIB_PATH):
    raise RuntimeError("building libdalign.so failed")
  return LIB_PATH


_lib = None
_lib_lock = threading.Lock()


def load():
  global _lib
  with _lib_lock:
    if _lib is not None:
      return _lib
    if not os.path.exists(LIB_PATH):
      raise ImportError(f"{LIB_PATH} is missing: the HIP extension has not been built "
                        "(run __graft_entry__.build()); describealign_amd has no CPU fallback")
    # Every da_ctx owns a compute stream, a copy stream and one stream per chain DP in flight (a
    # persistent one-workgroup kernel that runs for ~1 s beside the GEMMs of later pairs).  The HIP
    # runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); streams that
    # share a queue execute in order, which would park a GEMM behind a chain DP.  Read at HIP
    # initialisation, so it has to be in the environment before the library is loaded.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
    lib = C.CDLL(LIB_PATH)
    vp, i64, i32 = C.c_void_p, C.c_int64, C.c_int
    P = C.POINTER
    lib.da_abi_version.restype = i32
    lib.da_create.argtypes = [i32, i32, P(vp)]
    lib.da_destroy.argtypes = [vp]; lib.da_destroy.restype = None
    lib.da_last_error.argtypes = [vp]; lib.da_last_error.restype = C.c_char_p
    lib.da_pcm_upload.argtypes = [vp, i32, vp, i64, i32, i32]
    lib.da_pcm_upload_async.argtypes = [vp, i32, vp, i64, i32, i32]
    lib.da_host_alloc.argtypes = [C.c_size_t, P(vp)]
    lib.da_host_free.argtypes = [vp]
    lib.da_pcm_stream_open.argtypes = [i32, i32, i64, P(vp)]
    lib.da_pcm_stream_piece.argtypes = [vp, vp, i64]
    lib.da_pcm_stream_sync.argtypes = [vp]
    lib.da_pcm_stream_frames.argtypes = [vp]; lib.da_pcm_stream_frames.restype = i64
    lib.da_pcm_stream_error.argtypes = [vp]; lib.da_pcm_stream_error.restype = C.c_char_p
    lib.da_pcm_adopt.argtypes = [vp, i32, vp]
    lib.da_pcm_exchange.argtypes = [vp, i32, vp]
    lib.da_pcm_stream_close.argtypes = [vp]; lib.da_pcm_stream_close.restype = None
    lib.da_features_resident.argtypes = [vp, i32, vp, i64, P(i64)]
    lib.da_features.argtypes = [vp, vp, i64, i32, i32, vp, i64, P(i64)]
    lib.da_match.argtypes = [vp, vp, i64, P(i64), vp, i64, P(i64), i32, i64, i64, vp, vp, vp, P(i64)]
    lib.da_match_begin.argtypes = [vp, vp, i64, P(i64), vp, i64, P(i64), i32, i64, i64]
    lib.da_match_finish.argtypes = [vp, P(i64)]
    lib.da_match_fetch.argtypes = [vp, vp, vp, vp, i64]
    lib.da_match_corr.argtypes = [vp, vp, vp, i64, vp]
    lib.da_trim.argtypes = [vp]
    lib.da_match_export_device.argtypes = [vp, vp, vp, i64]
    lib.da_match_import_device.argtypes = [vp, vp, vp, i64]
    lib.da_match_import_reserve.argtypes = [vp, i64, P(vp), P(vp)]
    lib.da_match_import_commit.argtypes = [vp, i64]
    lib.da_match_dump_tile.argtypes = [vp, i64, i64, vp, vp, vp]
    lib.da_chain.argtypes = [vp, vp, vp, vp, i64, C.c_double, vp, vp, P(i64)]
    lib.da_chain_begin.argtypes = [vp, P(C.c_uint64)]
    lib.da_chain_begin_exclusive.argtypes = [vp, P(C.c_uint64)]
    lib.da_pair_stage.argtypes = [vp, vp, i64, vp, i64, i32, P(i64), P(i64), P(i64), P(C.c_uint64)]
    lib.da_chain_finish.argtypes = [vp, C.c_uint64, C.c_double, vp, vp, P(i64)]
    lib.da_chain_resident.argtypes = [vp, C.c_double, vp, vp, P(i64)]
    lib.da_chain_poll.argtypes = [vp, C.c_uint64]
    lib.da_chain_masked.argtypes = [vp]
    lib.da_refine.argtypes = [vp, vp, i64, vp, i64, vp, vp, vp, vp, i32, C.c_double, vp, P(i64), P(i64)]
    lib.da_stats.argtypes = [vp, P(Stats)]
    lib.da_replace_segments.argtypes = [vp, vp, i64, vp, i64, i32, vp, vp, i32, i32]
    lib.da_stretch_resident.argtypes = [vp, vp, vp, i32, i32, vp, i64, vp]
    lib.da_stretch_schedule.argtypes = [vp, i32, vp, P(i64)]
    if lib.da_abi_version() != ABI_VERSION:
      raise ImportError("libdalign.so ABI version mismatch; rebuild it")
    _lib = lib
    return lib


def _ptr(a: np.ndarray):
  return a.ctypes.data_as(C.c_void_p)


class _Pinned:
  """Owner of one page-locked allocation; freed when the last array viewing it is collected."""

  def __init__(self, nbytes):
    self._lib = load()
    self.ptr = C.c_void_p()
    if self._lib.da_host_alloc(int(nbytes), C.byref(self.ptr)) != 0 or not self.ptr:
      raise MemoryError(f"da_host_alloc({nbytes}) failed")

  def __del__(self):
    try:
      if self.ptr:
        self._lib.da_host_free(self.ptr)
        self.ptr = C.c_void_p()
    except Exception:
      pass


def pinned_empty(shape, dtype=np.int16) -> np.ndarray:
  """A page-locked numpy array (da_host_alloc): the buffer a decoder fills for pcm_upload_async."""
  count = int(np.prod(shape))
  nbytes = max(1, count * np.dtype(dtype).itemsize)
  owner = _Pinned(nbytes)
  buf = (C.c_uint8 * nbytes).from_address(owner.ptr.value)
  buf._owner = owner                     # every numpy view keeps `buf` (its base) alive, and `buf` the allocation
  return np.frombuffer(buf, dtype=dtype, count=count).reshape(shape)


class PcmStream:
  """Streaming ingest handle (da_pcm_stream): owned by one decoder thread.  piece() enqueues the host->device
  copy of interleaved int16 frames (an (n, C) array, ideally a view of pinned_empty memory) and returns at once;
  the array must stay untouched until sync() has returned.  Context.pcm_adopt takes the device buffer over."""

  def __init__(self, device: int, channels: int, frames_hint: int = 0):
    self._lib = load()
    self._h = C.c_void_p()
    rc = self._lib.da_pcm_stream_open(int(device), int(channels), int(frames_hint), C.byref(self._h))
    if rc != 0 or not self._h:
      raise RuntimeError(f"da_pcm_stream_open failed with code {rc} (needs an MI355X / gfx950 device)")
    self.device, self.channels = int(device), int(channels)
    self._keep = []

  def _check(self, rc):
    if rc != 0:
      msg = self._lib.da_pcm_stream_error(self._h).decode("utf-8", "replace")
      raise RuntimeError(msg if msg else f"da_pcm_stream error {rc}")

  def piece(self, frames: np.ndarray):
    frames = np.asarray(frames)
    if frames.dtype != np.int16 or not frames.flags.c_contiguous or frames.size % self.channels:
      raise ValueError("a piece is a C-contiguous int16 array of whole interleaved frames")
    self._check(self._lib.da_pcm_stream_piece(self._h, _ptr(frames), frames.size // self.channels))
    self._keep.append(frames)                       # alive until the copy has left it

  def sync(self):
    self._check(self._lib.da_pcm_stream_sync(self._h))
    self._keep.clear()

  @property
  def frames(self) -> int:
    return int(self._lib.da_pcm_stream_frames(self._h))

  def close(self):
    if self._h:
      self._lib.da_pcm_stream_close(self._h)
      self._h = C.c_void_p()
      self._keep.clear()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass


def chain_host(i, v, q, min_len: float = 0.0):
  """da_chain without a device context (host-only DP) -- usable from CPU worker processes."""
  lib = load()
  i = np.ascontiguousarray(i, dtype=np.int32); v = np.ascontiguousarray(v, dtype=np.int32)
  q = np.ascontiguousarray(q, dtype=np.float64)
  n = len(i)
  pi = np.empty(max(n, 1), dtype=np.int32); pv = np.empty(max(n, 1), dtype=np.int32)
  m = C.c_int64(n)
  rc = lib.da_chain(None, _ptr(i), _ptr(v), _ptr(q), n, float(min_len), _ptr(pi), _ptr(pv), C.byref(m))
  if rc == ERR_MISMATCH:
    raise RuntimeError("Alignment failed, are the input files mismatched?")
  if rc != 0:
    raise RuntimeError(f"da_chain failed with code {rc}")
  return pi[:m.value].copy(), pv[:m.value].copy()


class Context:
  """One da_ctx: one GPU, one HIP stream.  Not thread-safe; use one per thread."""

  def __init__(self, device: int = 0, precision: int = PREC_F32):
    self._lib = load()
    self._h = C.c_void_p()
    rc = self._lib.da_create(int(device), int(precision), C.byref(self._h))
    if rc != 0:
      raise RuntimeError(f"da_create(device={device}) failed with code {rc}: no usable gfx950 (MI355X) device; "
                         "describealign_amd has no CPU fallback")
    self.precision = precision
    self.device = device
    self._n = {}
    self._channels = {}
    self._rows = {}
    self._inflight = {}
    self._pool = []      # recycled result buffers (see _recycled)

  def close(self):
    if getattr(self, "_h", None) is not None and self._h:
      self._lib.da_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  def __enter__(self):
    return self

  def __exit__(self, *exc):
    self.close()

  def _check(self, rc):
    if rc != 0:
      msg = self._lib.da_last_error(self._h).decode("utf-8", "replace")
      raise RuntimeError(msg if msg else f"libdalign error {rc}")

  # ---- features ----------------------------------------------------------------------------
  @staticmethod
  def _pcm_layout(pcm: np.ndarray, allow_copy: bool):
    """(array to hand to the library, frames, channels, planar) for int16 PCM given as (C, N) planar (the
    reference's array layout), as C-contiguous (N, C) interleaved frames, or as the (C, N) transposed VIEW of
    such frames (what media.parse_audio_from_file returns: the decoder's own layout, passed on without a copy)."""
    pcm = np.asarray(pcm)
    if pcm.dtype != np.int16 or pcm.ndim != 2:
      raise ValueError("PCM must be a 2-D int16 array")
    if pcm.shape[0] in (1, 2):
      channels, n = pcm.shape
      if pcm.flags.c_contiguous:
        return pcm, n, channels, 1
      if pcm.T.flags.c_contiguous:
        return pcm.T, n, channels, 0
    elif pcm.shape[1] in (1, 2):
      n, channels = pcm.shape
      if pcm.flags.c_contiguous:
        return pcm, n, channels, 0
    else:
      raise ValueError("PCM must have 1 or 2 channels")
    if not allow_copy:
      raise ValueError("PCM must be C-contiguous (planar (C, N), interleaved (N, C), or the transpose of either)")
    if pcm.shape[0] in (1, 2):
      return np.ascontiguousarray(pcm), pcm.shape[1], pcm.shape[0], 1
    return np.ascontiguousarray(pcm), pcm.shape[0], pcm.shape[1], 0

  def pcm_upload(self, side: int, pcm: np.ndarray):
    """pcm: int16 (C, N) planar (the reference's array layout) or (N, C) interleaved if
    given as a C-contiguous (N, C) array with C in {1, 2} and N > 2."""
    arr, n, channels, planar = self._pcm_layout(pcm, allow_copy=True)
    self._check(self._lib.da_pcm_upload(self._h, side, _ptr(arr), n, channels, planar))
    self._n[side] = n
    self._channels[side] = channels
    self._rows.pop(side, None)
    return n, channels

  def pcm_adopt(self, side: int, stream: "PcmStream"):
    """Take over the device buffer a PcmStream has filled (no copy); the side's next features_resident waits
    for the last piece on the device.  The stream is left empty, ready for the next file."""
    n, channels = stream.frames, stream.channels
    self._check(self._lib.da_pcm_adopt(self._h, side, stream._h))
    self._inflight[side] = list(stream._keep)        # pieces still in flight stay alive until the features call
    stream._keep.clear()
    self._n[side] = n
    self._channels[side] = channels
    self._rows.pop(side, None)
    return n, channels

  def pcm_exchange(self, side: int, stream: "PcmStream"):
    """pcm_adopt that leaves `stream` holding what this side held (buffer and frame count): resident files rotate through one
    context without a copy.  The side must be empty or have come from a stream itself (interleaved)."""
    n, channels = stream.frames, stream.channels
    self._check(self._lib.da_pcm_exchange(self._h, side, stream._h))
    self._inflight[side] = list(stream._keep)
    stream._keep.clear()
    self._n[side] = n
    self._channels[side] = channels
    self._rows.pop(side, None)
    return n, channels

  def pcm_upload_async(self, side: int, pcm: np.ndarray):
    """Enqueue the host->device copy of `pcm` (int16 (C, N) or (N, C), ideally page-locked: see
    pinned_empty) and return at once; the side's next features_resident waits for it on the device.
    `pcm` must stay alive and unchanged until that call has returned (a reference is kept here)."""
    arr, n, channels, planar = self._pcm_layout(pcm, allow_copy=False)
    self._check(self._lib.da_pcm_upload_async(self._h, side, _ptr(arr), n, channels, planar))
    self._inflight[side] = arr
    self._n[side] = n
    self._channels[side] = channels
    self._rows.pop(side, None)
    return n, channels

  def features_resident(self, side: int, download: bool = True):
    """Run the fused feature kernel on the PCM already resident for `side`.
    Returns the five rows (describealign.py:1101-1104) as float32 arrays, or None."""
    lengths = (C.c_int64 * 2)()
    if not download:
      self._check(self._lib.da_features_resident(self._h, side, None, 0, lengths))
      self._inflight.pop(side, None)
      return None
    le = ((self._n[side] // 105) + 1) // 2
    self._rows.pop(side, None)                   # the previous rows of this side: free for recycling once the caller drops them
    out = self._recycled((5, max(le, 1)), np.float32)
    self._check(self._lib.da_features_resident(self._h, side, _ptr(out), out.shape[1], lengths))
    self._inflight.pop(side, None)               # the asynchronous upload (if any) has been consumed
    le, lo = lengths[0], lengths[1]
    self._rows[side] = (out, le, lo)           # match_begin recognises these rows and skips their upload
    return [out[0, :le]] + [out[k, :lo] for k in range(1, 5)]      # row views of one buffer

  def pair_stage(self, mode: int = MATCH_HASHED):
    """The device stage of one pair of a batch in ONE native call (da_pair_stage): features of both resident PCM sides,
    matching, and the chain DP enqueued.  Returns (video rows, audio rows, number of matches, chain ticket); the rows are the
    lists features_resident() would return.  The interpreter lock is free for the other threads of a pipeline from the
    first kernel to the last."""
    outs, lens = [], []
    for side in (SIDE_VIDEO, SIDE_AUDIO):
      if side not in self._n:
        raise RuntimeError(f"pair_stage: no PCM uploaded for side {side}")
      le = ((self._n[side] // 105) + 1) // 2
      self._rows.pop(side, None)
      outs.append(self._recycled((5, max(le, 1)), np.float32))
      lens.append((C.c_int64 * 2)())
    n = C.c_int64(0); t = C.c_uint64(0)
    self._check(self._lib.da_pair_stage(self._h, _ptr(outs[0]), outs[0].shape[1], _ptr(outs[1]), outs[1].shape[1], mode,
                                        lens[0], lens[1], C.byref(n), C.byref(t)))
    rows = []
    for side, out, ln in zip((SIDE_VIDEO, SIDE_AUDIO), outs, lens):
      self._inflight.pop(side, None)
      self._rows[side] = (out, ln[0], ln[1])
      rows.append([out[0, :ln[0]]] + [out[k, :ln[1]] for k in range(1, 5)])
    self._pending_rows = None
    return rows[0], rows[1], n.value, t.value

  def _recycled(self, shape, dtype):
    """A page-locked result buffer of this shape that nobody holds any more, else a new one.  The pool keeps the flat base
    arrays; what is handed out (and every row view cut from it) has that base as its numpy `base`, so CPython's reference
    count of the base says exactly when the caller -- and whoever it passed rows to -- is done with it.
    Why pooled: a 2 h side's rows are 30 MB; allocated afresh per pair they arrive as new mmap'd pages, zeroed and faulted in
    under the process-wide address-space lock that every other thread of a batch pipeline needs for its own buffers.
    Why page-locked: the download is then one DMA on the context's stream; a copy to pageable memory is staged by the
    runtime through the null stream, which waits for every blocking stream of the device -- the CU-masked streams of the
    chain DPs (hipExtStreamCreateWithCUMask takes no flags) -- i.e. for the previous pair's DP."""
    import sys
    count = int(np.prod(shape))
    pool = self._pool
    # any free buffer at least this large will do (a directory's files all differ in length: exact sizes would never be
    # reused), the tightest first and none more than a quarter too large
    best = None
    for base in pool:
      if base.dtype == dtype and count <= base.size <= count + count // 4 + 65536 and sys.getrefcount(base) == 3:    # the pool, `base`, getrefcount's argument
        if best is None or base.size < best.size:
          best = base
    if best is not None:
      return best[:count].reshape(shape)
    # new buffers are rounded up (6 % + to a multiple of 64 Ki elements), so that the next file of about this length fits too
    room = ((count + count // 16 + 65535) // 65536) * 65536
    base = pinned_empty((room,), dtype)
    base = base.base if isinstance(base.base, np.ndarray) else base     # the flat array every view's `base` collapses to
    pool.append(base)
    # The pool has to hold every buffer a batch pipeline keeps in flight (window x 2 sides: ~100), or each pair allocates two
    # page-locked buffers and frees two -- and hipHostFree waits for the whole device to go idle, with the runtime's lock held:
    # the GPU-feeding thread's next launch then sits behind the chain DP in flight (measured: 40 % of a configs[1] batch's GEMMs
    # started 13 ms late, profiles/r05_pipeline_stalls.txt).  So nothing is dropped below a byte budget; above it only as many
    # FREE buffers go as it takes to get back under it (the ones that fit the current request worst first), never all at once.
    budget = int(os.environ.get("DALIGN_ROW_POOL_BYTES", str(8 << 30)))
    total = sum(b.nbytes for b in pool)
    if total > budget:
      free = [b for b in pool if b is not base and sys.getrefcount(b) == 3]     # the pool, `b`, getrefcount's argument
      free.sort(key=lambda b: -abs(b.size - room))
      for b in free:
        if total <= budget:
          break
        pool[:] = [q for q in pool if q is not b]
        total -= b.nbytes
    return base[:count].reshape(shape)

  def features(self, pcm: np.ndarray, side: int = SIDE_VIDEO):
    """Upload + feature kernel: the five feature rows as a list of float32 arrays."""
    self.pcm_upload(side, pcm)
    return self.features_resident(side)

  # ---- matching ----------------------------------------------------------------------------
  @staticmethod
  def _pack_rows(feats):
    le = len(feats[0]); lo = len(feats[1])
    if any(len(f) != lo for f in feats[1:]) or not (le == lo or le == lo + 1):
      raise ValueError("feature rows must have lengths (L or L+1, L, L, L, L)")
    rows = np.zeros((5, max(le, 1)), dtype=np.float32)
    for k, f in enumerate(feats):
      rows[k, :len(f)] = f
    return rows, (C.c_int64 * 2)(le, lo)

  def _resident_rows(self, side, feats):
    """(row buffer, lengths) if `feats` are the untouched row views features_resident() returned
    last for `side` on this context, else None."""
    held = self._rows.get(side)
    if held is None or len(feats) != 5:
      return None
    out, le, lo = held
    for k, f in enumerate(feats):
      # (a row view's `base` is the pooled flat buffer `out` itself is a view of: numpy collapses view chains)
      if not isinstance(f, np.ndarray) or (f.base is not out and f.base is not out.base) or f.dtype != np.float32 \
         or len(f) != (le if k == 0 else lo) or f.ctypes.data != out[k].ctypes.data:
        return None
    return out, (C.c_int64 * 2)(le, lo)

  def match_begin(self, video_features, audio_features, mode: int = MATCH_HASHED, rows=None):
    """Enqueue preparation + the similarity GEMM for one pair and return without waiting."""
    res = self._resident_rows(SIDE_VIDEO, video_features), self._resident_rows(SIDE_AUDIO, audio_features)
    if res[0] is not None and res[1] is not None:
      # the rows are the ones features_resident() just produced on this context: they are still on
      # the device, so neither re-packed nor uploaded
      (vrows, vlen), (arows, alen) = res
      mode |= MATCH_RESIDENT_ROWS
    else:
      vrows, vlen = self._pack_rows(video_features)
      arows, alen = self._pack_rows(audio_features)
    rb, re = (0, -1) if rows is None else rows
    self._pending_rows = (vrows, arows, vlen, alen)          # must outlive the asynchronous work
    self._check(self._lib.da_match_begin(self._h, _ptr(vrows), vrows.shape[1], vlen, _ptr(arows), arows.shape[1], alen,
                                         mode, rb, re))

  def match_finish(self) -> int:
    """Wait for the GEMM, verify + sort on the device; returns the number of matches (resident)."""
    n = C.c_int64(0)
    self._check(self._lib.da_match_finish(self._h, C.byref(n)))
    self._pending_rows = None
    return n.value

  def match_fetch(self, k: int, alloc=None):
    """Copy the k resident matches out (own copy stream: may overlap the next pair's GEMM)."""
    if alloc is None:
      oi = np.empty(k, dtype=np.int32); ov = np.empty(k, dtype=np.int32); oq = np.empty(k, dtype=np.float64)
    else:
      oi, ov, oq = alloc(k)          # caller-provided buffers (e.g. shared memory for a worker process)
    if k:
      self._check(self._lib.da_match_fetch(self._h, _ptr(oi), _ptr(ov), _ptr(oq), k))
    return oi, ov, oq

  def match(self, video_features, audio_features, mode: int = MATCH_HASHED, rows=None, capacity=None, alloc=None):
    """Verified matches (i, v, qual) sorted by (i, v) -- describealign.py:595-673."""
    self.match_begin(video_features, audio_features, mode, rows)
    return self.match_fetch(self.match_finish(), alloc)

  def match_corr(self, i, v):
    i = np.ascontiguousarray(i, dtype=np.int32); v = np.ascontiguousarray(v, dtype=np.int32)
    out = np.empty((len(i), 3), dtype=np.float32)
    self._check(self._lib.da_match_corr(self._h, _ptr(i), _ptr(v), len(i), _ptr(out)))
    return out

  def trim(self):
    """Give back the matching stage's scratch memory (tens of GB for a long pair); the resident match
    list, PCM and feature rows stay."""
    self._check(self._lib.da_trim(self._h))

  def match_export_device(self, d_keys: int, d_q: int, n: int):
    """Copy the n resident matches device-to-device into caller-owned device buffers (addresses):
    uint64 keys (i << 32 | v) and float64 qualities."""
    self._check(self._lib.da_match_export_device(self._h, C.c_void_p(d_keys), C.c_void_p(d_q), int(n)))

  def match_import_device(self, d_keys: int, d_q: int, n: int):
    """Make a sorted match list that lives in device memory the resident result of this context
    (chain_begin / chain_resident then run on it)."""
    self._check(self._lib.da_match_import_device(self._h, C.c_void_p(d_keys), C.c_void_p(d_q), int(n)))

  def match_import_reserve(self, n: int):
    """Device arrays (keys int64 address, qualities float64 address) for n matches that a gather fills in place;
    match_import_commit(m) makes the first m the resident list.  The list resident so far stays exportable until then."""
    k, q = C.c_void_p(), C.c_void_p()
    self._check(self._lib.da_match_import_reserve(self._h, int(n), C.byref(k), C.byref(q)))
    return int(k.value or 0), int(q.value or 0)

  def match_import_commit(self, n: int):
    self._check(self._lib.da_match_import_commit(self._h, int(n)))

  def match_dump_tile(self, video_tile: int, audio_tile: int):
    """Raw MFMA accumulators (divided by their scale: 1 - corr_j for f32, 1 - guard - corr_j for bf16) of one 32 x 32 tile of the last match:
    (acc[3, 32 rows, 32 cols] float32, video_frames[32], audio_frames[32])."""
    acc = np.empty((3, 32, 32), dtype=np.float32)
    vfr = np.empty(32, dtype=np.int32); afr = np.empty(32, dtype=np.int32)
    self._check(self._lib.da_match_dump_tile(self._h, int(video_tile), int(audio_tile), _ptr(acc), _ptr(vfr), _ptr(afr)))
    return acc, vfr, afr

  def chain(self, i, v, q, min_len: float = 0.0):
    """Heaviest non-decreasing chain -- describealign.py:654-698.  Returns (path_i, path_v)."""
    i = np.ascontiguousarray(i, dtype=np.int32); v = np.ascontiguousarray(v, dtype=np.int32)
    q = np.ascontiguousarray(q, dtype=np.float64)
    n = len(i)
    pi = np.empty(max(n, 1), dtype=np.int32); pv = np.empty(max(n, 1), dtype=np.int32)
    m = C.c_int64(n)
    self._check(self._lib.da_chain(self._h, _ptr(i), _ptr(v), _ptr(q), n, float(min_len), _ptr(pi), _ptr(pv), C.byref(m)))
    return pi[:m.value].copy(), pv[:m.value].copy()

  def chain_begin(self, exclusive: bool = False) -> int:
    """Hand the resident matches of the last match()/match_finish() to the device chain DP and
    enqueue it on its own stream; returns a ticket for chain_finish.  The context is free for the
    next match_begin at once.  exclusive: the caller waits for this DP before it launches anything else
    (da_chain_begin_exclusive: the DP is not confined to the few CUs per XCD that DPs beside a GEMM get)."""
    t = C.c_uint64(0)
    self._check((self._lib.da_chain_begin_exclusive if exclusive else self._lib.da_chain_begin)(self._h, C.byref(t)))
    return t.value

  def chain_finish(self, ticket: int, min_len: float = 0.0):
    """Wait for that DP; returns (path_i, path_v).  Raises the reference's mismatch error when
    the path is shorter than min_len (:698)."""
    m = C.c_int64(0)
    rc = self._lib.da_chain_finish(self._h, C.c_uint64(ticket), float(min_len), None, None, C.byref(m))
    if rc == 0:
      return np.empty(0, dtype=np.int32), np.empty(0, dtype=np.int32)
    if rc != ERR_CAPACITY:
      if rc == ERR_MISMATCH:
        raise RuntimeError("Alignment failed, are the input files mismatched?")
      self._check(rc)
    pi = np.empty(m.value, dtype=np.int32); pv = np.empty(m.value, dtype=np.int32)
    self._check(self._lib.da_chain_finish(self._h, C.c_uint64(ticket), float(min_len), _ptr(pi), _ptr(pv), C.byref(m)))
    return pi[:m.value], pv[:m.value]

  def chain_done(self, ticket: int) -> bool:
    """True once chain_finish(ticket) will not block."""
    rc = self._lib.da_chain_poll(self._h, C.c_uint64(ticket))
    if rc < 0:
      self._check(rc)
    return rc == 1

  def chain_masked(self) -> bool:
    """True when the streams of chain_begin()'s DPs are confined to their CU mask (see da_chain_masked)."""
    return self._lib.da_chain_masked(self._h) == 1

  def chain_resident(self, min_len: float = 0.0):
    """Stage-2 chain DP on the matches still resident from the last match (describealign.py:654-698)."""
    return self.chain_finish(self.chain_begin(exclusive=True), min_len)

  def refine(self, a_scaled, v_scaled, cl_x0, cl_x1, cl_offset, cl_slope, min_len: float = 0.0):
    """Banded line extension + second DP -- describealign.py:895-993.  Returns (path[M,5], n_points)."""
    a = np.ascontiguousarray(a_scaled, dtype=np.float64); v = np.ascontiguousarray(v_scaled, dtype=np.float64)
    x0 = np.ascontiguousarray(cl_x0, dtype=np.float64); x1 = np.ascontiguousarray(cl_x1, dtype=np.float64)
    off = np.ascontiguousarray(cl_offset, dtype=np.float64); sl = np.ascontiguousarray(cl_slope, dtype=np.float64)
    cap = len(a) + len(v) + 16
    for _ in range(2):
      path = getattr(self, "_refine_buf", None)    # the capacity buffer is this context's own and is never handed out
      if path is None or len(path) < cap:
        path = self._refine_buf = np.empty((cap, 5), dtype=np.float64)
      rows = C.c_int64(len(path)); npts = C.c_int64(0)
      rc = self._lib.da_refine(self._h, _ptr(a), len(a), _ptr(v), len(v), _ptr(x0), _ptr(x1), _ptr(off),
                               _ptr(sl), len(x0), float(min_len), _ptr(path), C.byref(rows), C.byref(npts))
      if rc == ERR_CAPACITY:
        cap = rows.value + 16
        continue
      self._check(rc)
      return path[:rows.value].copy(), npts.value
    raise RuntimeError("da_refine: capacity negotiation failed")

  # ---- audio replacement (--stretch_audio) ---------------------------------------------------
  @staticmethod
  def _nodes(audio_times, video_times):
    at = np.ascontiguousarray(audio_times, dtype=np.float64); vt = np.ascontiguousarray(video_times, dtype=np.float64)
    if at.ndim != 1 or at.shape != vt.shape:
      raise ValueError("audio_times and video_times must be 1-D arrays of equal length")
    return at, vt

  def replace_segments(self, video_arr, audio_desc_arr, audio_desc_times, video_times, no_pitch_correction=False):
    """replace_aligned_segments (describealign.py:230-416): float16 (C, N) arrays, `video_arr`
    is modified in place like the reference's."""
    if video_arr.dtype != np.float16 or audio_desc_arr.dtype != np.float16 or video_arr.ndim != 2 or \
       audio_desc_arr.ndim != 2 or video_arr.shape[0] != audio_desc_arr.shape[0]:
      raise ValueError("video_arr and audio_desc_arr must be float16 (C, N) arrays with equal C")
    if not video_arr.flags.c_contiguous:
      raise ValueError("video_arr must be C-contiguous (it is updated in place)")
    aud = np.ascontiguousarray(audio_desc_arr)
    at, vt = self._nodes(audio_desc_times, video_times)
    self._check(self._lib.da_replace_segments(self._h, _ptr(video_arr), video_arr.shape[1], _ptr(aud), aud.shape[1],
                                              video_arr.shape[0], _ptr(at), _ptr(vt), len(at), int(bool(no_pitch_correction))))

  def stretch_resident(self, audio_desc_times, video_times, no_pitch_correction=False):
    """The --stretch_audio block of combine() (describealign.py:1135-1153, :136) on the PCM uploaded
    with pcm_upload for both sides.  Returns (int16 (N, C) interleaved frames, loudness factors)."""
    at, vt = self._nodes(audio_desc_times, video_times)
    n = self._n[SIDE_VIDEO]
    fac = np.zeros(2, dtype=np.float64)
    out = np.empty((n, 2), dtype=np.int16)                  # sized for stereo; trimmed below
    self._check(self._lib.da_stretch_resident(self._h, _ptr(at), _ptr(vt), len(at), int(bool(no_pitch_correction)),
                                              _ptr(out), n, _ptr(fac)))
    ch = self._channels.get(SIDE_VIDEO, 2)
    return out.reshape(-1)[:n * ch].reshape(n, ch), fac[:ch]

  def stretch_schedules(self):
    """Jump schedules [(input index, signed distance)] of the stretched intervals of the last call."""
    n = C.c_int64(0)
    self._check(self._lib.da_stretch_schedule(self._h, -1, None, C.byref(n)))
    out = []
    for k in range(n.value):
      m = C.c_int64(0)
      rc = self._lib.da_stretch_schedule(self._h, k, None, C.byref(m))
      if rc not in (0, ERR_CAPACITY):
        self._check(rc)
      buf = np.empty((max(m.value, 1), 2), dtype=np.int64)
      cap = C.c_int64(m.value)
      self._check(self._lib.da_stretch_schedule(self._h, k, _ptr(buf), C.byref(cap)))
      out.append(buf[:cap.value].copy())
    return out

  def stats(self) -> dict:
    s = Stats()
    self._check(self._lib.da_stats(self._h, C.byref(s)))
    return s.as_dict()
