"""One-process-per-GPU plumbing for batch sharding (torch.distributed: RCCL on GPUs, gloo on CPU).

The alignment path has no data-path collective: pairs are independent, so ranks only meet
for the start/stop barriers and the max-over-ranks of the elapsed time that bench.py reports.
"""
from __future__ import annotations

import os


class Group:
  def __init__(self, backend=None, init_single=False):
    """init_single: create the process group even for one rank (RANK / WORLD_SIZE / MASTER_* as usual), so
    that the collective code paths -- RCCL included -- run on a one-GPU box."""
    self.rank = int(os.environ.get("RANK", "0"))
    self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    self.world = int(os.environ.get("WORLD_SIZE", "1"))
    self.backend = backend
    self.dist = None
    self.device = None
    if self.world > 1 or init_single:
      import torch
      import torch.distributed as dist
      backend = backend or "nccl"
      if backend == "nccl":
        torch.cuda.set_device(self.local_rank)
        self.device = torch.device("cuda", self.local_rank)
        dist.init_process_group(backend="nccl", device_id=self.device)
      else:
        self.device = torch.device("cpu")
        dist.init_process_group(backend=backend)
      self.dist = dist
      self.backend = backend

  def group_size(self) -> int:
    """The process group's own world size (1 without a group): what actually runs, whatever the environment said."""
    return int(self.dist.get_world_size()) if self.dist is not None else 1

  def barrier(self):
    if self.dist is not None:
      self.dist.barrier()

  def all_ok(self, ok: bool) -> bool:
    """True when `ok` holds on every rank (one MIN all-reduce): lets one failing rank abort a collective
    step everywhere instead of leaving the others blocked in it."""
    if self.dist is None:
      return bool(ok)
    import torch
    t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=self.device)
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN)
    return bool(t.item() > 0.5)

  def max_over_ranks(self, value: float) -> float:
    if self.dist is None:
      return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=self.device)
    self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
    return float(t.item())

  def sum_over_ranks(self, value: float) -> float:
    if self.dist is None:
      return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=self.device)
    self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
    return float(t.item())

  def gather_matches_to_root(self, ctx, n_local):
    """The one exchange step of a single long pair tiled over ranks: every rank
    has matched its own contiguous block of audio rows and holds its sorted list on ITS device.
    The lists are gathered on rank 0 and become the resident match list of rank 0's context, in rank
    order -- which is (i, v) order, because the blocks are contiguous and each list is sorted.

    RCCL (backend "nccl"): the lists never touch the host -- exported device-to-device into the
    tensors RCCL sends over xGMI, imported device-to-device on rank 0 (16 B per match: ~1e9 matches of
    an 8 h pair = 16 GB into rank 0's 288 GB).  gloo (CPU tests): staged through host tensors.
    Returns the total number of matches on rank 0, None elsewhere."""
    import numpy as np
    import torch
    if self.dist is None:
      return n_local
    dist, dev = self.dist, self.device
    n = torch.tensor([n_local], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    root = self.rank == 0
    if self.backend == "nccl":
      return self._gather_device(ctx, n_local, counts)
    cap = max(1, max(counts))
    keys = torch.empty(cap, dtype=torch.int64, device=dev)
    qual = torch.empty(cap, dtype=torch.float64, device=dev)
    keys[n_local:] = 0; qual[n_local:] = 0                 # only the padding; the list itself is written below
    mi, mv, mq = ctx.match_fetch(n_local)
    keys[:n_local] = torch.from_numpy((mi.astype(np.int64) << 32) | mv.astype(np.int64))
    qual[:n_local] = torch.from_numpy(np.ascontiguousarray(mq, dtype=np.float64))
    got_k = [torch.empty_like(keys) for _ in range(self.world)] if root else None
    got_q = [torch.empty_like(qual) for _ in range(self.world)] if root else None
    dist.gather(keys, got_k, dst=0)
    dist.gather(qual, got_q, dst=0)
    if not root:
      return None
    all_k = torch.cat([got_k[r][:counts[r]] for r in range(self.world)])
    all_q = torch.cat([got_q[r][:counts[r]] for r in range(self.world)])
    total = int(all_k.numel())
    if hasattr(ctx, "match_import_device") and isinstance(getattr(ctx, "device", None), int) and torch.cuda.is_available():
      # gloo between processes that do have GPUs (tests, one-GPU emulation of a node): the gathered list goes
      # up in one copy and is imported device-to-device like the RCCL path's -- no per-match host work
      gdev = torch.device("cuda", ctx.device)
      dk = all_k.to(gdev); dq = all_q.to(gdev)
      torch.cuda.synchronize(gdev)
      ctx.match_import_device(dk.data_ptr(), dq.data_ptr(), total)
      del dk, dq
    else:
      k = all_k.numpy()
      # host lists: hand them to the context through a device upload of its own
      ctx._gathered = ((k >> 32).astype(np.int32), (k & 0xffffffff).astype(np.int32), all_q.numpy().copy())
    return total

  def _gather_device(self, ctx, n_local, counts):
    """RCCL: exact-size point-to-point transfers that LAND in rank 0's context.  Rank 0 reserves the arrays of
    its next resident match list (da_match_import_reserve), wraps them as tensors and posts one receive per
    rank into the slice that rank's block occupies -- rank order is (i, v) order -- while its own block is
    copied into the first slice on the device; the others export their list into tensors of exactly its
    length and send them.  Nothing is padded to the longest list, nothing is concatenated, nothing is copied a
    second time: an 8 h pair's 1.1e9 matches are 18 GB once, not three times.

    Every rank agrees on rank 0's reserve (an 18 GB allocation that can fail) and on its own export buffers
    BEFORE any transfer is posted: a failure on one rank is raised on all of them instead of leaving the others
    blocked in a send that nobody receives."""
    import torch
    dist, dev = self.dist, self.device
    offs = gather_offsets(counts)
    total = offs[-1]
    root = self.rank == 0
    err, all_k, all_q, keys, qual, pk, pq = None, None, None, None, None, 0, 0
    try:
      if root:
        pk, pq = ctx.match_import_reserve(total)
        all_k = _device_view(pk, max(1, total), "<i8", torch.int64, dev)
        all_q = _device_view(pq, max(1, total), "<f8", torch.float64, dev)
      elif n_local:
        keys = torch.empty(n_local, dtype=torch.int64, device=dev)
        qual = torch.empty(n_local, dtype=torch.float64, device=dev)
    except Exception as e:                                  # noqa: BLE001 -- reported on every rank below
      err = e
    if not self.all_ok(err is None):
      if err is not None:
        raise err
      raise RuntimeError("gather of the match lists: another rank could not provide its buffers")
    if root:
      ops = [dist.P2POp(dist.irecv, t[offs[r]:offs[r + 1]], r)
             for r in range(1, self.world) if counts[r] for t in (all_k, all_q)]
      reqs = dist.batch_isend_irecv(ops) if ops else []
      if n_local:                                           # own block: device to device on the context's stream
        ctx.match_export_device(pk, pq, n_local)
      for q in reqs:
        q.wait()
      _sync_device(dev)
      ctx.match_import_commit(total)
      return total
    if n_local:
      # the context copies on ITS stream (non-blocking, no implicit order with torch's): the allocation must be complete
      _sync_device(dev)
      ctx.match_export_device(keys.data_ptr(), qual.data_ptr(), n_local)
      for q in dist.batch_isend_irecv([dist.P2POp(dist.isend, keys, 0), dist.P2POp(dist.isend, qual, 0)]):
        q.wait()
      _sync_device(dev)
    return None

  def agree_on(self, values):
    """Rank 0's list of integers on every rank (one broadcast); raises on every rank when a rank had derived
    different ones -- decisions that every rank computes from its own copy of the data (the row blocks of a tiled
    pair) must not differ by one float comparison."""
    if self.dist is None:
      return [int(v) for v in values]
    import torch
    mine = torch.tensor([int(v) for v in values], dtype=torch.int64, device=self.device)
    agreed = mine.clone()
    self.dist.broadcast(agreed, src=0)
    same = bool((mine == agreed).all().item())
    if not self.all_ok(same):
      raise RuntimeError("ranks derived different values for a decision they must share: "
                         f"rank {self.rank} has {mine.tolist()}, rank 0 has {agreed.tolist()}")
    return [int(v) for v in agreed.tolist()]

  def broadcast_result(self, result, error=None):
    """Rank 0's align() result -- (audio_times, video_times, similarity, path, median_slope) -- to every
    rank (rank 0 alone runs the sequential host stages of a tiled pair).  `error`: rank 0's failure
    message, raised as RuntimeError on every rank."""
    import numpy as np
    import torch
    if self.dist is None:
      if error:
        raise RuntimeError(error)
      return result
    dist, dev = self.dist, self.device
    head = torch.zeros(4, dtype=torch.float64, device=dev)
    if self.rank == 0:
      if error:
        head[0] = -1.0
      else:
        x, y, sim, path, med = result
        head = torch.tensor([len(x), path.shape[0], float(sim), float(med)], dtype=torch.float64, device=dev)
    dist.broadcast(head, src=0)
    if head[0].item() < 0:
      msg = [error]
      dist.broadcast_object_list(msg, src=0)
      raise RuntimeError(msg[0])
    nx, rows = int(head[0].item()), int(head[1].item())
    body = torch.empty(2 * nx + 5 * rows, dtype=torch.float64, device=dev)
    if self.rank == 0:
      flat = np.concatenate([np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64), np.ascontiguousarray(path, dtype=np.float64).ravel()])
      body.copy_(torch.from_numpy(flat))
    dist.broadcast(body, src=0)
    if self.rank == 0:
      return result
    flat = body.cpu().numpy()
    return flat[:nx].copy(), flat[nx:2 * nx].copy(), float(head[2].item()), flat[2 * nx:].reshape(rows, 5).copy(), float(head[3].item())

  def broadcast_rows(self, rows, meta=None):
    """Rank 0's list of float32 feature rows (and a small picklable `meta`) on every rank: two broadcasts (lengths + meta, then
    one flat tensor).  For a tiled pair's callers that have the PCM on rank 0 only: every rank needs all feature rows of both
    sides (120 MB a side for 8 h), none needs the PCM."""
    import numpy as np
    import torch
    if self.dist is None:
      return rows, meta
    head = [None]
    if self.rank == 0:
      head = [([len(r) for r in rows], meta)]
    self.dist.broadcast_object_list(head, src=0)
    lens, meta = head[0]
    flat = torch.empty(int(sum(lens)), dtype=torch.float32, device=self.device)
    if self.rank == 0:
      flat.copy_(torch.from_numpy(np.concatenate([np.asarray(r, dtype=np.float32) for r in rows])))
    self.dist.broadcast(flat, src=0)
    if self.rank == 0:
      return rows, meta
    host = flat.cpu().numpy()
    out, at = [], 0
    for n in lens:
      out.append(host[at:at + n].copy()); at += n
    return out, meta

  def close(self):
    if self.dist is not None:
      self.dist.destroy_process_group()
      self.dist = None


class _DeviceArray:
  """Library-owned device memory presented through __cuda_array_interface__ (torch.as_tensor aliases it, no copy)."""

  def __init__(self, ptr, n, typestr):
    self.__cuda_array_interface__ = dict(shape=(int(n),), typestr=typestr, data=(int(ptr), False), version=2)


def gather_offsets(counts):
  """Start of every rank's slice in the gathered list (rank order = (i, v) order), total last."""
  offs = [0]
  for c in counts:
    offs.append(offs[-1] + int(c))
  return offs


def _sync_device(device):
  import torch
  if getattr(device, "type", "cpu") == "cuda":
    torch.cuda.synchronize(device)


def _device_view(ptr, n, typestr, dtype, device):
  import torch
  t = torch.as_tensor(_DeviceArray(ptr, n, typestr), device=device)
  assert t.dtype == dtype and t.data_ptr() == ptr, "torch copied instead of aliasing the library's buffer"
  return t


def shard_pairs(n_pairs: int, world: int):
  """Round-robin assignment of pair indices to ranks/GPUs (no collectives needed)."""
  world = max(1, int(world))
  return [list(range(r, n_pairs, world)) for r in range(world)]


def row_blocks_by_load(active, world: int):
  """Contiguous [begin, end) audio-row blocks with about equal numbers of ACTIVE rows (non-quiet audio frames: the
  rows the matching stage works on, describealign.py:657-658) -- a programme with a silent hour would otherwise leave
  a rank idle.  `active`: boolean array over the rows.  Every row belongs to exactly one block."""
  import numpy as np
  world = max(1, int(world))
  active = np.asarray(active, dtype=bool)
  n = len(active)
  csum = np.concatenate([[0], np.cumsum(active, dtype=np.int64)])
  total = int(csum[-1])
  if total == 0:
    return row_blocks(n, world)
  cuts = [0]
  for r in range(1, world):
    target = (total * r + world - 1) // world               # first row index with at least `target` active rows before it
    cuts.append(max(cuts[-1], int(np.searchsorted(csum, target, side="left"))))
  cuts.append(n)
  return [(cuts[r], cuts[r + 1]) for r in range(world)]


def row_blocks(n_rows: int, world: int):
  """Contiguous [begin, end) audio-row blocks, one per rank, sizes differing by at most one."""
  world = max(1, int(world))
  base, extra = divmod(max(0, int(n_rows)), world)
  out, b = [], 0
  for r in range(world):
    e = b + base + (1 if r < extra else 0)
    out.append((b, e))
    b = e
  return out
