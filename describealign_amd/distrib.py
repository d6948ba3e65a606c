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
    cap = max(1, max(counts))
    on_gpu = self.backend == "nccl"
    keys = torch.empty(cap, dtype=torch.int64, device=dev)
    qual = torch.empty(cap, dtype=torch.float64, device=dev)
    keys[n_local:] = 0; qual[n_local:] = 0                 # only the padding; the list itself is written below
    if on_gpu:
      # the context copies on ITS stream (non-blocking, no implicit order with torch's): the tensors'
      # allocation and the tail fill above must have completed before it writes into them
      torch.cuda.current_stream(dev).synchronize()
      if n_local:
        ctx.match_export_device(keys.data_ptr(), qual.data_ptr(), n_local)
    else:
      mi, mv, mq = ctx.match_fetch(n_local)
      keys[:n_local] = torch.from_numpy((mi.astype(np.int64) << 32) | mv.astype(np.int64))
      qual[:n_local] = torch.from_numpy(np.ascontiguousarray(mq, dtype=np.float64))
    root = self.rank == 0
    got_k = [torch.empty_like(keys) for _ in range(self.world)] if root else None
    got_q = [torch.empty_like(qual) for _ in range(self.world)] if root else None
    dist.gather(keys, got_k, dst=0)
    dist.gather(qual, got_q, dst=0)
    if not root:
      return None
    all_k = torch.cat([got_k[r][:counts[r]] for r in range(self.world)])
    all_q = torch.cat([got_q[r][:counts[r]] for r in range(self.world)])
    total = int(all_k.numel())
    if on_gpu:
      torch.cuda.synchronize(dev)
      ctx.match_import_device(all_k.data_ptr(), all_q.data_ptr(), total)
    elif hasattr(ctx, "match_import_device") and isinstance(getattr(ctx, "device", None), int) and torch.cuda.is_available():
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

  def close(self):
    if self.dist is not None:
      self.dist.destroy_process_group()
      self.dist = None


def shard_pairs(n_pairs: int, world: int):
  """Round-robin assignment of pair indices to ranks/GPUs (no collectives needed)."""
  world = max(1, int(world))
  return [list(range(r, n_pairs, world)) for r in range(world)]


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
