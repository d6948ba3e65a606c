"""One-process-per-GPU plumbing for batch sharding (torch.distributed: RCCL on GPUs, gloo on CPU).

The alignment path has no data-path collective: pairs are independent, so ranks only meet
for the start/stop barriers and the max-over-ranks of the elapsed time that bench.py reports.
"""
from __future__ import annotations

import os


class Group:
  def __init__(self, backend=None):
    self.rank = int(os.environ.get("RANK", "0"))
    self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    self.world = int(os.environ.get("WORLD_SIZE", "1"))
    self.backend = backend
    self.dist = None
    self.device = None
    if self.world > 1:
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

  def all_gather_matches(self, mi, mv, mq):
    """The one exchange step of a single long pair tiled over ranks: every
    rank matched its own contiguous block of audio rows; gather counts, then the padded (i, v, q)
    lists, and concatenate in rank order -- which is (i, v) order, because blocks are contiguous
    and each rank's list is already sorted.  With RCCL the tensors live in HBM and travel over
    xGMI; with gloo (tests) they are host tensors."""
    import numpy as np
    if self.dist is None:
      return mi, mv, mq
    import torch
    dev = self.device
    n = torch.tensor([len(mi)], dtype=torch.int64, device=dev)
    counts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(self.world)]
    self.dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    cap = max(1, max(counts))
    # one int64 payload: i, v and the float64 quality bit pattern, padded to the largest count
    buf = np.zeros((3, cap), dtype=np.int64)
    buf[0, :len(mi)] = mi; buf[1, :len(mv)] = mv
    buf[2, :len(mq)] = np.ascontiguousarray(mq, dtype=np.float64).view(np.int64)
    mine = torch.from_numpy(buf).to(dev)
    parts = [torch.empty_like(mine) for _ in range(self.world)]
    self.dist.all_gather(parts, mine)
    out_i, out_v, out_q = [], [], []
    for r, c in enumerate(counts):
      a = parts[r].cpu().numpy()
      out_i.append(a[0, :c].astype(np.int32)); out_v.append(a[1, :c].astype(np.int32))
      out_q.append(a[2, :c].copy().view(np.float64))
    return np.concatenate(out_i), np.concatenate(out_v), np.concatenate(out_q)

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
