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

  def close(self):
    if self.dist is not None:
      self.dist.destroy_process_group()
      self.dist = None


def shard_pairs(n_pairs: int, world: int):
  """Round-robin assignment of pair indices to ranks/GPUs (no collectives needed)."""
  world = max(1, int(world))
  return [list(range(r, n_pairs, world)) for r in range(world)]
