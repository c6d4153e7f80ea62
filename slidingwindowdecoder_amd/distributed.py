"""Multi-GPU sharding of the shot batch (one process per GPU, torch.distributed; backend "nccl" is
RCCL on ROCm, "gloo" is used by the CPU tests).

Shots are independent and the windows of one shot are sequentially dependent, so the batch is split
contiguously over ranks, every rank runs the whole sliding-window pipeline on its shard, and the
only collective is one all_gather of the per-shot decisions at the end (8 bytes per shot:
observable-flip mask + flagged bit).  No per-window or per-iteration exchange exists in the
algorithm, so none is invented here.
"""
from __future__ import annotations

import os

import numpy as np


def shard_bounds(num_shots: int, rank: int, world: int):
    """Contiguous [lo, hi) of `rank`; the first (num_shots % world) ranks get one extra shot."""
    base, extra = divmod(num_shots, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def gather_decisions(local, num_shots: int):
    """all_gather of per-shot rows (torch tensor [n_local, k]) -> tensor [num_shots, k] on every
    rank.  Shards may differ by one row; they are padded to a common length for the collective."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local  # (a one-rank process group still runs the collective: the same code path at every N)
    world = dist.get_world_size()
    if world == 1 and local.is_cuda and dist.get_backend() != "nccl":
        return local  # a one-rank group on a backend that cannot all_gather device tensors (gloo): nothing to exchange
    sizes = [shard_bounds(num_shots, r, world) for r in range(world)]
    nmax = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((nmax,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def decode_sharded(det_data: np.ndarray, decode_fn):
    """Decode this rank's contiguous shard of `det_data` [num_shots, num_det] with `decode_fn`
    (-> int32 array [n_local, 2] of per-shot decisions) and return the gathered [num_shots, 2]
    decisions on every rank.  `decode_fn` is the device pipeline in production
    (SlidingWindowDecoder.decode + last_obs_flips / last_flagged)."""
    import torch
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    lo, hi = shard_bounds(det_data.shape[0], rank, world)
    local = np.ascontiguousarray(decode_fn(det_data[lo:hi]), dtype=np.int32).reshape(hi - lo, -1)
    t = torch.from_numpy(local)
    if dist.is_initialized() and dist.get_backend() == "nccl":
        t = t.cuda()
    return gather_decisions(t, det_data.shape[0]).cpu().numpy()
