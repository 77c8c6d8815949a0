"""Ray sharding over the GPUs of one node (one process per GPU, torch.distributed; backend 'nccl' is
RCCL on ROCm, 'gloo' on CPU for tests).  Rays are independent (radar_operator.py:253-256), so the only
communication is one gather of the output volume."""
import numpy as np


def shard_indices(n_sweeps, rays_per_sweep, rank, world):
    """Block-cyclic by ray index inside every sweep: rank r takes rays r, r + world, ... of each sweep,
    which balances precipitation sectors and the elevation-dependent ray lengths."""
    idx = [s * rays_per_sweep + i for s in range(n_sweeps) for i in range(rank, rays_per_sweep, world)]
    return np.array(idx, dtype=np.int64)


def gather_rays(local, local_idx, n_total, dst=0, group=None):
    """Gather per-ray arrays [n_local, ...] of every rank into [n_total, ...] on rank dst (None elsewhere).
    `local` is a torch tensor on the rank's device (or CPU with gloo); ranks may own different counts."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        out[torch.as_tensor(local_idx, device=local.device)] = local
        return out
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    counts = [len(shard_counts) for shard_counts in _all_indices(local_idx, world, group)]
    n_max = max(counts)
    pad = torch.zeros((n_max,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    out = torch.empty((n_total,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r, idx in enumerate(_all_indices(local_idx, world, group)):
        out[torch.as_tensor(idx, device=local.device)] = bufs[r][:len(idx)]
    return out


_idx_cache = {}


def _all_indices(local_idx, world, group):
    import torch.distributed as dist
    key = (id(group), world, len(local_idx), int(local_idx[0]) if len(local_idx) else -1)
    if key not in _idx_cache:
        gathered = [None] * world
        dist.all_gather_object(gathered, np.asarray(local_idx).tolist(), group=group)
        _idx_cache[key] = [np.array(g, dtype=np.int64) for g in gathered]
    return _idx_cache[key]
