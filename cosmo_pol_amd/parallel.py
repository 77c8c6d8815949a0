"""Ray sharding over the GPUs of one node (one process per GPU, torch.distributed; backend 'nccl' is
RCCL on ROCm, 'gloo' on CPU for tests).  Rays are independent (radar_operator.py:253-256), so the only
communication is ONE gather of the output volume: every rank packs all output fields of its rays into rows
of bytes, rank 0 receives the rows and puts them at their ray positions."""
from collections import OrderedDict

import numpy as np


def shard_indices(n_sweeps, rays_per_sweep, rank, world):
    """Block-cyclic by ray index inside every sweep: rank r takes rays r, r + world, ... of each sweep,
    which balances precipitation sectors and the elevation-dependent ray lengths."""
    idx = [s * rays_per_sweep + i for s in range(n_sweeps) for i in range(rank, rays_per_sweep, world)]
    return np.array(idx, dtype=np.int64)


def all_shards(n_sweeps, rays_per_sweep, world):
    """The ray indices of every rank.  A pure function of the scan shape and the world size: every rank
    computes the same lists, so no rank has to ask another which rays it owns."""
    return [shard_indices(n_sweeps, rays_per_sweep, r, world) for r in range(world)]


def _is_dist(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


def pack_rays(fields):
    """fields: ordered {name: tensor [n_local, ...]} -> (uint8 tensor [n_local, row_bytes], layout).
    layout = [(name, dtype, trailing shape, byte offset, byte count)], identical on every rank."""
    import torch
    cols, layout, off = [], [], 0
    n = None
    for name, t in fields.items():
        n = t.shape[0] if n is None else n
        assert t.shape[0] == n, name
        b = t.contiguous().view(torch.uint8).reshape(n, -1) if n else \
            torch.empty((0, int(np.prod(t.shape[1:], dtype=np.int64)) * t.element_size()), dtype=torch.uint8, device=t.device)
        layout.append((name, t.dtype, tuple(t.shape[1:]), off, b.shape[1]))
        off += b.shape[1]
        cols.append(b)
    return torch.cat(cols, dim=1), layout


def unpack_rays(packed, layout):
    out = OrderedDict()
    n = packed.shape[0]
    for name, dtype, shape, off, nb in layout:
        out[name] = packed[:, off:off + nb].clone().view(dtype).reshape((n,) + shape)      # clone: a fresh, aligned storage
    return out


class VolumeGather(object):
    """One gather of all output fields to rank dst.  start() packs the local rays and issues the collective
    (asynchronously on request: the next scan's kernels then run beside it), finish() waits and returns
    {name: tensor [n_total, ...]} on rank dst, None elsewhere."""

    def __init__(self, shards, n_total, dst=0, group=None):
        self.shards, self.n_total, self.dst, self.group = shards, n_total, dst, group
        self.n_max = max(len(s) for s in shards)
        self.work = self.layout = self.bufs = self.pad = None

    def start(self, fields, async_op=False):
        import torch
        import torch.distributed as dist
        packed, self.layout = pack_rays(fields)
        if not _is_dist(self.group):
            self.bufs = [packed]
            return self
        rank = dist.get_rank(self.group)
        assert packed.shape[0] == len(self.shards[rank])
        if self.pad is None or self.pad.shape[1] != packed.shape[1]:
            self.pad = torch.zeros((self.n_max, packed.shape[1]), dtype=torch.uint8, device=packed.device)
            self.bufs = [torch.empty_like(self.pad) for _ in self.shards] if rank == self.dst else None
        self.pad[:packed.shape[0]] = packed
        self.work = dist.gather(self.pad, self.bufs, dst=self.dst, group=self.group, async_op=async_op)
        return self

    def finish(self):
        import torch
        import torch.distributed as dist
        if self.work is not None:
            self.work.wait()
            self.work = None
        if _is_dist(self.group) and dist.get_rank(self.group) != self.dst:
            return None
        dev = self.bufs[0].device
        rows = torch.empty((self.n_total, self.bufs[0].shape[1]), dtype=torch.uint8, device=dev)
        for idx, buf in zip(self.shards, self.bufs):
            rows[torch.as_tensor(idx, device=dev)] = buf[:len(idx)]
        return unpack_rays(rows, self.layout)


def gather_volume(fields, shards, n_total, dst=0, group=None):
    """Blocking form of VolumeGather."""
    return VolumeGather(shards, n_total, dst, group).start(fields).finish()


def gather_rays(local, shards, n_total, dst=0, group=None):
    """One per-ray array [n_local, ...] of every rank into [n_total, ...] on rank dst (None elsewhere).
    shards: the ray indices of every rank (all_shards); without a process group, [the local indices]."""
    out = gather_volume(OrderedDict(x=local), shards, n_total, dst, group)
    return None if out is None else out['x']


def any_rank(flag, device=None, group=None):
    """True on every rank if flag is true on any (one small all-reduce): lets all ranks raise together
    instead of leaving the others waiting in the gather."""
    import torch
    import torch.distributed as dist
    if not _is_dist(group):
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return bool(int(t.item()))


def run_rays_device(ctx, azimuth, elevation, device):
    """cpol_run_rays_dev into torch tensors on `device` (the rank's GPU): the gather then moves device memory
    over RCCL / xGMI without a detour through the host.  Returns {name: tensor [n_rays, ...]} per ray
    (obs as [n_rays, 9, n_gates]) and the ray status tensor."""
    import ctypes
    import torch
    from . import _lib
    az = torch.as_tensor(np.ascontiguousarray(azimuth, dtype=np.float64), device=device)
    el = torch.as_tensor(np.ascontiguousarray(elevation, dtype=np.float64), device=device)
    R, G = len(az), ctx.n_gates
    t = OrderedDict()
    t['obs'] = torch.empty((_lib.N_OBS, R, G), dtype=torch.float32, device=device)
    for k in ('rvel', 'mask', 'lat', 'lon'):
        t[k] = torch.empty((R, G), dtype=torch.float64, device=device)
    for k in ('dist', 'height'):
        t[k] = torch.empty((R, G), dtype=torch.float32, device=device)
    if ctx.doppler == 3:
        t['dspectrum'] = torch.empty((R, G, ctx.n_vel), dtype=torch.float64, device=device)
    status = torch.zeros((R,), dtype=torch.int32, device=device)
    P = lambda x, typ: ctypes.cast(x.data_ptr(), typ)
    o = _lib.CpolOutputs(P(t['obs'], _lib.c_float_p), P(t['rvel'], _lib.c_double_p), P(t['mask'], _lib.c_double_p),
                         P(t['lat'], _lib.c_double_p), P(t['lon'], _lib.c_double_p), P(t['dist'], _lib.c_float_p),
                         P(t['height'], _lib.c_float_p), P(status, _lib.c_int_p),
                         P(t['dspectrum'], _lib.c_double_p) if ctx.doppler == 3 else None)
    torch.cuda.synchronize(device)
    if R:
        ctx.run_rays_dev(az.data_ptr(), el.data_ptr(), R, o)
    ctx.synchronize()
    if not ctx.doppler:
        t['rvel'].fill_(float('nan'))
    t['obs'] = t['obs'].permute(1, 0, 2)          # per ray: [n_rays, 9, n_gates]
    return t, status
