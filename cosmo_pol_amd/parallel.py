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


SINGLE_RANK_COLLECTIVES = False      # rehearsal switch: issue the collective even in a group of one rank


def _is_dist(group=None):
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or SINGLE_RANK_COLLECTIVES)


def _row_layout(fields):
    """[(name, dtype, trailing shape, byte offset, byte count)] of one ray's row; every field starts on a multiple of
    eight bytes so that a column range of the row buffer can be viewed as the field's dtype."""
    layout, off = [], 0
    for name, t in fields.items():
        nb = int(np.prod(t.shape[1:], dtype=np.int64)) * t.element_size()
        layout.append((name, t.dtype, tuple(t.shape[1:]), off, nb))
        off += (nb + 7) & ~7
    return layout, off


def _field_view(rows, dtype, shape, off, nb):
    """The field stored in columns [off, off + nb) of a [n, row_bytes] uint8 buffer, as a (strided) [n, ...] view."""
    n = rows.shape[0]
    return rows[:, off:off + nb].view(dtype).reshape((n,) + tuple(shape))


def pack_rays(fields, out=None):
    """fields: ordered {name: tensor [n_local, ...]} -> (uint8 tensor [n_local, row_bytes], layout).  With `out` (a
    [>= n_local, row_bytes] buffer) the fields are written into its first rows by strided copies, nothing is
    allocated.  layout is identical on every rank."""
    import torch
    layout, row_bytes = _row_layout(fields)
    n = next(iter(fields.values())).shape[0] if fields else 0
    if out is None:
        dev = next(iter(fields.values())).device
        out = torch.zeros((n, row_bytes), dtype=torch.uint8, device=dev)
    assert out.shape[1] == row_bytes and out.shape[0] >= n
    if n:
        for (name, dtype, shape, off, nb), t in zip(layout, fields.values()):
            assert t.shape[0] == n, name
            _field_view(out[:n], dtype, shape, off, nb).copy_(t)
    return out[:n], layout


def unpack_rays(packed, layout):
    """{name: [n, ...] view into `packed`} (strided: the rows hold all fields side by side)."""
    return OrderedDict((name, _field_view(packed, dtype, shape, off, nb)) for name, dtype, shape, off, nb in layout)


def mask_to_codes(mask, n_sub):
    """The output mask is the mean of the sub-beams' interpolation codes with values in [0, 1) set to 0
    (scattering_sz.py:144-151): sum / n_sub, sum an integer in [-n_sub, n_sub].  The sum travels as int8 / int16
    and decodes to the same float64, bit for bit."""
    import torch
    return torch.round(mask * n_sub).to(torch.int8 if n_sub <= 127 else torch.int16)


def codes_to_mask(codes, n_sub):
    """sum / n_sub with the values in [0, 1) set to 0, i.e. the sums 0 .. n_sub - 1: selected on the one-byte codes,
    so that the float64 volume is written once."""
    import torch
    keep = (codes < 0) | (codes >= n_sub)
    return (codes * keep).to(torch.float64) / n_sub


class VolumeGather(object):
    """One gather of the output fields to rank dst.  start() packs the local rays into rows of bytes and issues the
    collective (asynchronously on request: the next scan's kernels then run beside it); finish() waits, puts every
    rank's rows at their ray positions with one indexed copy into a volume buffer that lives as long as this object,
    and returns {name: tensor [n_total, ...]} on rank dst (views of that buffer, valid until the next finish()),
    None elsewhere.

    static: names of fields that depend on the scan geometry only (position of the gates).  They travel with the
    first volume and again whenever geometry_key differs from the previous call's; otherwise rank dst serves them
    from what it received before.  mask_sub: number of sub-beams; the float64 field 'mask' then travels as the
    integer sum of the sub-beam codes (mask_to_codes) and is decoded on rank dst."""

    def __init__(self, shards, n_total, dst=0, group=None, static=(), mask_sub=0):
        self.shards, self.n_total, self.dst, self.group = shards, n_total, dst, group
        self.n_max = max(len(s) for s in shards)
        self.static, self.mask_sub = tuple(static), int(mask_sub)
        self.work = self.layout = None
        self._key = object()              # geometry key of the static fields rank dst holds
        self._static_vol = None
        self._sets = {}                   # buffers per row layout: pad, receive buffer, volume rows, index
        self._cur = None
        self.assemble_stream = None
        self.assembled = None             # torch event: the volume of the last finish() is in place
        self.assemble_ms_events = None    # (begin, end) events around the last indexed copy (rank dst, GPU)

    def _buffers(self, row_bytes, device):
        import torch
        import torch.distributed as dist
        key = (row_bytes, str(device))
        if key not in self._sets:
            world = len(self.shards)
            is_dst = (not _is_dist(self.group)) or dist.get_rank(self.group) == self.dst
            b = {'pad': torch.zeros((self.n_max, row_bytes), dtype=torch.uint8, device=device)}
            if is_dst:
                b['recv'] = torch.zeros((world, self.n_max, row_bytes), dtype=torch.uint8, device=device)
                # rows beyond a rank's share go to a spare row behind the volume
                idx = np.full((world, self.n_max), self.n_total, dtype=np.int64)
                for r, s_ in enumerate(self.shards):
                    idx[r, :len(s_)] = s_
                b['idx'] = torch.as_tensor(idx.reshape(-1), device=device)
                b['rows'] = torch.zeros((self.n_total + 1, row_bytes), dtype=torch.uint8, device=device)
            self._sets[key] = b
        return self._sets[key]

    def start(self, fields, async_op=False, geometry_key=None, stream=None):
        """stream: a torch.cuda stream on which the fields are (or will be) complete; packing and the collective are
        queued behind it without a host synchronisation.  None: the current stream."""
        import contextlib
        import torch
        import torch.distributed as dist
        with_static = bool(self.static) and (geometry_key is None or geometry_key != self._key)
        self._sent_static = with_static
        if with_static:                   # the same decision on every rank: it follows from the caller's key alone
            self._key = geometry_key if geometry_key is not None else object()
        dev = next(iter(fields.values())).device
        ctx = torch.cuda.stream(stream) if (stream is not None and dev.type == 'cuda') else contextlib.nullcontext()
        with ctx:
            if self.assembled is not None and dev.type == 'cuda':
                # the buffers of this object are free again once its previous volume has been put in place
                torch.cuda.current_stream(dev).wait_event(self.assembled)
            send = OrderedDict()
            for name, t in fields.items():
                if name in self.static and not with_static:
                    continue
                send[name] = mask_to_codes(t, self.mask_sub) if (name == 'mask' and self.mask_sub) else t
            layout, row_bytes = _row_layout(send)
            b = self._buffers(row_bytes, dev)
            self.layout, self._cur = layout, b
            n = next(iter(send.values())).shape[0]
            pack_rays(send, out=b['pad'])
            if not _is_dist(self.group):
                b['recv'][0].copy_(b['pad'])
                return self
            rank = dist.get_rank(self.group)
            assert n == len(self.shards[rank])
            recv = [b['recv'][r] for r in range(len(self.shards))] if rank == self.dst else None
            self.work = dist.gather(b['pad'], recv, dst=self.dst, group=self.group, async_op=async_op)
        return self

    def finish(self):
        import contextlib
        import torch
        import torch.distributed as dist
        b = self._cur
        is_dst = (not _is_dist(self.group)) or dist.get_rank(self.group) == self.dst
        dev = b['pad'].device
        side = None
        if dev.type == 'cuda':
            if self.assemble_stream is None:
                self.assemble_stream = torch.cuda.Stream(dev)
            side = self.assemble_stream
            side.wait_stream(torch.cuda.current_stream(dev))
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            if self.work is not None:
                self.work.wait()              # on a GPU: the assemble stream waits for the collective, the host does not
                self.work = None
            if not is_dst:
                if side is not None:
                    self.assembled = torch.cuda.Event()
                    self.assembled.record(side)
                return None
            if side is not None:
                t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0.record(side)
            # rows are whole multiples of eight bytes (_row_layout): moved as 64-bit words, an eighth of the elements
            # a byte-wise indexed copy would handle
            b['rows'].view(torch.int64).index_copy_(0, b['idx'], b['recv'].view(-1, b['recv'].shape[-1]).view(torch.int64))
            if side is not None:
                t1.record(side)
                self.assemble_ms_events = (t0, t1)
            vol = unpack_rays(b['rows'][:self.n_total], self.layout)
            if self.mask_sub and 'mask' in vol:
                vol['mask'] = codes_to_mask(vol['mask'], self.mask_sub)
            if self.static:
                if self._sent_static:
                    self._static_vol = OrderedDict((k, vol[k].clone()) for k in self.static if k in vol)
                for k, v in (self._static_vol or {}).items():
                    vol[k] = v
            if side is not None:
                self.assembled = torch.cuda.Event()
                self.assembled.record(side)
        return vol

    def wait(self):
        """Host wait until the volume returned by the last finish() is in place (GPU; a no-op on the CPU)."""
        if self.assembled is not None:
            self.assembled.synchronize()


def gather_volume(fields, shards, n_total, dst=0, group=None, mask_sub=0):
    """Blocking form of VolumeGather."""
    g = VolumeGather(shards, n_total, dst, group, mask_sub=mask_sub).start(fields)
    vol = g.finish()
    g.wait()
    return vol


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


def run_rays_device(ctx, azimuth, elevation, device, diag=False):
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
    if diag:
        t['diag'] = torch.empty((len(ctx.var_names), R, G), dtype=torch.float64, device=device)
    status = torch.zeros((R,), dtype=torch.int32, device=device)
    P = lambda x, typ: ctypes.cast(x.data_ptr(), typ)
    o = _lib.CpolOutputs(P(t['obs'], _lib.c_float_p), P(t['rvel'], _lib.c_double_p), P(t['mask'], _lib.c_double_p),
                         P(t['lat'], _lib.c_double_p), P(t['lon'], _lib.c_double_p), P(t['dist'], _lib.c_float_p),
                         P(t['height'], _lib.c_float_p), P(status, _lib.c_int_p),
                         P(t['dspectrum'], _lib.c_double_p) if ctx.doppler == 3 else None,
                         P(t['diag'], _lib.c_double_p) if diag else None)
    torch.cuda.synchronize(device)
    if R:
        ctx.run_rays_dev(az.data_ptr(), el.data_ptr(), R, o)
    ctx.synchronize()
    if not ctx.doppler:
        t['rvel'].fill_(float('nan'))
    t['obs'] = t['obs'].permute(1, 0, 2)          # per ray: [n_rays, 9, n_gates]
    if diag:
        t['diag'] = t['diag'].permute(1, 0, 2)    # per ray: [n_rays, n_vars, n_gates]
    return t, status
