"""CPU: ray sharding + output gather over torch.distributed (gloo, world_size 2).  The per-ray
compute is replaced by a deterministic stand-in so that the test needs no GPU; what is checked is
that every ray is computed exactly once and lands in its place of the gathered volume."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from cosmo_pol_amd import parallel  # noqa: E402


def test_shard_indices_partition():
    for n_sweeps, rps, world in [(20, 360, 8), (3, 7, 2), (1, 5, 8), (2, 4, 1)]:
        seen = np.concatenate([parallel.shard_indices(n_sweeps, rps, r, world) for r in range(world)])
        assert sorted(seen.tolist()) == list(range(n_sweeps * rps))
        counts = [len(parallel.shard_indices(n_sweeps, rps, r, world)) for r in range(world)]
        assert max(counts) - min(counts) <= n_sweeps


def _worker(rank, world, port, n_sweeps, rps, n_gates, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        n_total = n_sweeps * rps
        idx = parallel.shard_indices(n_sweeps, rps, rank, world)
        # stand-in for cpol_run_rays: value = 1000 * ray + gate
        local = torch.tensor(idx[:, None] * 1000. + np.arange(n_gates)[None, :], dtype=torch.float32)
        out = parallel.gather_rays(local, idx, n_total)
        local64 = torch.tensor(idx[:, None] * 1.0 + 0 * np.arange(n_gates)[None, :], dtype=torch.float64)
        out64 = parallel.gather_rays(local64, idx, n_total)
        if rank == 0:
            want = np.arange(n_total)[:, None] * 1000. + np.arange(n_gates)[None, :]
            ok = bool(np.array_equal(out.numpy(), want.astype(np.float32))) and \
                bool(np.array_equal(out64.numpy()[:, 0], np.arange(n_total, dtype=np.float64)))
            q.put(ok)
        else:
            assert out is None and out64 is None
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n_sweeps,rps', [(3, 7), (2, 8)])
def test_gather_two_ranks_gloo(n_sweeps, rps):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_sweeps, rps, 5, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


def test_gather_single_process():
    idx = np.array([2, 0, 1])
    out = parallel.gather_rays(torch.tensor([[2.], [0.], [1.]]), idx, 3)
    assert out[:, 0].tolist() == [0., 1., 2.]
