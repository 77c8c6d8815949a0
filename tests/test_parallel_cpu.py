"""CPU: ray sharding + output gather over torch.distributed (gloo, world_size 2).  The per-ray
compute is replaced by a deterministic stand-in so that the test needs no GPU; what is checked is
that every ray is computed exactly once and lands in its place of the gathered volume, for every
output field, through parallel.gather_volume and through RadarOperator itself."""
import os
import socket
import sys
from collections import OrderedDict

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
        shards = parallel.all_shards(n_sweeps, rps, world)
        seen = np.concatenate(shards)
        assert sorted(seen.tolist()) == list(range(n_sweeps * rps))
        counts = [len(s) for s in shards]
        assert max(counts) - min(counts) <= n_sweeps
        for r in range(world):
            np.testing.assert_array_equal(shards[r], parallel.shard_indices(n_sweeps, rps, r, world))


def test_pack_unpack_roundtrip():
    f = OrderedDict(a=torch.arange(24, dtype=torch.float32).reshape(4, 2, 3), b=torch.arange(4, dtype=torch.float64)[:, None] * 1.5,
                    c=torch.arange(8, dtype=torch.int32).reshape(4, 2))
    packed, layout = parallel.pack_rays(f)
    assert packed.dtype == torch.uint8 and packed.shape == (4, 2 * 3 * 4 + 8 + 2 * 4)
    back = parallel.unpack_rays(packed, layout)
    for k in f:
        assert back[k].dtype == f[k].dtype and torch.equal(back[k], f[k])
    # no rays at all (a rank that owns none)
    e = OrderedDict(a=torch.empty((0, 2, 3), dtype=torch.float32), b=torch.empty((0, 1), dtype=torch.float64))
    packed, layout = parallel.pack_rays(e)
    assert packed.shape == (0, 2 * 3 * 4 + 8)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _spawn(target, args, world=2):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, q) + args) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    return q.get(timeout=5)


def _fake_fields(idx, n_gates):
    """Stand-in for the kernels: every field is a function of (ray, gate) only."""
    idx = np.asarray(idx, dtype=np.float64)
    g = np.arange(n_gates)[None, :]
    return OrderedDict(
        obs=torch.tensor((idx[:, None, None] * 1000. + np.arange(9)[None, :, None] * 100. + g[None]).astype(np.float32)),
        rvel=torch.tensor(idx[:, None] * 1.0 + 0.001 * g), mask=torch.tensor(idx[:, None] * 0. + (g % 3 == 0)),
        height=torch.tensor((idx[:, None] * 2.0 + g).astype(np.float32)))


def _worker_volume(rank, world, port, q, shapes, n_gates):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    ok = True
    try:
        # several scan shapes through the SAME process group, one after the other: the ranks' ray lists follow
        # from the scan shape alone, so no rank can be left waiting for another's list
        for n_sweeps, rps in shapes:
            n_total = n_sweeps * rps
            shards = parallel.all_shards(n_sweeps, rps, world)
            local = _fake_fields(shards[rank], n_gates)
            g = parallel.VolumeGather(shards, n_total).start(local, async_op=True)
            vol = g.finish()
            one = parallel.gather_rays(local['rvel'], shards, n_total)
            if rank == 0:
                want = _fake_fields(np.arange(n_total), n_gates)
                for k in want:
                    ok = ok and vol[k].dtype == want[k].dtype and bool(torch.equal(vol[k], want[k]))
                ok = ok and bool(torch.equal(one, want['rvel']))
            else:
                assert vol is None and one is None
        assert parallel.any_rank(rank == 1) is True and parallel.any_rank(False) is False
        if rank == 0:
            q.put(ok)
    finally:
        dist.destroy_process_group()


def test_gather_two_ranks_gloo():
    assert _spawn(_worker_volume, ([(3, 7), (2, 8), (1, 5), (1, 6), (1, 1)], 5)) is True


def test_gather_single_process():
    idx = np.array([2, 0, 1])
    out = parallel.gather_rays(torch.tensor([[2.], [0.], [1.]]), [idx], 3)
    assert out[:, 0].tolist() == [0., 1., 2.]


class _FakeCtx(object):
    """The library context's run_rays with the kernels replaced by a function of (azimuth, elevation, gate)."""
    n_gates, doppler, n_vel, n_sub = 6, 1, 0, 1
    var_names = ['U', 'T']

    def run_rays(self, az, el, raise_on_domain=True, check_status=True, diag=False):
        az, el = np.asarray(az, dtype=np.float64), np.asarray(el, dtype=np.float64)
        g = np.arange(self.n_gates)[None, :]
        base = az[:, None] * 10. + el[:, None] * 1000. + g
        status = np.zeros(len(az), dtype=np.int32)
        status[(el > 80.)] = 4          # "leaves the model domain"
        out = {'obs': np.stack([(base + 0.5 * i).astype(np.float32) for i in range(9)]), 'rvel': base * 2., 'mask': base * 0.,
               'lat': base + 0.25, 'lon': base + 0.75, 'dist': g.astype(np.float32) + 0 * base.astype(np.float32),
               'height': (base * 3).astype(np.float32), 'ray_status': status}
        if diag:
            out['diag'] = np.stack([base * 7. + 1., base * 11. + 2.])          # [variables, rays, gates]
        return out


def _worker_operator(rank, world, port, q):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from cosmo_pol_amd import RadarOperator, cfg, runtime
        from oracle import cosmo_oracle as O

        class _Eng(object):
            ctx, n_gates = _FakeCtx(), _FakeCtx.n_gates

            def set_refraction(self, *a):
                return 0

        class _St(object):
            eng = _Eng()
        runtime.get_engine = lambda *a, **k: _St()
        runtime.ensure_luts = runtime.ensure_model = runtime.ensure_quadrature = lambda *a, **k: None
        conf = O.make_config(radar={'range': 3000., 'radial_resolution': 500.}, doppler={'scheme': 1})
        op = RadarOperator(conf)
        op.dic_vars = {'T': type('V', (), {'attributes': {}})()}
        op.lut_sz = {}
        ok = True
        # two PPIs of different shape, then an RHI, in the same process group
        for els, azs in (([1., 5.], np.arange(0., 70., 10.)), ([2.], np.arange(0., 50., 10.))):
            radar = op.get_PPI(els, azimuths=azs)
            if rank == 0:
                want = _FakeCtx().run_rays(np.tile(azs, len(els)), np.repeat(els, len(azs)))
                ok = ok and radar.nrays == len(els) * len(azs)
                ok = ok and np.array_equal(radar.fields['KDP']['data'].filled(np.nan), want['obs'][3])
                ok = ok and np.array_equal(radar.fields['RVEL']['data'].filled(np.nan), want['rvel'])
                ok = ok and np.array_equal(radar.gate_altitude['data'], want['height'])
            else:
                assert radar is None
        # diagnostic mode: the sub-beam sums of the model variables travel with the volume (radar_operator.py:245-246)
        opd = RadarOperator(conf, diagnostic_mode=True)
        opd.dic_vars, opd.lut_sz = op.dic_vars, {}
        els, azs = [3.], np.arange(0., 50., 10.)
        radar = opd.get_PPI(els, azimuths=azs)
        if rank == 0:
            want = _FakeCtx().run_rays(azs, np.repeat(els, len(azs)), diag=True)
            ok = ok and np.array_equal(radar.fields['U']['data'].filled(np.nan), want['diag'][0])
            ok = ok and np.array_equal(radar.fields['T']['data'].filled(np.nan), want['diag'][1])
            ok = ok and np.array_equal(radar.fields['ZH']['data'].filled(np.nan), 10 * np.log10(want['obs'][0]))
        else:
            assert radar is None
        # a ray that leaves the domain on ONE rank raises on both (no rank is left waiting in the gather)
        try:
            op.get_RHI(30., elevations=[10., 85.])
            ok = False
        except IndexError:
            pass
        if rank == 0:
            q.put(bool(ok))
        else:
            assert ok
    finally:
        dist.destroy_process_group()


def test_radar_operator_two_ranks_gloo():
    """RadarOperator.get_PPI / get_RHI under a world-size-2 gloo group: rays sharded, the volume gathered on
    rank 0 with one collective, domain errors raised on every rank."""
    assert _spawn(_worker_operator, ()) is True


def _run_bench(extra):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--small', '--stub-engine',
                        '--steps', '3', '--warmup', '1'] + extra, capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`bench.py --gpus 2` started plainly (no RANK in the environment) starts two ranks itself; with the CPU
    stand-in for the kernels the whole multi-rank structure runs here: one volume sharded by azimuth, one gather
    of the outputs to rank 0 (overlapped with the next volume), one JSON line from rank 0 that carries the per-rank
    timings.  The volume that arrives is the same bits whether every field travels every time (--gather full) or the
    scan geometry travels once and the mask as integer codes (the default)."""
    d = _run_bench([])
    assert d['n_gpus'] == 2 and d['scaling'] == 'strong' and d['steps'] == 3
    assert d['gathered_volume_ok'] is True and d['gather_overlapped'] is True
    assert d['config']['rays_per_rank'] == 72 and 'INVALID' in d['config']['workload']
    assert abs(d['value'] - 144 * d['config']['gates'] * 3 / (d['ms_per_step'] * 3e-3)) < 1e-6 * d['value']
    assert d['gather_mode'] == 'compact' and d['gather_bytes_per_gate'] == 45
    assert [r['rank'] for r in d['per_rank']] == [0, 1]
    for r in d['per_rank']:
        assert r['rays'] == 72 and r['compute_ms'] > 0 and r['wait_ms'] >= 0 and r['assemble_ms'] >= 0
    assert abs(d['compute_ms_per_step_max_rank'] - max(r['compute_ms'] for r in d['per_rank'])) < 1e-9
    full = _run_bench(['--gather', 'full'])
    assert full['gather_mode'] == 'full' and full['gather_bytes_per_gate'] == 76 and full['gathered_volume_ok'] is True
    assert full['gathered_volume_sha1'] == d['gathered_volume_sha1'] and len(d['gathered_volume_sha1']) == 40


def test_mask_codes_are_lossless():
    """The output mask (mean of the sub-beam codes, [0, 1) -> 0, scattering_sz.py:144-151) survives the trip as an
    integer sum bit for bit, for every possible sum and sub-beam count."""
    for n_sub in (1, 3, 9, 27, 81, 200):
        sums = torch.arange(-n_sub, n_sub + 1, dtype=torch.float64)
        m = sums / n_sub
        m = torch.where((m >= 0) & (m < 1), torch.zeros_like(m), m)
        codes = parallel.mask_to_codes(m, n_sub)
        assert codes.dtype == (torch.int8 if n_sub <= 127 else torch.int16)
        assert torch.equal(parallel.codes_to_mask(codes, n_sub), m)


def test_static_fields_travel_once():
    """VolumeGather with static fields in a single process: the second start() with the same geometry key packs a
    shorter row, finish() still returns every field; a new key sends them again."""
    idx = [np.array([1, 0, 2])]
    g = parallel.VolumeGather(idx, 3, static=('height',), mask_sub=9)
    f = OrderedDict(rvel=torch.tensor([[1.], [0.], [2.]]), mask=torch.tensor([[0.], [1.], [-5. / 9.]], dtype=torch.float64),
                    height=torch.tensor([[10.], [0.], [20.]], dtype=torch.float32))
    v = g.start(f, geometry_key='a').finish()
    first = g.layout
    assert v['height'][:, 0].tolist() == [0., 10., 20.] and v['rvel'][:, 0].tolist() == [0., 1., 2.]
    assert v['mask'][:, 0].tolist() == [1., 0., -5. / 9.]
    f2 = OrderedDict(f)
    f2['rvel'] = f['rvel'] * 2
    f2['height'] = f['height'] * 0 - 1                      # not sent: same geometry key
    v = g.start(f2, geometry_key='a').finish()
    assert [l[0] for l in g.layout] == ['rvel', 'mask'] and len(first) == 3
    assert v['height'][:, 0].tolist() == [0., 10., 20.] and v['rvel'][:, 0].tolist() == [0., 2., 4.]
    v = g.start(f2, geometry_key='b').finish()
    assert v['height'][:, 0].tolist() == [-1., -1., -1.]
