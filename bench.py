#!/usr/bin/env python
"""Headline benchmark: polarimetric radar gates/s of the forward-operator hot path.

Workload (BASELINE.json configs[3], "CH_PPI_lema"): one C-band PPI volume of 20 elevations x
360 azimuths x 500 gates on a COSMO-1-shaped model grid (80 x 774 x 1158), 2-moment
microphysics (4 hydrometeor classes), 3 x 3 Gauss-Hermite antenna quadrature (9 sub-beams per
gate), attenuation correction on, Doppler scheme 2 (RCS-weighted fall speed; `--doppler 1` selects the
closed-form scheme the reference's CH_PPI_lema.yml names).  Synthetic model state and synthetic
lookup tables (tests/synthetic.py).  One step = one whole volume: geolocation ->
trilinear interpolation -> PSD set-up / bucket sort -> PSD integration -> per-ray scans, all
inputs and outputs resident in HBM.  `--config C1|C2|C3|C5` runs the other reference configurations.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  Started by torch.distributed.run (RANK / WORLD_SIZE in the environment)
the script is one rank; started plainly with --gpus N > 1 it launches the N ranks itself, as children, before
anything touches a GPU.  The rays of ONE volume are sharded block-cyclically by azimuth over the ranks, model and
tables replicated, and every step ends with one gather of all output fields (observables, radial velocity,
mask, geometry) from device memory to rank 0 - the split of the reference's pool.map over rays
(radar_operator.py:253-256).  The gather of one volume runs beside the kernels of the next (two output
buffers).  `--scaling weak` instead lets every rank simulate a whole volume of its own (one radar of a
network per GPU, no collective).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "polarimetric radar gates/sec (Zh,Zdr,Kdp,Ah) per GPU and per 8×MI355X node"
ELEVATIONS = [-0.2, 0.4, 1, 1.6, 2.5, 3.5, 4.5, 5.5, 6.5, 7.5, 8.5, 9.5, 11, 13, 16, 20, 25, 30, 35, 40]
PEAK_F32_MFMA_TFLOPS = 157.3      # dense f32-input MFMA peak of MI355X
N_D = 1024
FLOP_PER_UNIT = 27 * N_D          # 12 components x (mul + add) + 3 for N(D), per diameter bin
FLOP_PER_UNIT_DOP2 = 6 * N_D      # Doppler scheme 2: two more columns (N rcs, N v rcs), mul + add + weight each

# The reference's five option files (option_files/*.yml) as workloads; C4 is the headline.  kind: scan type; rays: (n_sweeps, rays per sweep).
WORKLOADS = {
    'C1': dict(name='test_strat PPI', kind='ppi', mp=1, gh=1, dop=1, freq=5.6, rng=50000, res=75, bw=1.45,
               elevations=[1.0], az_step=1.45, coords=[46.04079, 8.83321, 2050]),
    'C2': dict(name='MXPOL_RHI', kind='rhi', mp=1, gh=3, dop=1, freq=9.41, rng=50000, res=75, bw=1.45,
               azimuths=[47.], el_step=1.45, coords=[46.04079, 8.83321, 2050]),
    'C3': dict(name='MXPOL_RHI_new_scheme', kind='rhi', mp=2, gh=3, dop=1, freq=9.41, rng=50000, res=75, bw=1.45,
               azimuths=[140.], el_step=1.45, coords=[46.04079, 8.83321, 2050]),
    'C4': dict(name='CH_PPI_lema volume', kind='ppi', mp=2, gh=3, dop=2, freq=5.6, rng=250000, res=500, bw=1.0,
               elevations=ELEVATIONS, az_step=1.0, coords=[46.04079, 8.83321, 2050]),
    'C5': dict(name='MXPOL_RHI_PAYERNE_DOPPLER', kind='rhi', mp=1, gh=3, dop=3, freq=9.41, rng=11000, res=75, bw=1.45,
               azimuths=[90.], el_step=1.45, coords=[46.04079, 8.83321, 2050], pri=700, fft=256),
}


def workload_config(small=False, doppler=2, microphysics=2, n_gh=3, config='C4', turbulence=0):
    from cosmo_pol_amd import cfg
    w = WORKLOADS[config]
    # Monte Lema's position; the altitude is 80 m above the synthetic terrain at that grid cell (a radar below
    # the terrain would see NaN air density at its first gate, which the reference turns into NaN RVEL for
    # the whole ray, doppler_sz.py:231)
    conf = {'radar': {'coords': list(w['coords']), 'type': 'ground', 'frequency': w['freq'],
                      'range': w['rng'], 'radial_resolution': w['res'], 'PRI': w.get('pri', 700),
                      'FFT_length': w.get('fft', 128), 'sensitivity': [-15, 10000], '3dB_beamwidth': w['bw']},
            'integration': {'scheme': 1, 'nh_GH': n_gh, 'nv_GH': n_gh, 'weight_threshold': 1.},
            'refraction': {'scheme': 1}, 'attenuation': {'correction': 1},
            'doppler': {'scheme': doppler, 'turbulence_correction': int(bool(turbulence))},
            'microphysics': {'scheme': microphysics}}
    if small:
        conf['radar']['range'] = min(50000, w['rng'])
    cfg.set_config(conf)
    return cfg.CONFIG


def volume_rays(small=False, config='C4'):
    """(azimuths, elevations, n_sweeps, rays_per_sweep) of the scan, ray index = sweep * rays_per_sweep + i."""
    w = WORKLOADS[config]
    if w['kind'] == 'ppi':
        az = np.arange(0., 360., w['az_step'] if not small else 10.0)
        els = w['elevations'] if not small else w['elevations'][:4]
        return np.tile(az, len(els)), np.repeat(np.array(els, dtype=np.float64), len(az)), len(els), len(az)
    el = np.arange(0., 90. + w['el_step'], w['el_step'] if not small else 10.0)       # radar_operator.py:294-298
    azs = w['azimuths']
    return np.repeat(np.array(azs, dtype=np.float64), len(el)), np.tile(el, len(azs)), len(azs), len(el)


# ----------------------------------------------------------------------------------------
# CPU baseline: the oracle (a port of the reference's numpy/C path), one ray per task over a
# fork pool like radar_operator.py:231-256.  Runs before anything touches the GPU.
# ----------------------------------------------------------------------------------------
_CPU = {}


def _cpu_ray(args):
    az, el = args
    from oracle import cosmo_oracle as O
    conf, dic, luts = _CPU['conf'], _CPU['dic'], _CPU['luts']
    with np.errstate(all='ignore'):
        beams = O.get_profiles_GH(dic, az, el, conf)
        O.get_doppler_velocity(beams, luts, conf, **_CPU['dop_kw'])
        out = O.get_radar_observables(beams, luts, conf)
    return float(np.nansum(out.values['ZH']))


def host_cpu_share():
    """(worker count, logical CPUs this process may run on, CPU model).  The workers are the CPUs the process is
    allowed on, capped by the job's CPU share where the environment states one (MAX_JOBS / OMP_NUM_THREADS)."""
    allowed = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    share = allowed
    for k in ('MAX_JOBS', 'OMP_NUM_THREADS'):
        v = os.environ.get(k, '')
        if v.isdigit() and int(v) > 0:
            share = min(share, int(v))
            break
    model = ''
    try:
        with open('/proc/cpuinfo') as fh:
            for line in fh:
                if line.startswith('model name'):
                    model = line.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return share, allowed, model


def cpu_baseline(conf, model, tabs, azs, elv, n_rays, cores, allowed, cpu_model):
    import multiprocessing as mp
    from oracle import cosmo_oracle as O
    os.environ['OMP_NUM_THREADS'] = '1'           # one thread per pool worker
    attrs = {'z-levels': model['zlev'], 'proj_info': model['proj_info'], 'resolution': model['resolution']}
    _CPU['conf'] = conf
    _CPU['dic'] = {n: O.ModelVar(a, attrs) for n, a in model['fields'].items()}
    _CPU['luts'] = {h: O.Lut(*t) for h, t in tabs.items()}
    _CPU['dop_kw'] = {'get_refl': O.get_refl_c} if conf['doppler']['scheme'] in (3, 4) else {}
    n_rays = min(n_rays, len(azs))
    idx = np.linspace(0, len(azs) - 1, n_rays).astype(int)
    tasks = [(float(azs[i]), float(elv[i])) for i in idx]
    ctx = mp.get_context('fork')
    with ctx.Pool(cores) as pool:
        pool.map(_cpu_ray, tasks[:cores])          # touch pages / import in the workers, untimed
        t0 = time.perf_counter()
        pool.map(_cpu_ray, tasks, chunksize=1)
        dt = time.perf_counter() - t0
    n_gates = len(O.radar_ranges(conf))
    return {'value': n_rays * n_gates / dt, 'unit': 'gates/s', 'cores': cores, 'kind': 'port',
            'cpu_model': cpu_model, 'host_logical_cpus': allowed,
            'sample': '%d of the scan\'s %d rays (every %dth), %d gates each, whole path '
                      '(profiles_GH + doppler + observables), fork pool of %d workers, %.1f s wall'
                      % (n_rays, len(azs), max(1, len(azs) // n_rays), n_gates, cores, dt)}


# ----------------------------------------------------------------------------------------
# the engine under test, and a CPU stand-in for it that lets the multi-rank structure (launch, sharding, gather,
# JSON line) be rehearsed without GPUs
# ----------------------------------------------------------------------------------------
class GpuRunner(object):
    def __init__(self, conf, model, tabs, local_rank, batch_units):
        import torch
        from cosmo_pol_amd import engine
        from cosmo_pol_amd.lut import Lookup_table
        torch.cuda.set_device(local_rank)
        self.torch = torch
        self.dev = torch.device('cuda', local_rank)
        self.eng = engine.Engine(conf, device=local_rank)
        self.eng.upload_luts({h: Lookup_table.from_arrays(*t) for h, t in tabs.items()})
        self.eng.upload_model(model['fields'], model['zlev'], model['proj_info'], model['resolution'])
        self.eng.set_gauss_hermite()
        self.ctx = self.eng.ctx
        if batch_units:
            self.ctx.set_batch_units(batch_units)
        self.n_gates, self.n_sub, self.n_vel = self.eng.n_gates, self.ctx.n_sub, self.ctx.n_vel
        self.doppler = self.ctx.doppler
        # the library's own stream as a torch stream: torch work (packing, collectives) can be queued behind the kernels
        self.lib_stream = torch.cuda.ExternalStream(self.ctx.stream, device=self.dev)
        self.spans = []

    def timed_run(self, az_t, el_t, out_struct):
        """run() between two events on the library's stream (no host synchronisation)."""
        e0, e1 = self.torch.cuda.Event(enable_timing=True), self.torch.cuda.Event(enable_timing=True)
        e0.record(self.lib_stream)
        self.run(az_t, el_t, out_struct)
        e1.record(self.lib_stream)
        self.spans.append((e0, e1))

    def compute_ms(self):
        """Sum of the timed_run spans so far (call after a synchronisation); clears them."""
        t = sum(a.elapsed_time(b) for a, b in self.spans)
        self.spans = []
        return t

    def outputs(self, R, pinned_host=False):
        torch, G = self.torch, self.n_gates
        kw = dict(device='cpu', pin_memory=True) if pinned_host else dict(device=self.dev)
        from cosmo_pol_amd import _lib
        t = {'obs': torch.empty((_lib.N_OBS, R, G), dtype=torch.float32, **kw)}
        for k in ('rvel', 'mask', 'lat', 'lon'):
            t[k] = torch.empty((R, G), dtype=torch.float64, **kw)
        for k in ('dist', 'height'):
            t[k] = torch.empty((R, G), dtype=torch.float32, **kw)
        t['ray_status'] = torch.zeros((R,), dtype=torch.int32, **kw)
        if self.doppler == 3:
            t['dspectrum'] = torch.empty((R, G, self.n_vel), dtype=torch.float64, **kw)
        return t

    def struct(self, t):
        from cosmo_pol_amd import _lib
        P = lambda x, typ: ctypes.cast(x.data_ptr(), typ)
        return _lib.CpolOutputs(P(t['obs'], _lib.c_float_p), P(t['rvel'], _lib.c_double_p), P(t['mask'], _lib.c_double_p),
                                P(t['lat'], _lib.c_double_p), P(t['lon'], _lib.c_double_p), P(t['dist'], _lib.c_float_p),
                                P(t['height'], _lib.c_float_p), P(t['ray_status'], _lib.c_int_p),
                                P(t['dspectrum'], _lib.c_double_p) if 'dspectrum' in t else None)

    def angles(self, az, el):
        return (self.torch.tensor(az, dtype=self.torch.float64, device=self.dev),
                self.torch.tensor(el, dtype=self.torch.float64, device=self.dev))

    def run(self, az_t, el_t, out_struct):
        """Queue one pass over the rays on the library's stream (device outputs)."""
        self.ctx.run_rays_dev(az_t.data_ptr(), el_t.data_ptr(), az_t.numel(), out_struct)

    def run_host(self, az_np, el_np, out_struct):
        """One pass with the outputs delivered to host memory (returns when they are there)."""
        rc = self.ctx.lib.cpol_run_rays(self.ctx.h, len(az_np), az_np.ctypes.data, el_np.ctypes.data, ctypes.byref(out_struct))
        self.ctx._check(rc, 'cpol_run_rays')

    def sync(self):
        self.ctx.synchronize()
        self.torch.cuda.synchronize()


class StubRunner(object):
    """CPU stand-in: every output is a fixed function of (azimuth, elevation, gate).  Results are NOT measurements."""
    n_sub, n_vel, doppler = 9, 0, 2

    def __init__(self, conf):
        import torch
        self.torch = torch
        self.dev = torch.device('cpu')
        r = conf['radar']
        self.n_gates = len(np.arange(r['radial_resolution'] / 2, r['range'], r['radial_resolution']))

    def outputs(self, R, pinned_host=False):
        torch, G = self.torch, self.n_gates
        t = {'obs': torch.empty((9, R, G), dtype=torch.float32)}
        for k in ('rvel', 'mask', 'lat', 'lon'):
            t[k] = torch.empty((R, G), dtype=torch.float64)
        for k in ('dist', 'height'):
            t[k] = torch.empty((R, G), dtype=torch.float32)
        t['ray_status'] = torch.zeros((R,), dtype=torch.int32)
        return t

    def struct(self, t):
        return t

    def angles(self, az, el):
        return self.torch.tensor(az, dtype=self.torch.float64), self.torch.tensor(el, dtype=self.torch.float64)

    lib_stream = None

    def timed_run(self, az_t, el_t, t):
        t0 = time.perf_counter()
        self.run(az_t, el_t, t)
        self._ms = getattr(self, '_ms', 0.) + (time.perf_counter() - t0) * 1e3

    def compute_ms(self):
        t, self._ms = getattr(self, '_ms', 0.), 0.
        return t

    def run(self, az_t, el_t, t):
        g = self.torch.arange(self.n_gates, dtype=self.torch.float64)[None, :]
        base = az_t[:, None] * 10. + el_t[:, None] * 1000. + g
        for i in range(9):
            t['obs'][i] = (base + 0.5 * i).float()
        t['rvel'].copy_(base * 2.)
        # mean of nine sub-beam codes: sums of -5, 9 and 0 in turn along the ray
        code = self.torch.tensor([0., -5., 9., 0.], dtype=self.torch.float64)[(self.torch.arange(self.n_gates) % 4)]
        m = code[None, :] / 9. + 0 * base
        t['mask'].copy_(self.torch.where((m >= 0) & (m < 1), self.torch.zeros_like(m), m))
        t['lat'].copy_(base + 0.25)
        t['lon'].copy_(base + 0.75)
        t['dist'].copy_((g + 0 * base).float())
        t['height'].copy_((base * 3).float())

    def sync(self):
        pass


def launch_ranks(n, argv):
    """Start n ranks of this script as children through torch.distributed.run and return their exit status.  The
    parent never touches a GPU, so nothing that has initialised one is ever replaced or forked."""
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20, help='timed volumes (a volume takes about 9 ms; the first one or two after a pause run slower)')
    ap.add_argument('--warmup', type=int, default=10, help='untimed volumes first (the clocks take a few volumes to settle after the set-up)')
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='strong',
                    help='strong (default): the rays of ONE volume are shared out over the ranks and gathered on rank 0; '
                         'weak: every rank simulates a whole volume of its own')
    ap.add_argument('--config', choices=sorted(WORKLOADS), default='C4', help='reference configuration (C4 = the headline volume)')
    ap.add_argument('--cpu-rays', type=int, default=0, help='rays of the CPU baseline sample (0 = sized to 10-30 s of CPU work)')
    ap.add_argument('--cpu-cores', type=int, default=0, help='workers of the CPU baseline (0 = the CPUs this job may use)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-host-copy', action='store_true', help='skip the second measurement with the outputs delivered to host memory')
    ap.add_argument('--no-overlap', action='store_true', help='N > 1: wait for each gather before the next volume starts')
    ap.add_argument('--rehearse-gather', action='store_true', help='N = 1: run the gather path in a one-rank RCCL group (rehearsal; result is not a scaling number)')
    ap.add_argument('--gather', choices=['compact', 'full'], default='compact',
                    help='N > 1: compact (default) sends the scan geometry once and the mask as integer codes, full sends all seven fields every volume')
    ap.add_argument('--doppler', type=int, choices=[0, 1, 2, 3], default=None,
                    help='doppler.scheme of the workload (3 = full Doppler spectrum, needs --microphysics 1)')
    ap.add_argument('--microphysics', type=int, choices=[1, 2], default=None,
                    help='2 = the headline 2-moment workload; 1 = the same volume with 1-moment input (3 classes)')
    ap.add_argument('--gh', type=int, default=None, help='Gauss-Hermite nodes per direction: 3 = the 9 sub-beams of the headline '
                    'workload, 9 = the 81 sub-beams of the literal CH_PPI_lema.yml (secondary run)')
    ap.add_argument('--turbulence', type=int, choices=[0, 1], default=0, help='C5: turbulence broadening of the spectrum')
    ap.add_argument('--host-batches', type=str, default=None, help='N,F: batch schedule of the host-delivery measurement (cpol_set_host_batching)')
    ap.add_argument('--batch-units', type=int, default=0, help='positions per internal batch (0 = library default)')
    ap.add_argument('--shard-of', type=int, default=0, help='N = 1 only: simulate the share of the rays that one rank of an M-rank run '
                    'takes (no process group, no gather): what a shard costs on its own; the line is not a volume measurement')
    ap.add_argument('--shard-rank', type=int, default=0, help='which rank\'s share --shard-of takes')
    ap.add_argument('--small', action='store_true', help='reduced grid / volume (debugging only; result INVALID)')
    ap.add_argument('--stub-engine', action='store_true',
                    help='CPU stand-in for the kernels over gloo: rehearses launch, sharding and gather without GPUs (result INVALID)')
    args = ap.parse_args(argv)

    if args.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:] if argv is None else list(argv)))

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    wl = WORKLOADS[args.config]
    doppler = wl['dop'] if args.doppler is None else args.doppler
    microphysics = wl['mp'] if args.microphysics is None else args.microphysics
    n_gh = wl['gh'] if args.gh is None else args.gh
    strong = args.scaling == 'strong'

    from tests import synthetic as syn
    conf = workload_config(args.small, doppler, microphysics, n_gh, args.config, args.turbulence)
    azs, elv, n_sweeps, rays_per_sweep = volume_rays(args.small, args.config)
    shape = (80, 774, 1158) if not args.small else (24, 200, 240)
    model = tabs = None
    if not args.stub_engine:
        model = syn.synthetic_model(*shape, scheme=microphysics, seed=20161111 + (0 if strong else rank),
                                    res=(0.01, 0.01) if not args.small else (0.05, 0.05))
        if args.turbulence:
            z = np.asarray(model['zlev'], dtype=np.float64)
            model['fields']['EDR'] = (2e-3 * np.exp(-z / 4000.) * (1.5 + np.sin(z / 700.))).astype('float32')
        tabs = {h: syn.synthetic_lut(h, wl['freq']) for h in syn.hydrom_types(microphysics)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.stub_engine:
        cores, allowed, cpu_model = host_cpu_share()
        if args.cpu_cores:
            cores = args.cpu_cores
        # about 20 s of CPU work: the C4 path costs ~0.19 core-seconds per ray, the Doppler spectrum ~20 x that
        per_ray = 0.19 * (n_gh * n_gh / 9.) * (20. if doppler == 3 else 1.)
        n_cpu = args.cpu_rays or int(max(cores, min(len(azs), 20. * cores / per_ray)))
        cpu = cpu_baseline(conf, model, tabs, azs, elv, n_cpu, cores, allowed, cpu_model)

    import torch
    import torch.distributed as dist
    from cosmo_pol_amd import parallel
    if args.stub_engine:
        runner = StubRunner(conf)
        if world > 1:
            dist.init_process_group('gloo')
    else:
        torch.cuda.set_device(local_rank)
        if args.rehearse_gather and world == 1:
            # one-GPU rehearsal of the multi-rank path: a group of one rank over RCCL, the gather issued all the same
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', '29533')
            dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', local_rank))
            parallel.SINGLE_RANK_COLLECTIVES = True
        if world > 1:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        runner = GpuRunner(conf, model, tabs, local_rank, args.batch_units)
    del model
    G = runner.n_gates

    shards = parallel.all_shards(n_sweeps, rays_per_sweep, world) if strong else [np.arange(len(azs))] * world
    mine = shards[rank]
    if args.shard_of and world == 1:
        mine = parallel.all_shards(n_sweeps, rays_per_sweep, args.shard_of)[args.shard_rank]
    R = len(mine)
    az_t, el_t = runner.angles(azs[mine], elv[mine])
    gather = strong and (world > 1 or args.rehearse_gather)
    n_buf = 2 if gather and not args.no_overlap else 1
    bufs = [runner.outputs(R) for _ in range(n_buf)]
    structs = [runner.struct(b) for b in bufs]
    # what travels: everything with --gather full; by default the scan geometry (lat, lon, range, height) once per
    # buffer set and the mask as the integer sum of its sub-beam codes - 45 instead of 76 bytes per gate
    compact = args.gather == 'compact'
    gathers = [parallel.VolumeGather(shards, len(azs), static=('lat', 'lon', 'dist', 'height') if compact else (),
                                     mask_sub=runner.n_sub if compact else 0) for _ in range(n_buf)] if gather else None
    t_wait = 0.
    volume = None
    assemble_ev = []

    def fields_of(t):
        from collections import OrderedDict
        f = OrderedDict(obs=t['obs'].permute(1, 0, 2))
        for k in ('rvel', 'mask', 'lat', 'lon', 'dist', 'height'):
            f[k] = t[k]
        return f

    def take(i):
        """Rank 0 waits until the volume of buffer set i is in place (the consumer's wait); the wait is timed."""
        nonlocal t_wait
        if gather:
            t0 = time.perf_counter()
            gathers[i].wait()
            t_wait += time.perf_counter() - t0

    def step(k):
        nonlocal volume
        i = k % n_buf
        take(i)                                    # the buffer set's previous volume has been consumed
        runner.timed_run(az_t, el_t, structs[i])
        if gather:
            # queued behind the kernels on the library's stream: packing, the collective, and on rank 0 the indexed
            # copy into the volume on a side stream; the host waits for none of it here
            gathers[i].start(fields_of(bufs[i]), async_op=not args.no_overlap, geometry_key='scan', stream=runner.lib_stream)
            v = gathers[i].finish()
            if v is not None:
                volume = v
                if gathers[i].assemble_ms_events:
                    assemble_ev.append(gathers[i].assemble_ms_events)
            if args.no_overlap:
                take(i)

    def sync_all():
        for i in range(n_buf):
            take(i)
        runner.sync()
        if dist.is_initialized():
            dist.barrier()

    for k in range(args.warmup):
        step(k)
    sync_all()
    assert int(bufs[0]['ray_status'].abs().sum().item()) == 0, "a ray left the model domain"
    timed = not args.stub_engine
    runner.compute_ms()
    t_wait = 0.
    assemble_ev = []
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    sync_all()
    dt = time.perf_counter() - t0
    t_compute = runner.compute_ms() * 1e-3
    t_gather = t_wait
    t_assemble = sum(a.elapsed_time(b) for a, b in assemble_ev) * 1e-3
    # The same steps once more with the library's stage timing on (HIP events around every stage on the library's
    # stream, work counters): the per-stage and per-kernel figures come from this pass, `value` from the plain one
    # above (the events and the small device-to-host copies of the counters cost about 4 % of a step).
    tim = None
    if timed:
        runner.ctx.timing_enable(True)
        sync_all()
        t1 = time.perf_counter()
        for k in range(args.steps):
            step(k)
        sync_all()
        dt_instr = time.perf_counter() - t1
        tim = runner.ctx.timing_read()
        runner.ctx.timing_enable(False)
        runner.compute_ms()
    per_rank = None
    if world > 1 or gather:
        t = torch.tensor([dt, t_compute, t_gather], dtype=torch.float64, device=runner.dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        mine_ms = {'rank': rank, 'rays': R, 'compute_ms': t_compute / args.steps * 1e3, 'wait_ms': t_gather / args.steps * 1e3,
                   'assemble_ms': t_assemble / args.steps * 1e3}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine_ms)
        dt, t_compute, t_gather = (float(x) for x in t.tolist())
    last = bufs[(args.steps - 1) % n_buf] if args.steps else bufs[0]
    finite = float(torch.isfinite(last['obs'][0]).float().mean().item())
    finite_rvel = float(torch.isfinite(last['rvel']).float().mean().item()) if doppler else None
    gathered_ok = volume_sha1 = None
    if gather and rank == 0 and volume is not None:
        import hashlib
        # rank 0's own rays sit at their positions of the gathered volume, for every field
        sel = torch.as_tensor(mine, device=volume['rvel'].device)
        gathered_ok = tuple(volume['obs'].shape) == (len(azs), 9, G)
        for k, v in fields_of(last).items():
            gathered_ok = gathered_ok and bool(torch.equal(volume[k][sel].nan_to_num(), v.nan_to_num()))
        h = hashlib.sha1()
        for k in ('obs', 'rvel', 'mask', 'lat', 'lon', 'dist', 'height'):
            h.update(volume[k].contiguous().cpu().numpy().tobytes())
        volume_sha1 = h.hexdigest()

    # second measurement at N = 1: the same steps with the outputs delivered to (pinned) host memory, the copy of
    # one batch of rays running beside the kernels of the next (batches shrinking geometrically, so that the last
    # batch's copy, the only one nothing hides, is short)
    host = None
    if timed and world == 1 and not args.no_host_copy:
        hb = runner.outputs(R, pinned_host=True)
        hs = runner.struct(hb)
        az_np, el_np = np.ascontiguousarray(azs[mine]), np.ascontiguousarray(elv[mine])
        if args.host_batches:
            nb_, lf_ = args.host_batches.split(',')
            runner.ctx.set_host_batching(int(nb_), float(lf_))
        # (a) one blocking call per volume (cpol_run_rays): returns when the volume is in host memory
        runner.run_host(az_np, el_np, hs)
        runner.sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            runner.run_host(az_np, el_np, hs)
        dtb = time.perf_counter() - t0
        names = ('obs', 'rvel', 'mask', 'lat', 'lon', 'dist', 'height')
        same = all(torch.equal(hb[k].nan_to_num(), last[k].cpu().nan_to_num()) for k in names)
        # (b) the same K volumes queued one after the other (cpol_run_rays_async, two host buffer sets): volume k - 1
        # is awaited, and could be consumed, while volume k runs; the K-th delivery ends inside the timed region
        hb2 = runner.outputs(R, pinned_host=True)
        hsets, hstructs = [hb, hb2], [hs, runner.struct(hb2)]
        runner.ctx.set_host_batching(1, 0.)
        for k in range(2):
            runner.ctx.run_rays_async(az_t.data_ptr(), el_t.data_ptr(), R, hstructs[k])
        runner.sync()
        for b in hsets:
            b['obs'].fill_(-1.)
        t0 = time.perf_counter()
        for k in range(args.steps):
            runner.ctx.run_rays_async(az_t.data_ptr(), el_t.data_ptr(), R, hstructs[k % 2])
            if k:
                runner.ctx.wait_async(1)
        runner.sync()
        dth = time.perf_counter() - t0
        same = same and all(torch.equal(b[k].nan_to_num(), last[k].cpu().nan_to_num()) for k in names for b in hsets[:min(2, args.steps)])
        runner.ctx.set_host_batching(4, 0.25)
        host = {'ms_per_step_host': dth / args.steps * 1e3, 'value_host': len(azs) * G * args.steps / dth,
                'host_delivery': 'K volumes queued back to back (cpol_run_rays_async), the outputs of one travelling to '
                                 'page-locked host memory beside the kernels of the next; all K deliveries inside the timed region',
                'ms_per_call_host_blocking': dtb / args.steps * 1e3,
                'host_outputs_equal_device_outputs': bool(same),
                'host_bytes_per_step': int(sum(v.numel() * v.element_size() for v in hb.values()))}

    if rank == 0:
        gates_per_step = (len(mine) if (args.shard_of and world == 1) else len(azs)) * G * (1 if strong else world)
        roofline = stage = None
        extra = {}
        if tim is not None:
            ms_int = tim['ms']['integrate'] / max(1, tim['launches']['integrate'])
            # the two Doppler scheme-2 columns count for k_integrate only when they ride on its units; with
            # 2-moment input they are units of their own on the vector ALUs (stage integrate_dop)
            flop_unit = FLOP_PER_UNIT + (FLOP_PER_UNIT_DOP2 if doppler == 2 and tim['doppler_units'] == 0 else 0)
            # Of a unit's 1024 terms only those whose float32 N(D) is not an exact zero are evaluated (the library
            # counts them); `achieved` prices the evaluated terms against the matrix-core peak, `algorithmic` is the
            # rate of the reference's full 1024-term contraction that the same launch stands for.
            evaluated = tim['evaluated_unit_diameters'] / (1024. * tim['valid_units']) if tim['valid_units'] else 1.
            flop_alg_launch = flop_unit * tim['valid_units'] / max(1, tim['launches']['integrate'])
            flop_per_launch = flop_alg_launch * evaluated
            achieved = flop_per_launch / (ms_int * 1e-3) / 1e12 if ms_int > 0 else 0.
            algorithmic = flop_alg_launch / (ms_int * 1e-3) / 1e12 if ms_int > 0 else 0.
            # HBM bytes per launch and instruction counts are not measured by this run: they come from the hardware-
            # counter passes of profiles/collect_pmc.py on the same workload (profiles/traffic.json, which names its
            # source); the launch times beside them are this run's
            traffic = tsrc = None
            prof = {}
            tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
            if os.path.exists(tpath) and not args.small and args.config == 'C4' and world == 1 and doppler == 2 and microphysics == 2 and n_gh == 3:
                prof = json.load(open(tpath))
                traffic = prof.get('k_integrate_hbm_bytes_per_launch')
                tsrc = prof.get('source')
            # what the kernel executes per evaluated term: a multiply-add per column of the image the unit is integrated
            # over (8 combined columns, or all 12 / 14) plus 3 for N(D) - fewer than the reference's count when the
            # units take the combined columns, so `achieved` (the reference's count) can exceed `executed_tflops`
            comb_frac = tim['combined_units'] / tim['valid_units'] if tim['valid_units'] else 0.
            ncol_all = 14 if (doppler == 2 and tim['doppler_units'] == 0) else 12
            executed = achieved * (2. * (8. * comb_frac + ncol_all * (1. - comb_frac)) + 3.) / (flop_unit / float(N_D)) if flop_unit else 0.
            roofline = {'bound': 'mfma', 'kernel': 'k_integrate', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS,
                        'executed_tflops': executed, 'executed_frac': executed / PEAK_F32_MFMA_TFLOPS,
                        'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic, 'traffic_source': tsrc,
                        'avg_launch_ms': ms_int, 'flop_per_launch': flop_per_launch, 'flop_per_unit': flop_unit,
                        'launches': tim['launches']['integrate'], 'evaluated_fraction': evaluated,
                        'own_quads_fraction': (tim['own_unit_diameters'] / (1024. * tim['valid_units']) if tim['valid_units'] else None),
                        # share of the units integrated over the seven combinations of columns the observables are made
                        # of (8-column image) instead of all twelve components (cpol_set_column_combination)
                        'combined_units_fraction': (tim['combined_units'] / tim['valid_units'] if tim['valid_units'] else None),
                        'algorithmic_tflops': algorithmic}
            # the two other kernels of weight, against the limits that bind them (counts from the same counter passes)
            others = []
            ms_front = tim['ms']['interp'] / max(1, tim['launches']['interp'])
            ms_obs = tim['ms']['observables'] / max(1, tim['launches']['observables'])
            if prof.get('k_front_valu_insts_per_launch') and ms_front > 0:
                # a SIMD issues one vector instruction of a wavefront per 4 cycles: 1024 SIMDs x 2.4 GHz / 4
                peak_v = 1024 * 2.4e9 / 4 / 1e9
                ach_v = prof['k_front_valu_insts_per_launch'] / (ms_front * 1e-3) / 1e9
                others.append({'kernel': 'k_front', 'bound': 'valu_issue', 'achieved': ach_v, 'peak': peak_v,
                               'unit': 'G wavefront-instructions/s', 'frac': ach_v / peak_v, 'avg_launch_ms': ms_front,
                               'vector_memory_busy_frac': prof.get('k_front_ta_busy_frac'),
                               'note': 'vector ALU issue and the vector memory path (TA) are both near their limit'})
            if prof.get('k_gate_reduce_hbm_bytes_per_launch') and ms_obs > 0:
                ach_b = prof['k_gate_reduce_hbm_bytes_per_launch'] / (ms_obs * 1e-3) / 1e9
                others.append({'kernel': 'k_gate_reduce (+ k_ray_scan)', 'bound': 'hbm', 'achieved': ach_b, 'peak': 8000.,
                               'unit': 'GB/s', 'frac': ach_b / 8000., 'avg_launch_ms': ms_obs,
                               'traffic': prof['k_gate_reduce_hbm_bytes_per_launch']})
            roofline['other_kernels'] = others
            stage = {k: v / args.steps for k, v in tim['ms'].items()}
            stage['ms_per_step_with_stage_timing'] = dt_instr / args.steps * 1e3
            extra = {'valid_units_per_step': tim['valid_units'] / args.steps,
                     'doppler_units_per_step': tim['doppler_units'] / args.steps,
                     'doppler_evaluated_fraction': (tim['doppler_unit_diameters'] / (1024. * tim['doppler_units'])
                                                    if tim['doppler_units'] else None)}
        invalid = ' [SMALL DEBUG RUN - INVALID]' if args.small else ''
        invalid += ' [STUB ENGINE, NO KERNELS - INVALID]' if args.stub_engine else ''
        invalid += ' [SHARE OF RANK %d OF %d ONLY - NOT A VOLUME]' % (args.shard_rank, args.shard_of) if (args.shard_of and world == 1) else ''
        scan = '%d elevations x %d azimuths' % (n_sweeps, rays_per_sweep) if wl['kind'] == 'ppi' else \
            'RHI, %d azimuth(s) x %d elevations' % (n_sweeps, rays_per_sweep)
        line = {
            'metric': METRIC, 'value': gates_per_step * args.steps / dt, 'unit': 'gates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': '%s (%s): %s x %d gates, COSMO-1 grid %dx%dx%d, %d-moment (%d classes), %dx%d '
                                   'Gauss-Hermite (%d sub-beams), attenuation on, Doppler scheme %d%s%s'
                                   % (wl['name'], args.config, scan, G, shape[2], shape[1], shape[0], microphysics,
                                      microphysics + 2, n_gh, n_gh, runner.n_sub, doppler,
                                      ' + turbulence' if args.turbulence else '', invalid),
                       'rays_per_rank': R, 'gates': G, 'sub_beams': runner.n_sub, 'classes': microphysics + 2,
                       'parallelism': ('one volume, rays block-cyclic by azimuth over %d rank(s), one gather of all outputs to rank 0'
                                       if strong else 'one volume per rank, %d rank(s), no collective') % world},
            'roofline': roofline,
            'stage_ms_per_step': stage,
            'finite_zh_fraction': finite, 'finite_rvel_fraction': finite_rvel,
            'cpu_baseline': cpu,
        }
        line.update(extra)
        if host:
            line.update(host)
        if gather:
            # compute_ms: a rank's kernels for its share of the rays (HIP events, no communication in it - what the
            # rank would need for that share on its own); wait_ms: host time a rank spends waiting for the volume
            # of a buffer set to be in place before it reuses the set (the GPU works meanwhile: with two sets this
            # is most of a step, it is not exposed time); assemble_ms: the indexed copy of the received rows into
            # the volume on rank 0
            line.update({'compute_ms_per_step_max_rank': t_compute / args.steps * 1e3,
                         'gather_wait_ms_per_step_max_rank': t_gather / args.steps * 1e3,
                         'gather_overlapped': not args.no_overlap, 'gathered_volume_ok': gathered_ok,
                         'gather_mode': args.gather, 'gather_bytes_per_gate': 45 if args.gather == 'compact' else 76,
                         'gathered_volume_sha1': volume_sha1, 'per_rank': per_rank})
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
