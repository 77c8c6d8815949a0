#!/usr/bin/env python
"""Headline benchmark: polarimetric radar gates/s of the forward-operator hot path.

Workload (BASELINE.json configs[3], "CH_PPI_lema"): one C-band PPI volume of 20 elevations x
360 azimuths x 500 gates on a COSMO-1-shaped model grid (80 x 774 x 1158), 2-moment
microphysics (4 hydrometeor classes), 3 x 3 Gauss-Hermite antenna quadrature (9 sub-beams per
gate), attenuation correction on, Doppler scheme 2 (RCS-weighted fall speed; `--doppler 1` selects the
closed-form scheme the reference's CH_PPI_lema.yml names).  Synthetic model state and synthetic
lookup tables (cosmo_pol_amd/synthetic.py).  One step = one whole volume: geolocation ->
trilinear interpolation -> PSD set-up / bucket sort -> PSD integration -> per-ray scans, all
inputs and outputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by torch.distributed.run with one rank per GPU.  Scaling is weak: every rank
simulates one volume (one radar of a network / one model time step) on its own replica of the
model state and tables, so there is no data-path collective; value = all ranks' gates / time.
`--scaling strong` instead shards the rays of ONE volume block-cyclically by azimuth over the
ranks and gathers the output volume on rank 0 with one RCCL gather.
"""
import argparse
import ctypes
import json
import os
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


def workload_config(small=False, doppler=2, microphysics=2, n_gh=3):
    from cosmo_pol_amd import cfg
    # Monte Lema's position; the altitude is 80 m above the synthetic terrain at that grid cell (a radar below
    # the terrain would see NaN air density at its first gate, which the reference turns into NaN RVEL for
    # the whole ray, doppler_sz.py:231)
    conf = {'radar': {'coords': [46.04079, 8.83321, 2050], 'type': 'ground', 'frequency': 5.6,
                      'range': 250000, 'radial_resolution': 500, 'PRI': 700, 'FFT_length': 128,
                      'sensitivity': [-15, 10000], '3dB_beamwidth': 1.},
            'integration': {'scheme': 1, 'nh_GH': n_gh, 'nv_GH': n_gh, 'weight_threshold': 1.},
            'refraction': {'scheme': 1}, 'attenuation': {'correction': 1}, 'doppler': {'scheme': doppler},
            'microphysics': {'scheme': microphysics}}
    if small:
        conf['radar']['range'] = 50000
    cfg.set_config(conf)
    return cfg.CONFIG


def volume_rays(small=False):
    az = np.arange(0., 360., 1.0 if not small else 10.0)
    els = ELEVATIONS if not small else ELEVATIONS[:4]
    azs = np.tile(az, len(els))
    elv = np.repeat(np.array(els, dtype=np.float64), len(az))
    return azs, elv


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
        O.get_doppler_velocity(beams, luts, conf)
        out = O.get_radar_observables(beams, luts, conf)
    return float(np.nansum(out.values['ZH']))


def cpu_baseline(conf, model, tabs, azs, elv, n_rays, cores):
    import multiprocessing as mp
    from oracle import cosmo_oracle as O
    os.environ.setdefault('OMP_NUM_THREADS', '1')
    attrs = {'z-levels': model['zlev'], 'proj_info': model['proj_info'], 'resolution': model['resolution']}
    _CPU['conf'] = conf
    _CPU['dic'] = {n: O.ModelVar(a, attrs) for n, a in model['fields'].items()}
    _CPU['luts'] = {h: O.Lut(*t) for h, t in tabs.items()}
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
            'sample': '%d of the volume\'s %d rays (every %dth), %d gates each, whole path '
                      '(profiles_GH + doppler + observables), fork pool of %d workers, %.1f s wall'
                      % (n_rays, len(azs), max(1, len(azs) // n_rays), n_gates, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--scaling', choices=['weak', 'strong'], default='weak')
    ap.add_argument('--cpu-rays', type=int, default=1440)
    ap.add_argument('--cpu-cores', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--doppler', type=int, choices=[0, 1, 2, 3], default=2,
                    help='doppler.scheme of the workload (3 = full Doppler spectrum, needs --microphysics 1)')
    ap.add_argument('--microphysics', type=int, choices=[1, 2], default=2,
                    help='2 = the headline 2-moment workload; 1 = the same volume with 1-moment input (3 classes)')
    ap.add_argument('--gh', type=int, default=3, help='Gauss-Hermite nodes per direction: 3 = the 9 sub-beams of the headline '
                    'workload, 9 = the 81 sub-beams of the literal CH_PPI_lema.yml (secondary run)')
    ap.add_argument('--batch-units', type=int, default=0, help='positions per internal batch (0 = library default)')
    ap.add_argument('--small', action='store_true', help='reduced grid / volume (debugging only; result INVALID)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))

    from cosmo_pol_amd import synthetic as syn
    conf = workload_config(args.small, args.doppler, args.microphysics, args.gh)
    azs, elv = volume_rays(args.small)
    shape = (80, 774, 1158) if not args.small else (24, 200, 240)
    model = syn.synthetic_model(*shape, scheme=args.microphysics, seed=20161111 + (rank if args.scaling == 'weak' else 0),
                                res=(0.01, 0.01) if not args.small else (0.05, 0.05))
    tabs = {h: syn.synthetic_lut(h, 5.6) for h in syn.hydrom_types(args.microphysics)}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(conf, model, tabs, azs, elv, args.cpu_rays, args.cpu_cores)

    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    from cosmo_pol_amd import engine, _lib
    from cosmo_pol_amd.lut import Lookup_table
    eng = engine.Engine(conf, device=local_rank)
    eng.upload_luts({h: Lookup_table.from_arrays(*t) for h, t in tabs.items()})
    eng.upload_model(model['fields'], model['zlev'], model['proj_info'], model['resolution'])
    eng.set_gauss_hermite()
    del model
    ctx = eng.ctx
    if args.batch_units:
        ctx.set_batch_units(args.batch_units)

    if args.scaling == 'strong' and world > 1:
        n_az = len(np.unique(azs))
        sel = np.array([i for i in range(len(azs)) if (i % n_az) % world == rank])
        my_az, my_el = azs[sel], elv[sel]
    else:
        my_az, my_el = azs, elv
    R, G = len(my_az), eng.n_gates
    dev = torch.device('cuda', local_rank)
    az_t = torch.tensor(my_az, dtype=torch.float64, device=dev)
    el_t = torch.tensor(my_el, dtype=torch.float64, device=dev)
    obs_t = torch.empty((_lib.N_OBS, R, G), dtype=torch.float32, device=dev)
    rvel_t = torch.empty((R, G), dtype=torch.float64, device=dev)
    mask_t = torch.empty((R, G), dtype=torch.float64, device=dev)
    lat_t = torch.empty((R, G), dtype=torch.float64, device=dev)
    lon_t = torch.empty((R, G), dtype=torch.float64, device=dev)
    dist_t = torch.empty((R, G), dtype=torch.float32, device=dev)
    hgt_t = torch.empty((R, G), dtype=torch.float32, device=dev)
    stat_t = torch.zeros((R,), dtype=torch.int32, device=dev)
    gather_list = None
    if args.scaling == 'strong' and world > 1 and rank == 0:
        gather_list = [torch.empty_like(obs_t) for _ in range(world)]
    torch.cuda.synchronize()

    def P(t, typ):
        return ctypes.cast(t.data_ptr(), typ)

    def step():
        o = _lib.CpolOutputs(P(obs_t, _lib.c_float_p), P(rvel_t, _lib.c_double_p), P(mask_t, _lib.c_double_p),
                             P(lat_t, _lib.c_double_p), P(lon_t, _lib.c_double_p), P(dist_t, _lib.c_float_p),
                             P(hgt_t, _lib.c_float_p), P(stat_t, _lib.c_int_p))
        ctx.run_rays_dev(az_t.data_ptr(), el_t.data_ptr(), R, o)
        if args.scaling == 'strong' and world > 1:
            ctx.synchronize()
            dist.gather(obs_t, gather_list, dst=0)

    def sync_all():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    sync_all()
    assert int(stat_t.abs().sum().item()) == 0, "a ray left the model domain"
    ctx.timing_enable(True)
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    tim = ctx.timing_read()
    ctx.timing_enable(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    finite = float(torch.isfinite(obs_t[0]).float().mean().item())
    finite_rvel = float(torch.isfinite(rvel_t).float().mean().item()) if args.doppler else None

    if rank == 0:
        gates_per_step = (len(azs) * G) * (world if args.scaling == 'weak' else 1)
        ms_int = tim['ms']['integrate'] / max(1, tim['launches']['integrate'])
        # the two Doppler scheme-2 columns count for k_integrate only when they ride on its units; with
        # 2-moment input they are units of their own on the vector ALUs (stage integrate_dop)
        flop_unit = FLOP_PER_UNIT + (FLOP_PER_UNIT_DOP2 if args.doppler == 2 and tim['doppler_units'] == 0 else 0)
        # Of a unit's 1024 terms only those whose float32 N(D) is not an exact zero are evaluated (the library
        # counts them); `achieved` prices the evaluated terms against the matrix-core peak, `algorithmic` is the
        # rate of the reference's full 1024-term contraction that the same launch stands for.
        evaluated = tim['evaluated_unit_diameters'] / (1024. * tim['valid_units']) if tim['valid_units'] else 1.
        flop_alg_launch = flop_unit * tim['valid_units'] / max(1, tim['launches']['integrate'])
        flop_per_launch = flop_alg_launch * evaluated
        achieved = flop_per_launch / (ms_int * 1e-3) / 1e12 if ms_int > 0 else 0.
        algorithmic = flop_alg_launch / (ms_int * 1e-3) / 1e12 if ms_int > 0 else 0.
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath) and not args.small:
            traffic = json.load(open(tpath)).get('k_integrate_hbm_bytes_per_launch')
        line = {
            'metric': METRIC, 'value': gates_per_step * args.steps / dt, 'unit': 'gates/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None, 'dtype': 'f32',
            'data': 'synthetic',
            'config': {'workload': 'CH_PPI_lema volume: 20 elevations x 360 azimuths x %d gates, COSMO-1 grid '
                                   '%dx%dx%d, %d-moment (%d classes), %dx%d Gauss-Hermite (%d sub-beams), attenuation on, '
                                   'Doppler scheme %d%s' % (G, shape[2], shape[1], shape[0], args.microphysics,
                                                            args.microphysics + 2, args.gh, args.gh, eng.ctx.n_sub,
                                                            args.doppler,
                                                            ' [SMALL DEBUG RUN - INVALID]' if args.small else ''),
                       'rays_per_rank': R, 'gates': G, 'sub_beams': eng.ctx.n_sub, 'classes': args.microphysics + 2,
                       'parallelism': 'rays sharded, %d rank(s)' % world},
            'roofline': {'bound': 'mfma', 'kernel': 'k_integrate', 'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS,
                         'unit': 'TFLOP/s', 'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'traffic': traffic,
                         'avg_launch_ms': ms_int, 'flop_per_launch': flop_per_launch, 'flop_per_unit': flop_unit,
                         'launches': tim['launches']['integrate'], 'evaluated_fraction': evaluated,
                         'algorithmic_tflops': algorithmic},
            'stage_ms_per_step': {k: v / args.steps for k, v in tim['ms'].items()},
            'valid_units_per_step': tim['valid_units'] / args.steps,
            'doppler_units_per_step': tim['doppler_units'] / args.steps,
            'doppler_evaluated_fraction': (tim['doppler_unit_diameters'] / (1024. * tim['doppler_units'])
                                           if tim['doppler_units'] else None),
            'finite_zh_fraction': finite, 'finite_rvel_fraction': finite_rvel,
            'cpu_baseline': cpu,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
