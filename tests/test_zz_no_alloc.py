"""GPU: cpol_run_rays_static, the entry point that allocates nothing (every workspace is a piece of the library's
static per-GPU array), against cpol_run_rays_dev through the C ABI: identical outputs, bit for bit, on two calls
in a row, the second with a smaller input; sizes beyond the stated maximum are refused."""
import ctypes

import numpy as np
import pytest

from cosmo_pol_amd import _lib
from tests.test_gpu_parity import make_engine, pipeline_conf, pipeline_model

pytestmark = pytest.mark.gpu


def _device_outputs(torch, n_rays, n_gates, dev):
    t = {'obs': torch.full((_lib.N_OBS, n_rays, n_gates), -7., dtype=torch.float32, device=dev),
         'rvel': torch.full((n_rays, n_gates), -7., dtype=torch.float64, device=dev),
         'mask': torch.full((n_rays, n_gates), -7., dtype=torch.float64, device=dev),
         'lat': torch.full((n_rays, n_gates), -7., dtype=torch.float64, device=dev),
         'lon': torch.full((n_rays, n_gates), -7., dtype=torch.float64, device=dev),
         'dist': torch.full((n_rays, n_gates), -7., dtype=torch.float32, device=dev),
         'height': torch.full((n_rays, n_gates), -7., dtype=torch.float32, device=dev),
         'ray_status': torch.full((n_rays,), -7, dtype=torch.int32, device=dev)}
    P = lambda x, typ: ctypes.cast(x.data_ptr(), typ)
    o = _lib.CpolOutputs(P(t['obs'], _lib.c_float_p), P(t['rvel'], _lib.c_double_p), P(t['mask'], _lib.c_double_p),
                         P(t['lat'], _lib.c_double_p), P(t['lon'], _lib.c_double_p), P(t['dist'], _lib.c_float_p),
                         P(t['height'], _lib.c_float_p), P(t['ray_status'], _lib.c_int_p))
    return t, o


def test_run_rays_static_equals_run_rays_dev():
    import torch
    dev = torch.device('cuda', 0)
    conf = pipeline_conf(2, 3, 3, dscheme=2)
    m = pipeline_model(2)
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    ctx, G = eng.ctx, eng.n_gates
    rng = np.random.default_rng(12)
    # two calls in a row on the same static array, the second smaller (its workspaces then overlay the first call's
    # leftovers at other offsets)
    for n_rays in (150, 37):
        assert n_rays * ctx.n_sub * G <= _lib.STATIC_MAX_POSITIONS
        az = torch.tensor(rng.uniform(0., 360., n_rays), dtype=torch.float64, device=dev)
        el = torch.tensor(rng.uniform(0.5, 40., n_rays), dtype=torch.float64, device=dev)
        want_t, want_o = _device_outputs(torch, n_rays, G, dev)
        ctx.run_rays_dev(az.data_ptr(), el.data_ptr(), n_rays, want_o)
        ctx.synchronize()
        got_t, got_o = _device_outputs(torch, n_rays, G, dev)
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info(0)[0]
        ctx.run_rays_static(az.data_ptr(), el.data_ptr(), n_rays, got_o)
        ctx.synchronize()
        assert torch.cuda.mem_get_info(0)[0] >= free_before          # no device memory was taken
        assert int(got_t['ray_status'].abs().sum().item()) == 0
        assert torch.isfinite(got_t['obs'][0]).float().mean().item() > 0.2
        for k in want_t:
            assert got_t[k].cpu().numpy().tobytes() == want_t[k].cpu().numpy().tobytes(), (n_rays, k)


def test_run_rays_static_refuses_what_it_cannot_hold():
    import torch
    dev = torch.device('cuda', 0)
    conf = pipeline_conf(2, 3, 3, dscheme=2)
    m = pipeline_model(2)
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    ctx, G = eng.ctx, eng.n_gates
    n_rays = _lib.STATIC_MAX_POSITIONS // (ctx.n_sub * G) + 1
    az = torch.zeros(n_rays, dtype=torch.float64, device=dev)
    el = torch.ones(n_rays, dtype=torch.float64, device=dev)
    with pytest.raises(_lib.CosmoPolError, match='CPOL_STATIC_MAX_POSITIONS'):
        ctx.run_rays_static(az.data_ptr(), el.data_ptr(), n_rays)
    # the largest size it takes runs, with the output buffers served from the static array too
    n_rays -= 1
    o = ctx.run_rays_static(az.data_ptr(), el.data_ptr(), n_rays)
    ctx.synchronize()
    ref = ctx.run_rays(np.zeros(n_rays), np.ones(n_rays))
    got = torch.empty((_lib.N_OBS, n_rays, G), dtype=torch.float32, device=dev)
    # copy the ctx-served result out of the static array through torch (device to device)
    ptr = ctypes.cast(o.obs, ctypes.c_void_p).value
    hip = ctypes.CDLL('libamdhip64.so')
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    assert hip.hipMemcpy(got.data_ptr(), ptr, got.numel() * 4, 3) == 0          # hipMemcpyDeviceToDevice
    assert got.cpu().numpy().tobytes() == ref['obs'].tobytes()
    # a configuration it does not serve is refused with a message, not run
    conf3 = pipeline_conf(1, 1, 1, dscheme=3)
    conf3['radar'].update({'PRI': 1000, 'FFT_length': 128})
    eng3 = make_engine(conf3)
    m1 = pipeline_model(1)
    eng3.upload_model(m1['fields'], m1['zlev'], m1['proj_info'], m1['resolution'])
    eng3.set_gauss_hermite()
    with pytest.raises(_lib.CosmoPolError, match='doppler_scheme 3'):
        eng3.ctx.run_rays_static(az.data_ptr(), el.data_ptr(), 2)


def test_run_rays_static_at_its_limit_with_reference_sized_tables():
    """The static array also holds the sort counters, which grow with the number of lookup-table nodes, not with the
    positions: the largest call (CPOL_STATIC_MAX_POSITIONS positions, 2-moment input, Doppler units of their own)
    with tables of the reference's size (46 elevations x 39 temperatures, compute_lut_sz.py:30-32) still fits, and
    gives what cpol_run_rays_dev gives."""
    import torch
    dev = torch.device('cuda', 0)
    conf = pipeline_conf(2, 3, 3, dscheme=2)
    m = pipeline_model(2)
    eng = make_engine(conf, n_e=46, n_t=None)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    ctx, G = eng.ctx, eng.n_gates
    n_rays = _lib.STATIC_MAX_POSITIONS // (ctx.n_sub * G)
    rng = np.random.default_rng(3)
    az = torch.tensor(rng.uniform(0., 360., n_rays), dtype=torch.float64, device=dev)
    el = torch.tensor(rng.uniform(0.5, 40., n_rays), dtype=torch.float64, device=dev)
    want_t, want_o = _device_outputs(torch, n_rays, G, dev)
    ctx.run_rays_dev(az.data_ptr(), el.data_ptr(), n_rays, want_o)
    ctx.synchronize()
    got_t, got_o = _device_outputs(torch, n_rays, G, dev)
    ctx.run_rays_static(az.data_ptr(), el.data_ptr(), n_rays, got_o)
    ctx.synchronize()
    for k in want_t:
        assert got_t[k].cpu().numpy().tobytes() == want_t[k].cpu().numpy().tobytes(), k
