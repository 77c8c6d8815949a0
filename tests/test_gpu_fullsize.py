"""Full-size run of the headline workload (20 elevations x 360 azimuths x 500 gates on the 80 x 774 x 1158
grid, 2-moment, 3 x 3 sub-beams) checked through properties that need no CPU run of the whole volume, plus a
handful of its rays against the oracle.  Everything goes through the C ABI."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench                                                   # noqa: E402  workload definition only
from cosmo_pol_amd import _lib, engine                        # noqa: E402
from tests import synthetic as syn                             # noqa: E402
from cosmo_pol_amd.lut import Lookup_table                    # noqa: E402
from oracle import cosmo_oracle as O                           # noqa: E402
from tests import util                                         # noqa: E402
from tests.test_gpu_parity import check_obs                    # noqa: E402

pytestmark = pytest.mark.gpu
OBS = _lib.OBS_NAMES


@pytest.fixture(scope='module')
def volume():
    import copy
    conf = copy.deepcopy(bench.workload_config(False, doppler=2))
    azs, elv = bench.volume_rays(False)[:2]
    model = syn.synthetic_model(80, 774, 1158, scheme=2, seed=20161111, res=(0.01, 0.01))
    tabs = {h: syn.synthetic_lut(h, 5.6) for h in syn.hydrom_types(2)}
    eng = engine.Engine(conf, device=0)
    eng.upload_luts({h: Lookup_table.from_arrays(*t) for h, t in tabs.items()})
    eng.upload_model(model['fields'], model['zlev'], model['proj_info'], model['resolution'])
    eng.set_gauss_hermite()
    out = eng.ctx.run_rays(azs, elv)
    return dict(conf=conf, azs=azs, elv=elv, model=model, tabs=tabs, eng=eng, out=out)


def test_volume_sane(volume):
    out = volume['out']
    assert out['obs'].shape == (9, 7200, 500)
    assert (out['ray_status'] == 0).all()
    m = np.unique(out['mask'])          # mean of the sub-beam masks, [0, 1) -> 0 (scattering_sz.py:144-151)
    assert m.max() <= 1.0 and m.min() >= -1.0 and not ((m > 0) & (m < 1)).any()
    zh = out['obs'][OBS.index('ZH')]
    assert 0.3 < np.isfinite(zh).mean() < 0.95
    # gates the model does not cover carry no observables
    assert np.isnan(zh[out['mask'] == 1]).all()
    assert np.isfinite(out['rvel']).mean() > 0.3


def test_scan_identities(volume):
    """PHIDP is the sequential float32 range cumsum of 2 KDP times res / 1000, bit for bit as np.cumsum forms it
    (scattering_sz.py:117-118, utilities.py:22-24), and ZDR = ZH / ZV everywhere (:125), recomputed on the host
    from the other output fields of the whole volume."""
    out, conf = volume['out'], volume['conf']
    o = {n: out['obs'][i] for i, n in enumerate(OBS)}
    two_k = 2 * o['KDP']
    two_k[np.isnan(two_k)] = 0
    want = np.cumsum(two_k, axis=1) * conf['radar']['radial_resolution'] / 1000.
    assert want.dtype == np.float32
    np.testing.assert_array_equal(o['PHIDP'], want)
    ok = np.isfinite(o['ZH'])
    np.testing.assert_array_equal(o['ZDR'][ok], (o['ZH'] / o['ZV'])[ok])


def test_attenuation_is_cumprod_of_AH(volume):
    """ZH with correction = ZH without x cumprod(10^(-0.1 AH res_km)), inclusive (scattering_sz.py:120-125)."""
    eng, conf, out = volume['eng'], volume['conf'], volume['out']
    import copy
    conf0 = copy.deepcopy(conf)
    conf0['attenuation']['correction'] = 0
    eng.configure(conf0)
    sel = np.arange(0, 7200, 7)
    out0 = eng.ctx.run_rays(volume['azs'][sel], volume['elv'][sel])
    eng.configure(conf)
    res_km = conf['radar']['radial_resolution'] / 1000.
    for zn, an in (('ZH', 'AH'), ('ZV', 'AV')):
        z1 = out['obs'][OBS.index(zn)][sel].astype(np.float64)
        z0 = out0['obs'][OBS.index(zn)].astype(np.float64)
        a = out0['obs'][OBS.index(an)].astype(np.float64)
        np.testing.assert_array_equal(a, out['obs'][OBS.index(an)][sel])
        # float32 factors and a sequential float32 product, as the reference forms them.  The factor itself is
        # where implementations differ: NumPy's AVX-512 float32 power is one ulp off the correctly rounded value
        # (which the kernel uses) for 10-20 % of arguments, and those ulps add up along the ray (measured on random
        # rays of 500 gates: up to 7e-6)
        a32 = out0['obs'][OBS.index(an)]
        f = 10 ** (-0.1 * a32 * np.float32(res_km))
        f[np.isnan(f)] = 1
        fac = np.cumprod(f, axis=1)
        assert fac.dtype == np.float32
        ok = np.isfinite(z0)
        np.testing.assert_array_equal(np.isfinite(z1), ok)
        np.testing.assert_allclose(z1[ok], (z0 * fac)[ok], rtol=1e-5)
        # with the correctly rounded factors the product is reproduced bit for bit
        f = (10. ** (-0.1 * a32 * np.float32(res_km)).astype(np.float64)).astype(np.float32)
        f[np.isnan(f)] = 1
        z1_32, z0_32 = out['obs'][OBS.index(zn)][sel], out0['obs'][OBS.index(zn)]
        np.testing.assert_array_equal(z1_32[ok], (z0_32 * np.cumprod(f, axis=1))[ok])


def test_batching_and_ray_order_do_not_matter(volume):
    """Every output of a ray is bit-identical whatever batch it is processed in and whatever its position in
    the call (the bucket sort only decides which workgroup integrates a unit, never how)."""
    eng, out = volume['eng'], volume['out']
    rng = np.random.default_rng(5)
    perm = rng.permutation(7200)[:2500]
    eng.ctx.set_batch_units(3 * 1000 * 1000)
    try:
        out2 = eng.ctx.run_rays(volume['azs'][perm], volume['elv'][perm])
    finally:
        eng.ctx.set_batch_units(32 << 20)
    for k in ('obs', 'rvel', 'mask', 'lat', 'lon', 'dist', 'height'):
        a = out[k][:, perm] if k == 'obs' else out[k][perm]
        b = out2[k]
        assert a.tobytes() == b.tobytes(), k


def test_sampled_rays_vs_oracle(volume):
    """Six rays of the volume through the CPU restatement of the whole path."""
    conf, out, model = volume['conf'], volume['out'], volume['model']
    attrs = {'z-levels': model['zlev'], 'proj_info': model['proj_info'], 'resolution': model['resolution']}
    dic = {n: O.ModelVar(a, attrs) for n, a in model['fields'].items()}
    luts = {h: O.Lut(*t) for h, t in volume['tabs'].items()}
    for r in (3, 1111, 2345, 3600 + 17, 5000, 7199):
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(dic, volume['azs'][r], volume['elv'][r], conf)
            want_v = O.get_doppler_velocity(beams, luts, conf)
            want = O.get_radar_observables(beams, luts, conf)
            want.values.update(want_v.values)
        check_obs(out['obs'][:, r], want.values, 'ray %d' % r, conf['radar']['radial_resolution'])
        np.testing.assert_array_equal(out['mask'][r], want.mask)
        util.assert_close(out['rvel'][r], want.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL %d' % r)
        np.testing.assert_array_equal(out['dist'][r], want.dist_profile)
        np.testing.assert_array_equal(out['height'][r], want.heights_profile)


def test_bench_line_contract(capsys, monkeypatch):
    """bench.py prints one JSON line carrying the fields the measurement contract names (reduced grid, run
    in-process; the CPU baseline forks a worker pool and is left to the real bench run)."""
    import json
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--small', '--steps', '1', '--warmup', '1', '--no-cpu-baseline'])
    monkeypatch.delenv('RANK', raising=False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
              'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['unit'] == 'gates/s' and d['n_gpus'] == 1 and d['vs_baseline'] is None and d['dtype'] == 'f32'
    assert set(('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic')) <= set(d['roofline'])
    assert abs(d['roofline']['frac'] - d['roofline']['achieved'] / d['roofline']['peak']) < 1e-12
    assert d['value'] > 0 and d['steps'] == 1 and d['warmup'] == 1
    # the roofline prices the terms the integration evaluated; the skipped ones are exact float32 zeros
    ev = d['roofline']['evaluated_fraction']
    assert 0. < ev <= 1.
    assert abs(d['roofline']['algorithmic_tflops'] * ev - d['roofline']['achieved']) < 1e-9 * d['roofline']['achieved']
    assert d['doppler_evaluated_fraction'] is None or 0. < d['doppler_evaluated_fraction'] <= 1.


def test_geometry_rounding_against_indexing(volume):
    """S, H, E and the rotated coordinates are float32 roundings of float64 results; the device's float64 sin / cos /
    asin / atan2 differ from the host libm's in the last bit now and then, which can flip such a rounding.  200 rays
    of the volume (900 000 positions): how many of the float32 values differ from the CPU restatement's, and does
    any of those move a gate to another grid cell or model level (interpolation_c.c:42-82)?  Beam / gate indexing
    must be identical."""
    conf, model, eng = volume['conf'], volume['model'], volume['eng']
    rng = np.random.default_rng(77)
    sel = np.sort(rng.choice(7200, 200, replace=False))
    az, el = volume['azs'][sel], volume['elv'][sel]
    az_off, el_off, w = eng.quad
    geo = eng.ctx.geolocate(az, el)
    rr = O.radar_ranges(conf)
    p = model['proj_info']
    want = {k: np.empty_like(geo[k]) for k in ('dist', 'height', 'elev', 'coords')}
    for r in range(len(az)):
        for b in range(len(w)):
            S, H, E = O.ref_4_3(rr, el_off[b] + el[r], conf['radar']['coords'])
            lon, lat = O.geod_fwd(conf['radar']['coords'][1], conf['radar']['coords'][0], az_off[b] + az[r], S.astype(np.float64))
            want['dist'][r, b], want['height'][r, b], want['elev'][r, b] = S, H, E
            want['coords'][r, b] = O.wgs_to_rotated(lat, lon, p['Latitude_of_southern_pole'], p['Longitude_of_southern_pole'])
    n_diff = {}
    for k in want:
        d = np.abs(geo[k].view(np.int32).astype(np.int64) - want[k].view(np.int32))
        assert d.max() <= 1, k
        n_diff[k] = int((d != 0).sum())
    print('float32 values off by one ulp, of %d positions: %s' % (geo['dist'].size, n_diff))
    # positions where a coordinate or the height differs: grid cell and level count in the four neighbour columns
    bad = (geo['height'] != want['height']) | (geo['coords'] != want['coords']).any(axis=-1)
    zlev = np.asarray(model['zlev'], dtype=np.float32)
    llc = np.array([p['Lo1'], p['La1']], dtype=np.float32)
    res = np.asarray(model['resolution'], dtype=np.float32)

    def cell_and_levels(coords, H):
        iy = np.floor((coords[:, 0] - llc[1]) / res[1]).astype(int)
        ix = np.floor((coords[:, 1] - llc[0]) / res[0]).astype(int)
        lev = [(zlev[:, iy + dy, ix + dx] >= H[None, :]).sum(axis=0) for dy in (0, 1) for dx in (0, 1)]
        return np.stack([iy, ix] + lev)
    a = cell_and_levels(geo['coords'][bad], geo['height'][bad])
    b = cell_and_levels(want['coords'][bad], want['height'][bad])
    moved = int((a != b).any(axis=0).sum())
    print('positions with a differing coordinate or height: %d, of which in another cell or level: %d' % (int(bad.sum()), moved))
    assert moved == 0
    # measured: 1 of the 4.5 million float32 values (a rotated coordinate) differs, none moves a gate
    assert sum(n_diff.values()) <= 1e-5 * 5 * geo['dist'].size
