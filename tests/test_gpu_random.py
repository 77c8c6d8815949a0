"""GPU: cpol_observables_from_beams (and, further down, the whole path) against the oracle on randomly drawn settings (microphysics scheme, Doppler
scheme 0 / 1 / 2, 1 - 25 sub-beams, 1 - 400 gates, attenuation correction on or off, radial resolution, share of empty
gates, contents left as they are or spread over ten decades).  The cases are fixed by their seeds; the bars are those
of tests/test_gpu_parity.py."""
import os

import numpy as np
import pytest

from oracle import cosmo_oracle as O
from tests import synthetic as syn, util
from tests.test_gpu_parity import beams_to_arrays, check_obs, make_engine

pytestmark = pytest.mark.gpu

# COSMOPOL_RANDOM_CASES=n draws n cases of either kind instead of the 16 and 6 that are run by default
N_MORE = int(os.environ.get('COSMOPOL_RANDOM_CASES', '0'))


@pytest.mark.parametrize('case', range(N_MORE or 16))
def test_random_settings_vs_oracle(case):
    rng = np.random.default_rng(7000 + case)
    scheme = int(rng.integers(1, 3))
    dscheme = int(rng.integers(0, 3))
    n_sub = int(rng.choice([1, 3, 9, 25]))
    n_gates = int(rng.integers(1, 400))
    att = int(rng.integers(0, 2))
    res = float(rng.choice([75., 250., 500.]))
    seed = int(rng.integers(0, 1 << 30))
    zero_frac = float(rng.uniform(0., 0.6))
    conf = O.make_config(microphysics={'scheme': scheme}, attenuation={'correction': att}, doppler={'scheme': dscheme},
                         radar={'radial_resolution': res, 'frequency': 5.6})
    eng = make_engine(conf)
    beams = syn.synthetic_beams(n_gates, n_sub, scheme, seed=seed, res=res, zero_frac=zero_frac,
                                nan_tail=min(7, n_gates // 3), below_head=min(2, n_gates // 4))
    if rng.uniform() < 0.4:
        for b in beams:
            for k, v in b['values'].items():
                if k.startswith('Q') and k.endswith('_v'):
                    v *= (10. ** rng.uniform(-8, 2, v.shape)).astype(np.float32)
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    luts = util.oracle_luts(scheme, 5.6)
    tag = 'scheme %d doppler %d sub-beams %d gates %d attenuation %d res %g' % (scheme, dscheme, n_sub, n_gates, att, res)
    with np.errstate(all='ignore'):
        # attenuation factors as correctly rounded float32 powers, which the kernel uses (see test_gpu_parity.py)
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf, pow10=O.pow10_rounded)
        check_obs(obs, want.values, tag, res)
        np.testing.assert_array_equal(mask, want.mask)
        if dscheme:
            want_v = O.get_doppler_velocity(util.oracle_beams(beams), luts, conf)
            util.assert_close(rvel, want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')


@pytest.mark.parametrize('case', range(N_MORE or 6))
def test_random_whole_path_vs_oracle(case):
    """cpol_run_rays (geometry, interpolation, PSD integration, reduction) for four rays of a randomly drawn model,
    radar site, quadrature and range setting; ranges and heights of the gates bit for bit."""
    from tests.test_gpu_parity import oracle_model_vars
    rng = np.random.default_rng(8000 + case)
    scheme = int(rng.integers(1, 3))
    dscheme = int(rng.integers(0, 3))
    nh, nv = [(1, 1), (3, 3), (3, 1), (5, 3)][int(rng.integers(0, 4))]
    res = float(rng.choice([250., 500.]))
    rmax = float(rng.choice([10000., 20000., 30000.]))
    coords = [46.0 + float(rng.uniform(-0.3, 0.3)), 8.5 + float(rng.uniform(-0.3, 0.3)), float(rng.uniform(300., 2500.))]
    conf = O.make_config(microphysics={'scheme': scheme}, doppler={'scheme': dscheme},
                         attenuation={'correction': int(rng.integers(0, 2))},
                         radar={'coords': coords, 'radial_resolution': res, 'range': rmax, 'frequency': 5.6,
                                '3dB_beamwidth': float(rng.uniform(0.8, 1.5))},
                         integration={'scheme': 1, 'nh_GH': nh, 'nv_GH': nv, 'weight_threshold': 1.})
    m = syn.synthetic_model(nz=int(rng.integers(8, 30)), ny=60, nx=64, scheme=scheme, seed=int(rng.integers(0, 1 << 30)),
                            llc=(-1.6, -1.5), res=(0.05, 0.05), pole=(-44.0, 8.5), ztop=float(rng.uniform(8000., 15000.)))
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    az = rng.uniform(0., 360., 4)
    el = np.concatenate([rng.uniform(0.5, 20., 3), rng.uniform(20., 89., 1)])
    out = eng.ctx.run_rays(az, el)
    assert (out['ray_status'] == 0).all()
    luts = util.oracle_luts(scheme, 5.6)
    dic = oracle_model_vars(m)
    for r in range(4):
        tag = 'scheme %d doppler %d GH %d x %d ray %d (az %.2f el %.2f)' % (scheme, dscheme, nh, nv, r, az[r], el[r])
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(dic, az[r], el[r], conf)
            # in the reference's order (radar_operator.py:242-243): the Doppler part first, on the beams as interpolated
            if dscheme:
                want_v = O.get_doppler_velocity(beams, luts, conf)
                util.assert_close(out['rvel'][r], want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL ' + tag)
            want = O.get_radar_observables(beams, luts, conf, pow10=O.pow10_rounded)
            if np.isfinite(want.values['ZH']).any():
                check_obs(out['obs'][:, r], want.values, tag, res)
            else:
                assert not np.isfinite(out['obs'][0, r]).any(), tag
            np.testing.assert_array_equal(out['mask'][r], want.mask)
        np.testing.assert_array_equal(out['dist'][r], want.dist_profile)
        np.testing.assert_array_equal(out['height'][r], want.heights_profile)
