"""CPU: the oracle (oracle/cosmo_oracle.py + oracle/oracle_c.c) against golden vectors
produced by running the reference (tests/golden/make_golden.py) and against the data
file the reference ships (interpolation/ex_bemas.txt -> tests/golden/ex_bemas.npz)."""
import numpy as np
import pytest

from oracle import cosmo_oracle as O
from tests import util


def test_known_lambda_rain():
    # hydrometeors/hydrometeors.py:567 quotes these two values
    h = O.Hydro('R', '1mom')
    h.set_psd(np.array([1e-3, 1e-4]))
    np.testing.assert_allclose(h.lambda_, [1.57084824, 2.62033279], rtol=2e-9)
    g = util.golden('psd')
    np.testing.assert_array_equal(h.lambda_, g['known_lambda_R'])


@pytest.mark.parametrize('kind,scheme', [('R', '1mom'), ('S', '1mom'), ('G', '1mom'), ('R', '2mom'),
                                         ('S', '2mom'), ('G', '2mom'), ('H', '2mom')])
def test_psd_parameters(kind, scheme):
    g = util.golden('psd')
    q, qn, t = g['q'], g['qn'], g['t']
    h = O.Hydro(kind, scheme)
    tag = kind + scheme[0]
    consts = np.array([h.a, h.b, h.alpha, h.beta, h.mu, h.nu, h.lambda_factor, h.vel_factor, h.ntot_factor])
    np.testing.assert_array_equal(consts, g['consts_' + tag])
    if scheme == '2mom':
        h.set_psd(qn, q)
    elif kind == 'S':
        h.set_psd(t, q)
    else:
        h.set_psd(q)
    d = np.linspace(h.d_min, h.d_max, 1024).astype('float32')
    with np.errstate(all='ignore'):
        N = h.get_N(d)
        v_int, n_int = h.integrate_V()
    np.testing.assert_array_equal(np.asarray(h.lambda_, dtype=np.float64), g['lambda_' + tag])
    np.testing.assert_array_equal(np.broadcast_to(np.asarray(h.N0, dtype=np.float64), q.shape), g['N0_' + tag])
    np.testing.assert_array_equal(np.asarray(N[::64], dtype=np.float64), g['N_' + tag])
    np.testing.assert_array_equal(np.asarray(v_int, dtype=np.float64), g['vint_' + tag])
    np.testing.assert_array_equal(np.broadcast_to(np.asarray(n_int, dtype=np.float64), q.shape), g['nint_' + tag])


def test_lookup_nearest_node():
    g = util.golden('lookup')
    luts = util.oracle_luts(1, 5.6)
    for h in ('R', 'S'):
        sl = luts[h].lookup_line(e=g['e'], t=g['t_' + h])
        np.testing.assert_array_equal(sl[:, ::128, :], g['line_' + h])


@pytest.mark.parametrize('name', ['obs_1mom_b1', 'obs_1mom_b9', 'obs_2mom_b9', 'obs_2mom_b3_x'])
def test_radar_observables(name):
    g = util.golden(name)
    cfg = util.cfg_from_meta(g['meta'])
    beams = util.oracle_beams(util.beams_from_golden(g))
    luts = util.oracle_luts(int(g['meta'][0]), float(g['meta'][4]))
    with np.errstate(all='ignore'):
        out = O.get_radar_observables(beams, luts, cfg)
    for k in ('ZH', 'ZDR', 'ZV', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'AH', 'AV'):
        assert out.values[k].dtype == g['out_' + k].dtype, k
        util.assert_close(out.values[k], g['out_' + k], rtol=1e-6, atol=0, name=name + ':' + k)
    np.testing.assert_array_equal(out.mask, g['out_mask'])


def test_pol_from_sz():
    g = util.golden('pol_from_sz')
    with np.errstate(all='ignore'):
        outs = O.get_pol_from_sz(g['sz'], float(g['wavelength']))
    for n, o in zip(['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'AH', 'AV', 'DELTA_HV'], outs):
        np.testing.assert_array_equal(o, g[n])


@pytest.mark.parametrize('ds', [1, 2, 3])
def test_doppler(ds):
    g = util.golden('doppler_s%d' % ds)
    cfg = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': ds},
                        radar={'frequency': 9.41, 'radial_resolution': 75., 'PRI': 700, 'FFT_length': 256})
    beams = util.oracle_beams(util.beams_from_golden(g))
    luts = util.oracle_luts(1, 9.41)
    with np.errstate(all='ignore'):
        out = O.get_doppler_velocity(beams, luts, cfg, get_refl=O.get_refl_c)
    util.assert_close(out.values['RVEL'], g['out_RVEL'], rtol=1e-9, atol=1e-12, name='RVEL')
    np.testing.assert_array_equal(out.mask, g['out_mask'])
    if ds == 3:
        np.testing.assert_array_equal(O.radar_constants(cfg)['VARRAY'], g['varray'])
        util.assert_close(out.values['DSPECTRUM'], g['out_DSPECTRUM'], rtol=1e-9, atol=0, name='DSPECTRUM')


def test_get_refl_c_and_numpy():
    g = util.golden('get_refl')
    args = (len(g['out']), g['Da'], g['Db'], g['D'], g['rcs'], g['N0'], g['mu'], g['lam'], g['nu'], g['step'], g['Dmin'])
    np.testing.assert_array_equal(O.get_refl_c(*args), g['out'])
    np.testing.assert_array_equal(O.get_refl_np(*args), g['out'])


def test_refraction_4_3():
    g = util.golden('refraction')
    coords = list(g['coords'])
    for tag in ('lema', 'mxpol'):
        rr = g['rr_' + tag]
        for el in (-0.2, 1.0, 3.3, 20.0, 40.0, 90.0):
            S, H, E = O.ref_4_3(rr, el, coords)
            np.testing.assert_array_equal(S, g['S_%s_%g' % (tag, el)])
            np.testing.assert_array_equal(H, g['H_%s_%g' % (tag, el)])
            np.testing.assert_array_equal(E, g['E_%s_%g' % (tag, el)])


def test_ex_bemas_refraction_and_geodesic():
    """ex_bemas.txt pins ref_4_3 and pyproj's WGS84 direct geodesic."""
    g = util.golden('ex_bemas')
    coords = [46.42561, 6.09996, 1672]        # option_files/CH_PPI_dole.yml:2
    rr = np.arange(250, 150000, 500.)
    for b in range(g['GH_pt'].shape[0]):
        S, H, E = O.ref_4_3(rr, g['GH_pt'][b, 1], coords)
        np.testing.assert_array_equal(S, g['dist_profile'][b])
        np.testing.assert_array_equal(H, g['heights_profile'][b])
        assert np.max(np.abs(E.view(np.int32) - g['elev_profile'][b].view(np.int32))) <= 1
        lon, lat = O.geod_fwd(coords[1], coords[0], g['GH_pt'][b, 0], g['dist_profile'][b].astype(np.float64))
        np.testing.assert_allclose(lat, g['lats_profile'][b], rtol=0, atol=1e-12)
        np.testing.assert_allclose(lon, g['lons_profile'][b], rtol=0, atol=1e-12)
    assert abs(g['GH_weight'].sum() - 1.0) < 1e-12


def test_interp_c_restatement():
    from tests import synthetic as syn
    g = util.golden('interp')
    nz, ny, nx, scheme, seed = [int(v) for v in g['seed_model']]
    m = syn.synthetic_model(nz=nz, ny=ny, nx=nx, scheme=scheme, seed=seed, llc=(-1.0, -0.8), res=(0.05, 0.04), ztop=9000.)
    seen = set()
    for name in ('T', 'QR_v', 'U', 'QNS_v'):
        out = O.interp_c(g['coords'], g['H'], m['fields'][name], m['zlev'], g['llc'], g['res'])
        want = g['out_' + name]
        np.testing.assert_array_equal(np.isnan(out), np.isnan(want))
        ok = ~np.isnan(want)
        np.testing.assert_array_equal(out[ok].view(np.int32), want[ok].view(np.int32))
        seen |= {'nan'} if np.isnan(want).any() else set()
        seen |= {'top'} if (want == -9999).any() else set()
    assert seen == {'nan', 'top'}      # the fixture exercises both sentinels


def test_cut_at_sensitivity():
    g = util.golden('cut')
    for tag, sens in (('scalar', -10), ('pair', [-5, 10000])):
        cfg = O.make_config(radar={'sensitivity': sens, 'radial_resolution': 500})
        vals = {'ZH': g['zh'].copy(), 'ZDR': np.ones(64, dtype='float32'), 'AH': np.ones(64, dtype='float32')}
        b = O.Beam(vals, np.zeros(64), None, None, np.zeros(64), None)
        with np.errstate(all='ignore'):
            O.cut_at_sensitivity([b], cfg)
        for k in ('ZH', 'ZDR', 'AH'):
            np.testing.assert_array_equal(b.values[k], g[k + '_' + tag])


def test_turbulence_spread_definition():
    """The turbulence spread the oracle defines (the reference's own is unreachable, doppler_sz.py:36,89, so this
    is unpinned): Doviak and Zrnic's two regimes with true division, one Gaussian per gate, power kept."""
    cfg = O.make_config(radar={'radial_resolution': 250., '3dB_beamwidth': 1.45, 'PRI': 1000, 'FFT_length': 128,
                               'frequency': 5.6})
    ranges = np.array([125., 5125., 120125., 60125.])
    edr = np.array([1e-3, 1e-3, 1e-3, 0.])
    std = O.get_turb_std(ranges, edr, cfg)
    th = 1.45 * np.pi / 180.
    # near gates: sigma_r >= 0.1 r sigma_theta (second regime); far gates: first regime
    assert 250. >= 0.1 * ranges[1] * th and 250. < 0.1 * ranges[2] * th
    np.testing.assert_allclose(std[2], ((ranges[2] * 1e-3 * th * 1.6 ** 1.5) / 0.72) ** (1. / 3.), rtol=1e-14)
    want1 = ((1e-3 * 250. * (1.35 * 1.6) ** 1.5) / (11. / 15. + 4. / 15. * (ranges[1] * th / 250.) ** 2) ** (-1.5)) ** (1. / 3.)
    np.testing.assert_allclose(std[1], want1, rtol=1e-14)
    assert std[3] == 0.
    v = O.radar_constants(cfg)['VARRAY']
    rng = np.random.default_rng(0)
    spec = rng.uniform(0., 1., (4, len(v))) * (rng.uniform(size=(4, len(v))) < 0.2)
    out = O.turb_spectrum_spread(spec, std, v)
    np.testing.assert_allclose(out.sum(axis=1), spec.sum(axis=1), rtol=1e-13)
    np.testing.assert_array_equal(out[3], spec[3])                     # no turbulence: untouched
    assert (out[:3] > 0).mean() > (spec[:3] > 0).mean()                # the others are spread out
    from scipy.ndimage import gaussian_filter1d
    f = gaussian_filter1d(spec[0], std[0] / (v[2] - v[1]))
    np.testing.assert_allclose(out[0], f / f.sum() * spec[0].sum(), rtol=1e-14)


def test_numpy_float32_power_vs_correctly_rounded():
    """The attenuation factors 10 ** (-0.1 A res_km) (scattering_sz.py:120-123) are float32 powers.  The HIP path takes
    the correctly rounded float32 value (float64 evaluation, one rounding); NumPy's float32 power may be vectorised
    and then is not correctly rounded.  Checked here, on whatever CPU runs the test, over a million arguments of the
    attenuation range: NumPy's value never differs from the correctly rounded one by more than one float32 ulp, and
    differs for at most a quarter of the arguments - which is what the long-ray parity test builds its bound on
    (one ulp per differing factor).  The measured share is printed (0 where NumPy's power is correctly rounded)."""
    rng = np.random.default_rng(1)
    a = (10. ** rng.uniform(-6, 1.5, 1_000_000)).astype(np.float32)        # specific attenuation, dB / km
    for res_km in (np.float32(0.075), np.float32(0.5)):
        x = -0.1 * a * res_km
        assert x.dtype == np.float32
        got = 10 ** x                                                        # the reference's expression
        assert got.dtype == np.float32
        want = O.pow10_rounded(x)
        # float64 pow is within an ulp of float64: its float32 rounding is the correctly rounded value except for
        # results within 2^-29 of a rounding boundary
        ulps = np.abs(got.view(np.int32).astype(np.int64) - want.view(np.int32))
        share = float((ulps != 0).mean())
        print('res %.3f km: NumPy float32 10**x differs from the correctly rounded value for %.1f %% of %d arguments, '
              'by %d ulp at most' % (float(res_km), 100 * share, x.size, int(ulps.max())))
        assert ulps.max() <= 1
        assert share <= 0.25
