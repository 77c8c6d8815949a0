"""CPU: host logic of the product package and the C-ABI surface (no compute calls)."""
import os
import re

import numpy as np
import pytest

from cosmo_pol_amd import _lib, cfg, engine, hydrometeors
from oracle import cosmo_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'cosmopol.h')).read()
    declared = set(re.findall(r'\b(cpol_[a-z_]+|get_all_radar_pts|get_refl)\s*\(', hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    lib = _lib.load()
    for name in declared:
        assert getattr(lib, name) is not None


def test_hydrometeor_constants_match_oracle():
    for k, s in [('R', '1mom'), ('S', '1mom'), ('G', '1mom'), ('R', '2mom'), ('S', '2mom'), ('G', '2mom'), ('H', '2mom')]:
        a, b = hydrometeors.create_hydrometeor(k, s), O.Hydro(k, s)
        for n in ('a', 'b', 'alpha', 'beta', 'mu', 'nu', 'lambda_factor', 'vel_factor', 'ntot_factor', 'd_min', 'd_max'):
            assert getattr(a, n) == getattr(b, n), (k, s, n)
    # doppler_sz.py:57 passes the integer scheme: 1-moment classes except hail
    assert hydrometeors.create_hydrometeor('R', 2).scheme == '1mom'
    assert hydrometeors.create_hydrometeor('H', 2).scheme == '2mom'


def test_gauss_hermite_subbeams_match_oracle():
    for nh, nv, thr in [(1, 1, 1.), (3, 3, 1.), (3, 9, 1.), (9, 9, 0.8), (3, 9, 0.95)]:
        conf = O.make_config(integration={'nh_GH': nh, 'nv_GH': nv, 'weight_threshold': thr},
                             radar={'3dB_beamwidth': 1.45})
        az, el, w = engine.gauss_hermite_subbeams(conf)
        subs, _ = O.gh_quadrature(conf)
        np.testing.assert_array_equal(az, [s[2] for s in subs])
        np.testing.assert_array_equal(el, [s[3] for s in subs])
        np.testing.assert_array_equal(w, [s[4] for s in subs])
        assert abs(w.sum() - 1) < 1e-12


def test_cfg_defaults_and_unknown_keys(tmp_path):
    p = tmp_path / 'o.yml'
    p.write_text("radar:\n    radial_res: 75\n    frequency: 9.41\nintegration:\n    na_GH: 7\n")
    cfg.init(str(p))
    assert cfg.CONFIG['radar']['radial_resolution'] == 500      # utilities/cfg.py:140-152 quirk
    assert cfg.CONFIG['radar']['frequency'] == 9.41
    assert cfg.CONFIG['microphysics']['scheme'] == 1
    cfg.init(str(tmp_path / 'missing.yml'))
    assert cfg.CONFIG == cfg.DEFAULTS


# ---------------------------------------------------------------------------------------------
# antenna quadrature (integration schemes 1, 2, 3, 5) against the reference's own run
# ---------------------------------------------------------------------------------------------
def _write_antenna(tmp_path, g):
    path = str(tmp_path / 'antenna.csv')
    np.savetxt(path, np.column_stack((g['antenna_angles'], g['antenna_db'])), delimiter=',')
    return path


def test_gautschi_quadrature_vs_golden():
    """quadrature.get_points_and_weights and utilities.vector_1d_to_polar as run by the reference."""
    import scipy.interpolate as interp
    from cosmo_pol_amd import quadrature
    g = np.load(os.path.join(GOLDEN, 'quadrature.npz'))
    ang = g['antenna_angles']
    power_sq = (10 ** (0.1 * g['antenna_db'])) ** 2
    fit = interp.interp1d(ang, power_sq, fill_value=0)
    for n in (1, 3, 5, 9):
        pts, w = quadrature.get_points_and_weights(fit, -ang.max(), ang.max(), n)
        np.testing.assert_allclose(pts, g['gautschi_pts_%d' % n], rtol=0, atol=1e-10)
        np.testing.assert_allclose(w, g['gautschi_w_%d' % n], rtol=1e-9, atol=1e-14)
    pts, w = quadrature.get_points_and_weights(num_points=4)
    np.testing.assert_allclose(pts, g['legendre_pts_4'], atol=1e-12)
    np.testing.assert_allclose(w, g['legendre_w_4'], rtol=1e-10)
    pol = quadrature.vector_1d_to_polar(ang, power_sq, g['polar_x'], g['polar_y'])
    np.testing.assert_allclose(pol, g['polar'], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize('name', ['gh_1x1', 'gh_3x3', 'gh_3x9_thr', 'multigauss', 'legendre_antenna', 'gautschi'])
def test_subbeam_set_vs_reference(name, tmp_path):
    """The sub-beam list (points, weights, order, thresholding) of get_profiles_GH as the reference builds it
    (interpolation.py:66-236, 260-266, 287-305)."""
    from cosmo_pol_amd import quadrature
    from tests.golden.make_golden import PROFILE_CASES
    g = np.load(os.path.join(GOLDEN, 'profiles.npz'))
    conf = O.make_config(radar={'3dB_beamwidth': 1.45}, integration=dict(PROFILE_CASES[name]['integration']))
    if conf['integration']['scheme'] in (3, 5):
        conf['integration']['antenna_diagram'] = _write_antenna(tmp_path, g)
    az, el, w = quadrature.subbeam_set(conf)
    pts = g[name + '_GH_pt']
    assert len(w) == len(pts)
    np.testing.assert_allclose(az + float(g['azimuth']), pts[:, 0], rtol=0, atol=1e-9)
    np.testing.assert_allclose(el + float(g['elevation']), pts[:, 1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(w, g[name + '_GH_weight'], rtol=1e-9, atol=1e-15)


def test_oracle_gh_quadrature_vs_reference():
    """The oracle's own scheme-1 set-up against the same run."""
    g = np.load(os.path.join(GOLDEN, 'profiles.npz'))
    from tests.golden.make_golden import PROFILE_CASES
    for name in ('gh_1x1', 'gh_3x3', 'gh_3x9_thr'):
        conf = O.make_config(radar={'3dB_beamwidth': 1.45}, integration=dict(PROFILE_CASES[name]['integration']))
        subs, _ = O.gh_quadrature(conf)
        np.testing.assert_allclose([s[2] + float(g['azimuth']) for s in subs], g[name + '_GH_pt'][:, 0], atol=1e-12)
        np.testing.assert_allclose([s[3] + float(g['elevation']) for s in subs], g[name + '_GH_pt'][:, 1], atol=1e-12)
        np.testing.assert_allclose([s[4] for s in subs], g[name + '_GH_weight'], rtol=1e-13)


def test_antenna_fit_recovers_gaussians():
    """optimize_gaussians (antenna_fit.py:36-67; it cannot run under Python 3 in the reference because of an
    integer division, so no golden): a diagram that IS a sum of Gaussians is fitted back to within 0.3 dB."""
    from cosmo_pol_amd import quadrature
    x = np.round(np.arange(-6., 6.0001, 0.1), 10)
    true = np.array([[0., 0., 0.62], [-26., 2.8, 0.4], [-26., -2.8, 0.4]])
    y = quadrature.gaussian_sum(x, true)
    p = quadrature.optimize_gaussians(x, y, 3)
    assert p.shape == (3, 3) and p[0, 0] == 0 and p[0, 1] == 0
    main = np.abs(x) < 1.5
    assert np.abs(quadrature.gaussian_sum(x, p) - y)[main].max() < 0.3


# ---------------------------------------------------------------------------------------------
# output container (radar/pyart_wrapper.py:107-232 without PyART)
# ---------------------------------------------------------------------------------------------
def _toy_radar(scan_type='ppi'):
    from cosmo_pol_amd.radar_output import SimulatedRadar
    rng = np.random.default_rng(0)
    n_sw, n_ray, n_g = 2, 3, 5
    zh = (10 ** rng.uniform(-1, 4, (n_sw * n_ray, n_g))).astype('float32')
    zh[0, 0] = 0.
    zh[1, 2] = np.nan
    fields = {'ZH': zh, 'ZDR': np.full_like(zh, 1.3), 'KDP': rng.normal(size=zh.shape).astype('float32'),
              'AH': np.full_like(zh, 0.5), 'AV': np.full_like(zh, 0.125),
              'DSPECTRUM': rng.uniform(size=zh.shape + (4,))}
    lat = rng.uniform(45, 47, zh.shape)
    els = np.repeat([1.0, 2.5], n_ray) if scan_type == 'ppi' else np.tile([1., 2., 3.], n_sw)
    azs = np.tile([0., 120., 240.], n_sw) if scan_type == 'ppi' else np.repeat([10., 190.], n_ray)
    r = SimulatedRadar(scan_type, els, azs, 250. + 500. * np.arange(n_g), fields, np.zeros(zh.shape), lat, lat + 1,
                       np.zeros(zh.shape), n_sw, n_ray, {'latitude': 46., 'longitude': 8., 'altitude': 1600., 'time': 0},
                       varray=np.linspace(-1, 1, 4))
    return r, zh


def test_simulated_radar_fields():
    r, zh = _toy_radar('ppi')
    d = r.fields['ZH']['data']
    # 0 -> NaN before the dB conversion, NaN masked (pyart_wrapper.py:159-161,193-195)
    assert d.mask[0, 0] and d.mask[1, 2] and d.mask.sum() == 2
    np.testing.assert_allclose(d[0, 1], 10 * np.log10(zh[0, 1]), rtol=1e-6)
    np.testing.assert_allclose(r.fields['ZDR']['data'][2, 2], 10 * np.log10(1.3), rtol=1e-6)
    assert r.fields['ZH']['units'] == 'dBZ' and r.fields['ZH']['valid_max'] == 55
    assert r.fields['KDP']['long_name'] == 'Specific diff. phase'
    assert r.fields['DSPECTRUM']['data'].shape == (6, 5, 4)            # stays linear
    np.testing.assert_array_equal(r.fields['ADP']['data'], 0.375)      # derived: AH - AV
    assert r.fields['range']['data'].shape == (6, 5) and r.fields['Latitude']['data'].shape == (6, 5)
    np.testing.assert_array_equal(r.fixed_angle['data'], [1.0, 2.5])
    np.testing.assert_array_equal(r.sweep_start_ray_index['data'], [0, 3])
    np.testing.assert_array_equal(r.sweep_end_ray_index['data'], [2, 5])
    assert r.get_field(1, 'ZH').shape == (3, 5)
    np.testing.assert_array_equal(r.get_azimuth(1), [0., 120., 240.])
    rr, _ = _toy_radar('rhi')
    np.testing.assert_array_equal(rr.fixed_angle['data'], [10., 190.])


def test_simulated_radar_netcdf_roundtrip(tmp_path):
    from scipy.io import netcdf_file
    r, zh = _toy_radar('ppi')
    path = str(tmp_path / 'scan.nc')
    r.to_netcdf(path)
    with netcdf_file(path, 'r', mmap=False) as nc:
        assert nc.dimensions['time'] == 6 and nc.dimensions['range'] == 5 and nc.dimensions['sweep'] == 2
        np.testing.assert_array_equal(nc.variables['azimuth'][:], r.azimuth['data'])
        np.testing.assert_array_equal(nc.variables['fixed_angle'][:], [1.0, 2.5])
        z = nc.variables['ZH'][:].copy()
        assert z[0, 0] == np.float32(-9999.) and z[1, 2] == np.float32(-9999.)
        np.testing.assert_allclose(z[0, 1], 10 * np.log10(zh[0, 1]), rtol=1e-6)
        assert nc.variables['ZH'].units == b'dBZ'
        assert nc.variables['DSPECTRUM'][:].shape == (6, 5, 4)
        assert float(nc.variables['altitude'].getValue()) == 1600.


# ---------------------------------------------------------------------------------------------
# lookup-table container and files (lookup/lut.py, lookup/read_lut.py)
# ---------------------------------------------------------------------------------------------
def _golden_lut(g, packed):
    from cosmo_pol_amd.lut import Lookup_table
    lut = Lookup_table()
    lut.add_axis('e', g['e']); lut.add_axis('t', g['t']); lut.add_axis('d', g['d']); lut.add_axis('sz', np.arange(12))
    with np.errstate(all='ignore'):
        lut.set_value_table(g['table'].copy(), map_to_int=packed)
    return lut


def test_lut_packing_and_lookups_vs_reference():
    g = np.load(os.path.join(GOLDEN, 'lutpack.npz'))
    with np.errstate(all='ignore'):
        plain, packed = _golden_lut(g, False), _golden_lut(g, True)
        np.testing.assert_array_equal(packed.value_table, g['int16'])
        np.testing.assert_array_equal(packed.range_orig_data, g['range'])
        np.testing.assert_array_equal(packed.get_float_data(packed.value_table), g['unpacked'])
        np.testing.assert_array_equal(plain.lookup_pts(g['coords'].copy()), g['pts_plain'])
        np.testing.assert_array_equal(packed.lookup_pts(g['coords'].copy()), g['pts_packed'])
    kw = dict(e=np.array([1.0, 3.0, 5.1]), t=np.array([263.0, 265.0, 270.0]))
    np.testing.assert_array_equal(plain.lookup_line(**kw), g['line_plain'])
    np.testing.assert_array_equal(packed.lookup_line(**kw), g['line_packed'])


def test_lut_file_roundtrip_and_frequency_interpolation(tmp_path):
    from cosmo_pol_amd import read_lut
    from tests import synthetic as syn
    from cosmo_pol_amd.lut import Lookup_table
    folder = str(tmp_path)
    tabs = {}
    for f in (5.6, 7.7):
        tabs[f] = {h: Lookup_table.from_arrays(*syn.synthetic_lut(h, f, n_e=3, n_t=4)) for h in ('R', 'S', 'G')}
        read_lut.save_lookup_tables(read_lut.lut_filename(1, f, folder), tabs[f], frequency=f, scheme=1)
    got = read_lut.get_lookup_tables('1mom', 5.6, folder=folder)
    assert sorted(got) == ['G', 'R', 'S']
    for h in got:
        assert isinstance(got[h].value_table, np.memmap)
        np.testing.assert_array_equal(got[h].value_table, tabs[5.6][h].value_table)
        assert got[h].axes_names == tabs[5.6][h].axes_names
        for a, b in zip(got[h].axes, tabs[5.6][h].axes):
            np.testing.assert_array_equal(a, b)
        assert got[h].axes_step == tabs[5.6][h].axes_step
    # a frequency without a file: linear interpolation between its neighbours, stored for the next call
    mid = read_lut.get_lookup_tables(1, 6.5, folder=folder)
    assert os.path.exists(read_lut.lut_filename(1, 6.5, folder))
    t1, t2 = tabs[5.6]['R'].value_table.astype(np.float64), tabs[7.7]['R'].value_table.astype(np.float64)
    np.testing.assert_allclose(mid['R'].value_table, t1 + (t2 - t1) / (7.7 - 5.6) * (6.5 - 5.6), rtol=1e-6)
    with pytest.raises(ValueError):
        read_lut.get_lookup_tables(1, 1.0, folder=folder)
    # int16 storage
    p = str(tmp_path / 'packed.clut')
    read_lut.save_lookup_tables(p, {'R': tabs[5.6]['R']}, pack_int16=True)
    back = read_lut.load_lookup_tables(p)['R']
    assert back.int_transform and back.value_table.dtype == np.int16
    ft = back.float_table()
    rng_ = tabs[5.6]['R'].value_table.max() - tabs[5.6]['R'].value_table.min()
    assert np.abs(ft - tabs[5.6]['R'].value_table).max() <= rng_ / 65534. * 1.01
    # plain-array export
    d = tmp_path / 'arrays'
    d.mkdir()
    e, t, dd, tab = syn.synthetic_lut('R', 5.6, n_e=3, n_t=4)
    for n, a in (('e', e), ('t', t), ('d', dd), ('table', tab)):
        np.save(str(d / ('R_%s.npy' % n)), a)
    arr = read_lut.from_arrays_dir(str(d))
    assert list(arr) == ['R']
    np.testing.assert_array_equal(arr['R'].value_table, tab)


# ---------------------------------------------------------------------------------------------
# refraction on the host (refraction/atm_refraction.py)
# ---------------------------------------------------------------------------------------------
def test_ref_4_3_host_vs_reference():
    from cosmo_pol_amd import refraction
    g = np.load(os.path.join(GOLDEN, 'refraction.npz'))
    for tag in ('lema', 'mxpol'):
        for el in (-0.2, 1.0, 3.3, 20.0, 40.0, 90.0):
            S, H, E = refraction.ref_4_3(g['rr_' + tag], el, list(g['coords']))
            np.testing.assert_array_equal(S, g['S_%s_%g' % (tag, el)])
            np.testing.assert_array_equal(H, g['H_%s_%g' % (tag, el)])
            np.testing.assert_array_equal(E, g['E_%s_%g' % (tag, el)])


def test_ref_ode_vs_reference():
    """Refraction scheme 2 against the reference's run (LSODA both sides; float32 outputs)."""
    from cosmo_pol_amd import refraction
    from cosmo_pol_amd.model import ModelVariable
    g = np.load(os.path.join(GOLDEN, 'refraction_ode.npz'))
    proj = {str(k): float(v) for k, v in zip(g['proj_keys'], g['proj_vals'])}
    N = ModelVariable('N', g['N'], {'z-levels': g['zlev'], 'proj_info': proj, 'resolution': list(g['resolution'])})
    for tag in ('c', 'x'):
        for el in (0.3, 1.0, 5.0, 30.0):
            S, H, E = refraction.get_radar_refraction(g['rr_' + tag], el, list(g['coords']), 2, N)
            np.testing.assert_allclose(S, g['S_%s_%g' % (tag, el)], rtol=2e-6, atol=1e-3)
            np.testing.assert_allclose(H, g['H_%s_%g' % (tag, el)], rtol=2e-6, atol=1e-3)
            np.testing.assert_allclose(E, g['E_%s_%g' % (tag, el)], rtol=2e-6, atol=1e-6)


def test_derived_model_variables():
    from cosmo_pol_amd import model
    shp = (3, 4, 5)
    raw = {'P': np.full(shp, 101325.), 'T': np.full(shp, 288.15), 'QV': np.zeros(shp), 'U': np.ones(shp), 'V': np.ones(shp),
           'W': np.zeros(shp), 'QR': np.full(shp, 1e-3), 'QS': np.zeros(shp), 'QG': np.zeros(shp), 'QNR': np.full(shp, 1e4)}
    out = model.derive_variables(raw, with_refractivity=True)
    assert abs(out['RHO'][0, 0, 0] - 101325. / (287.05 * 288.15 * (1 - 1e-3))) < 1e-6          # dry air + rain loading
    np.testing.assert_allclose(out['QR_v'], 1e-3 * out['RHO'], rtol=1e-6)
    np.testing.assert_allclose(out['QNR_v'], 1e4 * out['RHO'], rtol=1e-6)
    assert 'QH_v' not in out and out['T'].dtype == np.float32
    assert abs(out['N'][0, 0, 0] - 77.6 / 288.15 * 1013.25) < 1e-3                              # dry term only
    raw['QV'] = np.full(shp, 0.01)
    wet = model.derive_variables(raw, with_refractivity=True)
    assert wet['N'][0, 0, 0] > out['N'][0, 0, 0] + 40 and wet['RHO'][0, 0, 0] < out['RHO'][0, 0, 0]


# ----------------------------------------------------------------------------------------------------------
# GPM swath geometry and container (gpm/GPM_simulator.py, atm_refraction.py:125-149)
# ----------------------------------------------------------------------------------------------------------
def synthetic_dpr(n_lines=2, n_beams=5, lat0=46.0, lon0=8.5, alt=407e3, half_swath=60e3, heading=20.):
    """A DPR-like geometry: the sub-satellite point advances along `heading`, the footprints of a scan line lie
    across the track.  The footprints come from the oracle's direct geodesic (Karney), independent of the
    inverse solution under test."""
    from cosmo_pol_amd import gpm
    sc_lat, sc_lon = np.zeros(n_lines), np.zeros(n_lines)
    lat2, lon2 = np.zeros((n_lines, n_beams)), np.zeros((n_lines, n_beams))
    az_true = np.zeros((n_lines, n_beams))
    for i in range(n_lines):
        lo, la = O.geod_fwd(lon0, lat0, heading, np.array([5000. * i]))
        sc_lon[i], sc_lat[i] = lo[0], la[0]
        for j in range(n_beams):
            d = -half_swath + 2 * half_swath * (j + 0.37) / n_beams          # off-nadir, never exactly below
            a = heading + (90. if d >= 0 else -90.)
            lo, la = O.geod_fwd(sc_lon[i], sc_lat[i], a, np.array([abs(d)]))
            lon2[i, j], lat2[i, j] = lo[0], la[0]
            az_true[i, j] = a
    alts = np.full(n_lines, alt)
    pos = np.stack(gpm.geodetic_to_ecef(sc_lat, sc_lon, alts), axis=1)
    return {'Latitude': lat2, 'Longitude': lon2, 'scLat': sc_lat, 'scLon': sc_lon, 'dprAlt': alts, 'scPos': pos}, az_true


def test_gpm_angles():
    from cosmo_pol_amd import gpm
    d, az_true = synthetic_dpr()
    az, el, rng, coords = gpm.get_GPM_angles(d, 'Ku')
    assert az.shape == el.shape == rng.shape == (2, 5) and coords.shape == (2, 3)
    np.testing.assert_allclose((az - az_true + 180.) % 360. - 180., 0., atol=1e-8)
    np.testing.assert_array_equal(coords[:, 2], d['dprAlt'])
    # the elevation formula of GPM_simulator.py:221-232 written out for single beams
    for (i, j) in ((0, 0), (1, 3)):
        sx, sy, sz = gpm.geodetic_to_ecef(d['Latitude'][i, j], d['Longitude'][i, j], 0.)
        p = d['scPos'][i]
        r = np.sqrt((sx - p[0]) ** 2 + (sy - p[1]) ** 2 + (sz - p[2]) ** 2)
        H = np.sqrt(p[0] ** 2 + p[1] ** 2 + p[2] ** 2)
        RE = H - d['dprAlt'][i]
        theta = -np.arcsin((H ** 2 + r ** 2 - RE ** 2) / (2 * H * r)) / np.pi * 180.
        assert el[i, j] == -theta and rng[i, j] == r
    # ... and against the true geometry: angle at the satellite between the nadir and the footprint
    sx, sy, sz = gpm.geodetic_to_ecef(d['Latitude'], d['Longitude'], 0.)
    los = np.stack([sx, sy, sz], axis=-1) - d['scPos'][:, None, :]
    nadir = -d['scPos'][:, None, :]
    cosang = (los * nadir).sum(-1) / np.linalg.norm(los, axis=-1) / np.linalg.norm(nadir, axis=-1)
    np.testing.assert_allclose(el, 90. - np.degrees(np.arccos(cosang)), atol=0.3)
    assert (el > 80.).all() and (el < 90.).all()
    # ECEF: the equator / pole radii and a round trip of the height
    x, y, z = gpm.geodetic_to_ecef([0., 90.], [0., 0.], [0., 0.])
    np.testing.assert_allclose([x[0], z[1]], [6378137.0, 6356752.314245], rtol=1e-12)


def test_gpm_refraction_equals_oracle():
    from cosmo_pol_amd import refraction
    conf = O.make_config(radar={'coords': [46.1, 8.4, 407e3], 'range': 415e3, 'radial_resolution': 125., 'type': 'GPM'})
    for e in (89.5, 85.0, 81.3, 73.0):
        conf['radar']['range'] = 407e3 / np.sin(np.radians(e))          # about the distance to the ground
        got = refraction.get_GPM_refraction(e, conf)
        want = O.get_GPM_refraction(e, conf)
        for g, w in zip(got, want):
            assert g.dtype == np.float32 and g.tobytes() == w.tobytes()
        S, H, E = got
        assert 270 <= len(H) <= 360 and H.max() < 35000. and (np.diff(H) < 0).all() and (np.diff(S) > 0).all()
        assert abs(H.min()) < 6000.                               # the beam ends near the ellipsoid
    # sub-beams of one GPM ray (same range, different elevations) differ in length
    conf['radar']['range'] = 412e3
    assert len(refraction.get_GPM_refraction(85.0, conf)[0]) != len(refraction.get_GPM_refraction(85.7, conf)[0])


def test_simulated_gpm_container():
    from cosmo_pol_amd import gpm
    from cosmo_pol_amd.beam import Beam
    rng = np.random.default_rng(3)
    N, M = 2, 3
    prod, orc = [], []
    for idx in range(N * M):
        n = int(rng.integers(20, 30))
        mask = np.zeros(n)
        mask[:2] = -1                               # above the model top
        nb = int(rng.integers(0, 5))
        if nb:
            mask[n - nb:] = 1                        # below the terrain
        vals = {k: rng.uniform(1., 2., n) for k in ('ZH', 'ZDR')}
        lat, lon = rng.uniform(45, 46, n), rng.uniform(8, 9, n)
        dist, hgt = np.arange(n, dtype='float32'), np.arange(n, dtype='float32')[::-1].copy()
        prod.append(Beam({k: v.copy() for k, v in vals.items()}, mask.copy(), lat.copy(), lon.copy(), dist, hgt))
        orc.append(O.Beam({k: v.copy() for k, v in vals.items()}, mask.copy(), lat.copy(), lon.copy(), dist, hgt))
    sw = gpm.SimulatedGPM(prod, (N, M), 'Ku')
    bs, lats, lons, data = O.simulated_gpm(orc, (N, M), ['ZH', 'ZDR'])
    np.testing.assert_array_equal(sw.bin_surface, bs)
    np.testing.assert_array_equal(sw.lats, lats)
    np.testing.assert_array_equal(sw.lons, lons)
    for k in data:
        np.testing.assert_array_equal(sw.data[k], data[k])
    assert sw.band == 'Ku' and gpm.get_group('Ka') == 'HS'


def test_model_from_netcdf(tmp_path):
    """model.load_netcdf: half-level heights and W to full levels, contents per volume, rotated-pole metadata,
    and RadarOperator.load_model_file choosing the moment scheme from the species present."""
    from scipy.io import netcdf_file
    from cosmo_pol_amd import model, RadarOperator
    rng = np.random.default_rng(11)
    nz, ny, nx = 6, 5, 7
    path = str(tmp_path / 'lfff00000000.nc')
    hhl = np.linspace(12000., 400., nz + 1)[:, None, None] + rng.uniform(0, 50, (1, ny, nx))
    raw = {'P': 1e5 * np.exp(-0.5 * (hhl[:-1] + hhl[1:]) / 8000.), 'T': rng.uniform(230, 290, (nz, ny, nx)),
           'QV': rng.uniform(0, 5e-3, (nz, ny, nx)), 'U': rng.normal(size=(nz, ny, nx)), 'V': rng.normal(size=(nz, ny, nx)),
           'W': rng.normal(size=(nz + 1, ny, nx)), 'QR': rng.uniform(0, 1e-3, (nz, ny, nx)),
           'QS': rng.uniform(0, 1e-3, (nz, ny, nx)), 'QG': rng.uniform(0, 1e-3, (nz, ny, nx)),
           'QC': rng.uniform(0, 1e-4, (nz, ny, nx)), 'EDR': rng.uniform(0, 1e-2, (nz, ny, nx))}
    with netcdf_file(path, 'w') as f:
        for d, n in (('time', 1), ('level', nz), ('level1', nz + 1), ('rlat', ny), ('rlon', nx)):
            f.createDimension(d, n)
        f.createVariable('rlon', 'f8', ('rlon',))[:] = -1.0 + 0.02 * np.arange(nx)
        f.createVariable('rlat', 'f8', ('rlat',))[:] = 0.5 + 0.02 * np.arange(ny)
        rp = f.createVariable('rotated_pole', 'c', ())
        rp.grid_north_pole_latitude = 43.0
        rp.grid_north_pole_longitude = -170.0
        f.createVariable('HHL', 'f4', ('time', 'level1', 'rlat', 'rlon'))[0] = hhl
        for k, a in raw.items():
            lev = 'level1' if a.shape[0] == nz + 1 else 'level'
            f.createVariable(k, 'f4', ('time', lev, 'rlat', 'rlon'))[0] = a
    dic = model.load_netcdf(path, with_refractivity=True)
    assert set(dic) == {'U', 'V', 'W', 'T', 'RHO', 'QR_v', 'QS_v', 'QG_v', 'EDR', 'N'}
    at = dic['T'].attributes
    assert at['proj_info']['Latitude_of_southern_pole'] == -43.0 and at['proj_info']['Longitude_of_southern_pole'] == 10.0
    np.testing.assert_allclose([at['proj_info'][k] for k in ('Lo1', 'La1', 'Lo2', 'La2')], [-1.0, 0.5, -0.88, 0.58])
    np.testing.assert_allclose(at['resolution'], [0.02, 0.02])
    h32 = hhl.astype('float32').astype('float64')
    np.testing.assert_allclose(at['z-levels'], 0.5 * (h32[:-1] + h32[1:]), rtol=1e-7)
    f32 = lambda a: np.asarray(a, dtype='float32').astype('float64')
    np.testing.assert_allclose(dic['W'].data, 0.5 * (f32(raw['W'])[:-1] + f32(raw['W'])[1:]), rtol=1e-6, atol=1e-7)
    rho = model.air_density(f32(raw['P']), f32(raw['T']), f32(raw['QV']),
                            f32(raw['QC']) + f32(raw['QR']) + f32(raw['QS']) + f32(raw['QG']))
    np.testing.assert_allclose(dic['RHO'].data, rho, rtol=1e-6)
    np.testing.assert_allclose(dic['QS_v'].data, f32(raw['QS']) * rho, rtol=1e-6)
    assert dic['EDR'].data.dtype == np.float32 and dic['N'].data.shape == (nz, ny, nx)
    op = RadarOperator(O.make_config())
    op.load_model_file(path)
    assert op.current_microphys_scheme == 1 and 'EDR' in op.dic_vars and 'QH_v' not in op.dic_vars


def test_cut_at_sensitivity_three_elements_vs_oracle():
    """radar.sensitivity = [sensitivity, gain, SNR]: utilities.cut_at_sensitivity thresholds with the radar
    constant of constants.py:79-82 (utilities.py:74-76); the oracle restates the same and the closed form is
    checked against a hand evaluation."""
    from cosmo_pol_amd import utilities
    from cosmo_pol_amd.beam import Beam
    from cosmo_pol_amd.constants import constants
    conf = O.make_config(radar={'sensitivity': [-85., 42., 8.], 'radial_resolution': 500., 'range': 20000.,
                                'frequency': 5.6, '3dB_beamwidth': 1.45})
    cfg.set_config(conf)
    constants.update()
    wl = 299792458. / 5.6e9 * 1000
    want_rc = 180 - 10 * np.log10(np.pi ** 3 * 1.45 ** 2 * 0.93 ** 2 * 1000. / 1000) - 2 * 42. \
        + 10 * np.log10(1024 * np.log(2) * (wl / 1000.) ** 2)
    assert constants.RADAR_CONSTANT_DB == want_rc
    rng = np.random.default_rng(3)
    zh = (10 ** rng.uniform(-2, 4, 40)).astype('float32')
    mine = Beam({'ZH': zh.copy(), 'KDP': zh.copy(), 'AH': zh.copy()}, np.zeros(40), None, None, np.zeros(40), np.zeros(40))
    ob = O.Beam({'ZH': zh.copy(), 'KDP': zh.copy(), 'AH': zh.copy()}, np.zeros(40), None, None, np.zeros(40), np.zeros(40))
    utilities.cut_at_sensitivity([mine])
    O.cut_at_sensitivity([ob], conf)
    for k in ('ZH', 'KDP', 'AH'):
        np.testing.assert_array_equal(mine.values[k], ob.values[k])
    n_cut = int(np.isnan(mine.values['ZH']).sum())
    assert 0 < n_cut < 40 and not np.isnan(mine.values['AH']).any()
