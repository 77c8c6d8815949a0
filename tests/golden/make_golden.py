"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

The Python reference is imported from /root/reference through oracle/ref_shim.py and
its two C files are used through the gcc builds in oracle/_ref (make -C oracle ref).
This script contains no reference source; it only calls reference functions on the
deterministic synthetic inputs of cosmo_pol_amd/synthetic.py and stores inputs and
outputs.  The lookup tables are not stored (they are regenerated from the same
deterministic formula); everything else a test needs is in the .npz.

    make -C oracle ref && python tests/golden/make_golden.py
"""
import copy
import ctypes
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import ref_shim                      # noqa: E402
from tests import synthetic as syn               # noqa: E402
from tests.util import PROFILE_CASES             # noqa: E402

LUT_NE, LUT_NT = 8, 12      # reduced e / t axes keep the regenerated tables small


def ref_luts(scheme, frequency):
    from cosmo_pol.lookup.lut import Lookup_table
    out = {}
    for h in syn.hydrom_types(scheme):
        e, t, d, tab = syn.synthetic_lut(h, frequency, n_e=LUT_NE, n_t=LUT_NT)
        lut = Lookup_table()
        lut.add_axis('e', e)
        lut.add_axis('t', t)
        lut.add_axis('d', d)
        lut.add_axis('sz', np.arange(12))
        lut.set_value_table(tab)
        out[h] = lut
    return out


def ref_beams(beams):
    from cosmo_pol.utilities.beam import Beam
    out = []
    for b in beams:
        lat = np.linspace(46, 46.5, len(b['dist']))
        out.append(Beam(copy.deepcopy(b['values']), b['mask'].copy(), lat, lat + 1, b['dist'].copy(),
                        b['heights'].copy(), b['elev'].copy(), list(b['GH_pt']), b['GH_weight']))
    return out


def pack_beams(beams, prefix='in_'):
    d = {}
    for k in beams[0]['values']:
        d[prefix + 'val_' + k] = np.stack([b['values'][k] for b in beams])
    for k in ('mask', 'elev', 'dist', 'heights'):
        d[prefix + k] = np.stack([b[k] for b in beams])
    d[prefix + 'GH_pt'] = np.array([b['GH_pt'] for b in beams], dtype=np.float64)
    d[prefix + 'GH_weight'] = np.array([b['GH_weight'] for b in beams], dtype=np.float64)
    return d


def save(name, **arrs):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('wrote %-28s %7.1f kB' % (name + '.npz', os.path.getsize(path) / 1e3))


def gen_psd():
    ref_shim.install()
    from cosmo_pol.hydrometeors import hydrometeors as H
    rng = np.random.default_rng(3)
    n = 48
    q = (10 ** rng.uniform(-8, -2, n)).astype('float32')
    q[::7] = 0
    qn = (10 ** rng.uniform(0, 8, n)).astype('float32')
    qn[5::11] = 0
    t = rng.uniform(200, 310, n).astype('float32')
    out = {'q': q, 'qn': qn, 't': t}
    for h, sch in [('R', '1mom'), ('S', '1mom'), ('G', '1mom'), ('R', '2mom'), ('S', '2mom'), ('G', '2mom'), ('H', '2mom')]:
        hyd = H.create_hydrometeor(h, sch)
        d = np.linspace(hyd.d_min, hyd.d_max, 1024).astype('float32')
        if sch == '2mom':
            hyd.set_psd(qn, q)
        elif h == 'S':
            hyd.set_psd(t, q)
        else:
            hyd.set_psd(q)
        with np.errstate(all='ignore'):
            N = hyd.get_N(d)
            v_int, n_int = hyd.integrate_V()
        tag = h + sch[0]
        out['lambda_' + tag] = np.asarray(hyd.lambda_, dtype=np.float64)
        out['N0_' + tag] = np.broadcast_to(np.asarray(hyd.N0, dtype=np.float64), q.shape).copy()
        out['N_' + tag] = np.asarray(N[::64], dtype=np.float64)
        out['vint_' + tag] = np.asarray(v_int, dtype=np.float64)
        out['nint_' + tag] = np.broadcast_to(np.asarray(n_int, dtype=np.float64), q.shape).copy()
        out['consts_' + tag] = np.array([hyd.a, hyd.b, hyd.alpha, hyd.beta, hyd.mu, hyd.nu,
                                         hyd.lambda_factor, hyd.vel_factor, hyd.ntot_factor], dtype=np.float64)
    # known answer quoted in hydrometeors/hydrometeors.py:567
    r = H.create_hydrometeor('R', '1mom')
    r.set_psd(np.array([1e-3, 1e-4]))
    out['known_lambda_R'] = np.asarray(r.lambda_)
    save('psd', **out)


def gen_lookup():
    ref_shim.install()
    luts = ref_luts(1, 5.6)
    e = np.array([-3, 0, 0.999, 1, 1.0001, 3, 5, 6.5, 7, 13, 13.9, 14, 15, 90, 2.9999999], dtype='float32')
    out = {'e': e}
    for h in ('R', 'S'):
        lo = luts[h].axes_limits[1][0]
        t = (lo + np.array([-5, 0, 0.5, 1, 1.0000001, 3, 5, 7, 9, 9.5, 21, 22, 23, 40, 2.5], dtype='float32')).astype('float32')
        sl = luts[h].lookup_line(e=e, t=t)
        out['t_' + h] = t
        out['line_' + h] = sl[:, ::128, :].copy()
    save('lookup', **out)


def run_observables(scheme, n_gates, n_sub, att, freq, res, seed, nan_tail, below_head):
    cfg, constants = ref_shim.install({'microphysics': {'scheme': scheme},
                                       'attenuation': {'correction': att},
                                       'radar': {'frequency': freq, 'radial_resolution': res}})
    from cosmo_pol.scatter import scattering_sz
    beams = syn.synthetic_beams(n_gates, n_sub, scheme, seed=seed, res=res, nan_tail=nan_tail,
                                below_head=below_head)
    luts = ref_luts(scheme, freq)
    with np.errstate(all='ignore'):
        out = scattering_sz.get_radar_observables(ref_beams(beams), luts)
    d = pack_beams(beams)
    for k, v in out.values.items():
        d['out_' + k] = np.asarray(v)
    d['out_mask'] = np.asarray(out.mask)
    d['meta'] = np.array([scheme, n_gates, n_sub, att, freq, res, seed], dtype=np.float64)
    return d


def gen_observables():
    save('obs_1mom_b1', **run_observables(1, 96, 1, 1, 5.6, 75., 11, 9, 0))
    save('obs_1mom_b9', **run_observables(1, 64, 9, 0, 9.41, 500., 12, 6, 3))
    save('obs_2mom_b9', **run_observables(2, 80, 9, 1, 5.6, 500., 13, 8, 2))
    save('obs_2mom_b3_x', **run_observables(2, 48, 3, 1, 9.41, 300., 14, 0, 0))


def gen_pol():
    cfg, constants = ref_shim.install({'radar': {'frequency': 9.41}})
    from cosmo_pol.scatter import scattering_sz
    rng = np.random.default_rng(5)
    sz = np.abs(rng.standard_normal((64, 12))).astype('float32')
    sz[:, 1:3] *= -0.1
    sz[:, 5:7] *= 0.05
    sz[3] = np.nan
    with np.errstate(all='ignore'):
        outs = scattering_sz.get_pol_from_sz(sz)
    save('pol_from_sz', sz=sz, **{n: np.asarray(o) for n, o in
                                 zip(['ZH', 'ZV', 'ZDR', 'RHOHV', 'KDP', 'AH', 'AV', 'DELTA_HV'], outs)},
         wavelength=np.array(constants.WAVELENGTH))


def gen_doppler():
    for ds in (1, 2, 3):
        cfg, constants = ref_shim.install({'microphysics': {'scheme': 1}, 'doppler': {'scheme': ds},
                                           'radar': {'frequency': 9.41, 'radial_resolution': 75.,
                                                     'PRI': 700, 'FFT_length': 256}})
        from cosmo_pol.doppler import doppler_sz
        beams = syn.synthetic_beams(32, 3, 1, seed=21 + ds, res=75., elevation=8.0, nan_tail=3)
        luts = ref_luts(1, 9.41)
        with np.errstate(all='ignore'):
            out = doppler_sz.get_doppler_velocity(ref_beams(beams), luts)
        d = pack_beams(beams)
        d['out_RVEL'] = np.asarray(out.values['RVEL'])
        if 'DSPECTRUM' in out.values:
            d['out_DSPECTRUM'] = np.asarray(out.values['DSPECTRUM'])
            d['varray'] = np.asarray(constants.VARRAY)
        d['out_mask'] = np.asarray(out.mask)
        save('doppler_s%d' % ds, **d)


def gen_refraction():
    ref_shim.install()
    from cosmo_pol.refraction import atm_refraction
    coords = [46.04079, 8.83321, 1604]
    out = {'coords': np.array(coords)}
    for tag, rr in (('lema', np.arange(250., 150000., 500.)), ('mxpol', np.arange(37.5, 50000., 75.))):
        out['rr_' + tag] = rr
        for el in (-0.2, 1.0, 3.3, 20.0, 40.0, 90.0):
            S, H, E = atm_refraction.ref_4_3(rr, el, coords)
            out['S_%s_%g' % (tag, el)] = S
            out['H_%s_%g' % (tag, el)] = H
            out['E_%s_%g' % (tag, el)] = E
    save('refraction', **out)


def gen_interp():
    """Reference interpolation_c.c (gcc build) on a small terrain-following grid."""
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_ref', 'libref_interp.so'))
    fp = ctypes.POINTER(ctypes.c_float)
    lib.get_all_radar_pts.restype = fp
    m = syn.synthetic_model(nz=12, ny=40, nx=48, scheme=2, seed=99, llc=(-1.0, -0.8), res=(0.05, 0.04), ztop=9000.)
    zlev = m['zlev']
    rng = np.random.default_rng(17)
    n = 400
    llc = np.array([-1.0, -0.8], dtype='float32')
    res = np.array([0.05, 0.04], dtype='float32')
    coords = np.empty((n, 2), dtype='float32')
    coords[:, 0] = llc[1] + res[1] * rng.uniform(0.0, 38.99, n)      # rlat
    coords[:, 1] = llc[0] + res[0] * rng.uniform(0.0, 46.99, n)      # rlon
    coords[:8, 0] = llc[1] + res[1] * np.arange(8)                   # exactly on grid lines
    coords[8:16, 1] = llc[0] + res[0] * np.arange(8)
    H = rng.uniform(100., 9500., n).astype('float32')
    # exact level hits, lowest-layer gates, below-lowest-level-at-a-neighbour gates
    iy = np.floor((coords[:, 0] - llc[1]) / res[1]).astype(int)
    ix = np.floor((coords[:, 1] - llc[0]) / res[0]).astype(int)
    for g in range(16, 60):
        lev = rng.integers(0, 12)
        H[g] = zlev[lev, iy[g], ix[g]]
    for g in range(60, 120):
        H[g] = zlev[11, iy[g], ix[g]] + rng.uniform(-30, 200)
    out = {'coords': coords, 'H': H, 'llc': llc, 'res': res, 'seed_model': np.array([12, 40, 48, 2, 99])}
    i = ctypes.c_int
    p = lambda a: a.ctypes.data_as(fp)
    for name in ('T', 'QR_v', 'U', 'QNS_v'):
        data = np.ascontiguousarray(m['fields'][name])
        o = np.zeros(n, dtype='float32')
        lib.get_all_radar_pts(p(o), i(n), p(coords), i(n), i(2), p(H), i(n), p(data), i(12), i(40), i(48),
                              p(zlev), i(12), i(40), i(48), p(llc), i(2), p(res), i(2))
        out['out_' + name] = o
    save('interp', **out)


def gen_refl():
    """Reference doppler_c.c get_refl (gcc build) on random bins."""
    lib = ctypes.CDLL(os.path.join(ROOT, 'oracle', '_ref', 'libref_doppler.so'))
    fp = ctypes.POINTER(ctypes.c_float)
    rng = np.random.default_rng(23)
    nb = 40
    D = np.column_stack([np.linspace(0.2, hi, 1024).astype('float32') for hi in (8, 20, 15)]).astype('float32')
    step = (D[1] - D[0]).astype('float32')
    Dmin = np.array([0.2, 0.2, 0.2], dtype='float32')
    a = rng.uniform(0.2, 7.9, (nb, 3)).astype('float32')
    b = np.minimum(a + rng.uniform(0, 0.4, (nb, 3)), [8., 20., 15.]).astype('float32')   # clipped like doppler_sz.py:356
    b[::5] = a[::5]
    rcs = (1e-4 * D ** 6).astype('float32')
    N0 = np.array([7000., 300., 4000.], dtype='float32')
    mu = np.array([0.5, 0., 0.], dtype='float32')
    lam = np.array([2.2, 1.1, np.nan], dtype='float32')
    nu = np.array([1., 1., 1.], dtype='float32')
    o = np.zeros(nb, dtype='float32')
    i = ctypes.c_int
    p = lambda x: x.ctypes.data_as(fp)
    lib.get_refl(p(o), i(nb), p(a), i(nb), i(3), p(b), i(nb), i(3), p(D), i(1024), i(3), p(rcs), i(1024), i(3),
                 p(N0), i(3), p(mu), i(3), p(lam), i(3), p(nu), i(3), p(step), i(3), p(Dmin), i(3))
    save('get_refl', Da=a, Db=b, D=D, rcs=rcs, N0=N0, mu=mu, lam=lam, nu=nu, step=step, Dmin=Dmin, out=o)


def gen_cut():
    from tests import synthetic  # noqa: F401
    out = {}
    rng = np.random.default_rng(31)
    zh = (10 ** rng.uniform(-3, 4, 64)).astype('float32')
    zh[5] = np.nan
    for tag, sens in (('scalar', -10), ('pair', [-5, 10000])):
        cfg, constants = ref_shim.install({'radar': {'sensitivity': sens, 'radial_resolution': 500}})
        from cosmo_pol.utilities import utilities
        from cosmo_pol.utilities.beam import Beam
        vals = {'ZH': zh.copy(), 'ZDR': np.ones(64, dtype='float32'), 'AH': np.ones(64, dtype='float32')}
        b = Beam(vals, np.zeros(64), None, None, np.zeros(64), None)
        with np.errstate(all='ignore'):
            utilities.cut_at_sensitivity([b])
        out['ZH_' + tag] = b.values['ZH']
        out['ZDR_' + tag] = b.values['ZDR']
        out['AH_' + tag] = b.values['AH']
    save('cut', zh=zh, **out)


def synthetic_antenna():
    """One-way antenna diagram (angle in degrees, gain in dB): main lobe of 1.45 deg and two side lobes."""
    ang = np.round(np.arange(-6., 6.0001, 0.1), 10)
    sig = 1.45 / (2 * np.sqrt(2 * np.log(2)))
    lin = np.exp(-ang ** 2 / (2 * sig ** 2)) + 10 ** -2.6 * np.exp(-(np.abs(ang) - 2.8) ** 2 / (2 * 0.4 ** 2)) \
        + 10 ** -3.3 * np.exp(-(np.abs(ang) - 4.9) ** 2 / (2 * 0.5 ** 2))
    return ang, 10 * np.log10(lin) / 2.          # the reference squares the one-way gain (interpolation.py:124)


def gen_profiles():
    """interpolation.get_profiles_GH run by the reference itself (geodesy through stand-ins, see
    ref_shim.install_geo_standins): the sub-beam set of integration schemes 1, 2, 3 and 5 and the
    assembled beams on a small model."""
    import tempfile
    from oracle import cosmo_oracle as O
    m = syn.synthetic_model(nz=16, ny=60, nx=64, scheme=1, seed=5, llc=(-1.6, -1.5), res=(0.05, 0.05),
                            pole=(-44.0, 8.5), ztop=12000.)
    attrs = {'z-levels': m['zlev'], 'proj_info': m['proj_info'], 'resolution': m['resolution']}
    ang, gain_db = synthetic_antenna()
    tmp = tempfile.NamedTemporaryFile('w', suffix='.csv', delete=False)
    np.savetxt(tmp, np.column_stack((ang, gain_db)), delimiter=',')
    tmp.close()
    out = {'antenna_angles': ang, 'antenna_db': gain_db, 'names': np.array(sorted(PROFILE_CASES))}
    az, el = 123.0, 4.0
    for name in sorted(PROFILE_CASES):
        ov = copy.deepcopy(PROFILE_CASES[name])
        if ov['integration']['scheme'] in (3, 5):
            ov['integration']['antenna_diagram'] = tmp.name
        ov['radar'] = {'coords': [46.0, 8.5, 1500.], 'radial_resolution': 500., 'range': 20000.,
                       'frequency': 5.6, '3dB_beamwidth': 1.45}
        cfg, constants = ref_shim.install(ov)
        ref_shim.install_geo_standins()
        from cosmo_pol.interpolation import interpolation
        # an ordered mapping in the variable order the operator loads (radar_operator.py:32)
        dic = {n: O.ModelVar(m['fields'][n], attrs) for n in m['names']}
        with np.errstate(all='ignore'):
            beams = interpolation.get_profiles_GH(dic, az, el)
        out[name + '_GH_pt'] = np.array([b.GH_pt for b in beams], dtype=np.float64)
        out[name + '_GH_weight'] = np.array([b.GH_weight for b in beams], dtype=np.float64)
        out[name + '_mask'] = np.stack([b.mask for b in beams])
        out[name + '_elev'] = np.stack([b.elev_profile for b in beams])
        out[name + '_dist'] = np.stack([b.dist_profile for b in beams])
        out[name + '_heights'] = np.stack([b.heights_profile for b in beams])
        for v in ('T', 'QR_v', 'U'):
            out[name + '_val_' + v] = np.stack([b.values[v] for b in beams])
        print(name, len(beams), 'sub-beams, weight sum', out[name + '_GH_weight'].sum())
    os.unlink(tmp.name)
    out['azimuth'], out['elevation'] = az, el
    save('profiles', **out)


def gen_quadrature():
    """quadrature.get_points_and_weights (discrete Gautschi) and utilities.vector_1d_to_polar."""
    ref_shim.install()
    import scipy.interpolate as interp
    from cosmo_pol.interpolation import quadrature
    from cosmo_pol.utilities import utilities
    ang, gain_db = synthetic_antenna()
    power_sq = (10 ** (0.1 * gain_db)) ** 2
    out = {'antenna_angles': ang, 'antenna_db': gain_db}
    fit = interp.interp1d(ang, power_sq, fill_value=0)
    for n in (1, 3, 5, 9):
        pts, w = quadrature.get_points_and_weights(fit, -ang.max(), ang.max(), n)
        out['gautschi_pts_%d' % n], out['gautschi_w_%d' % n] = pts, w
    pts, w = quadrature.get_points_and_weights(num_points=4)
    out['legendre_pts_4'], out['legendre_w_4'] = pts, w
    x = np.polynomial.legendre.leggauss(5)[0] * ang.max()
    y = np.polynomial.legendre.leggauss(3)[0] * ang.max()
    out['polar_x'], out['polar_y'] = x, y
    out['polar'] = utilities.vector_1d_to_polar(ang, power_sq, x, y)
    save('quadrature', **out)


class _OldIndexArray(np.ndarray):
    """ndarray that accepts float indices the way NumPy < 1.12 did (truncation towards zero with a
    deprecation warning): the reference indexes with np.round(...) results (atm_refraction.py:57,59)."""
    def __getitem__(self, idx):
        fix = lambda i: int(i) if isinstance(i, (float, np.floating)) else i
        if isinstance(idx, tuple):
            idx = tuple(fix(i) for i in idx)
        else:
            idx = fix(idx)
        return np.asarray(super(_OldIndexArray, self).__getitem__(idx))


def gen_refraction_ode():
    """refraction/atm_refraction.py:45-84 (refraction scheme 2) run by the reference on a synthetic
    refractivity field (rotated-pole position of the radar through the stand-in)."""
    from oracle import cosmo_oracle as O
    cfg, constants = ref_shim.install()
    ref_shim.install_geo_standins()
    from cosmo_pol.refraction import atm_refraction
    m = syn.synthetic_model(nz=30, ny=60, nx=64, scheme=1, seed=5, llc=(-1.6, -1.5), res=(0.05, 0.05),
                            pole=(-44.0, 8.5), ztop=20000.)
    z = m['zlev'].astype(np.float64)
    # refractivity with a surface duct: strong gradient in the lowest 400 m above the terrain
    agl = z - z[-1][None]
    Nf = (320. * np.exp(-z / 7500.) + 60. * np.exp(-agl / 150.)).astype('float32')
    attrs = {'z-levels': m['zlev'].view(_OldIndexArray), 'proj_info': m['proj_info'], 'resolution': m['resolution']}
    Nobj = O.ModelVar(Nf.view(_OldIndexArray), attrs)
    coords = [46.0, 8.5, 1500.]
    out = {'N': Nf, 'zlev': m['zlev'], 'coords': np.array(coords), 'proj_keys': np.array(list(m['proj_info'].keys())),
           'proj_vals': np.array([float(v) for v in m['proj_info'].values()]), 'resolution': np.array(m['resolution'])}
    for tag, rr in (('c', np.arange(250., 100000., 500.)), ('x', np.arange(37.5, 20000., 75.))):
        out['rr_' + tag] = rr
        for el in (0.3, 1.0, 5.0, 30.0):
            S, H, E = atm_refraction.get_radar_refraction(rr, el, coords, 2, Nobj)
            out['S_%s_%g' % (tag, el)], out['H_%s_%g' % (tag, el)], out['E_%s_%g' % (tag, el)] = S, H, E
    save('refraction_ode', **out)


def gen_lutpack():
    """lookup/lut.py: int16 packing (float_to_uint16 / get_float_data), lookup_pts, lookup_line on a small table."""
    ref_shim.install()
    from cosmo_pol.lookup.lut import Lookup_table
    rng = np.random.default_rng(77)
    e, t, d = np.arange(0, 8, 2.), np.arange(262, 272, 2.), np.linspace(0.2, 8, 6).astype('float32')
    tab = (10 ** rng.uniform(-6, 1, (4, 5, 6, 12))) * rng.choice([-1, 1], (4, 5, 6, 12))
    tab[1, 2, 3, 4] = np.nan
    tab[0, 0, 0, 0] = np.inf
    out = {'e': e, 't': t, 'd': d, 'table': tab.copy()}
    coords = np.vstack([rng.uniform(0, 6, 40), rng.uniform(262, 270, 40), rng.uniform(0.2, 8, 40), rng.integers(0, 12, 40)])
    out['coords'] = coords
    for packed in (False, True):
        lut = Lookup_table()
        lut.add_axis('e', e); lut.add_axis('t', t); lut.add_axis('d', d); lut.add_axis('sz', np.arange(12))
        with np.errstate(all='ignore'):
            lut.set_value_table(tab.copy(), map_to_int=packed)
            pts = lut.lookup_pts(coords.copy())
            line = lut.lookup_line(e=np.array([1.0, 3.0, 5.1]), t=np.array([263.0, 265.0, 270.0]))
        tag = 'packed' if packed else 'plain'
        out['pts_' + tag] = pts
        out['line_' + tag] = line
        if packed:
            out['int16'] = lut.value_table
            out['range'] = np.array(lut.range_orig_data)
            with np.errstate(all='ignore'):
                out['unpacked'] = lut.get_float_data(lut.value_table)
    save('lutpack', **out)


if __name__ == '__main__':
    if len(sys.argv) > 1:
        for n in sys.argv[1:]:
            globals()['gen_' + n]()
        sys.exit(0)
    gen_psd()
    gen_lookup()
    gen_observables()
    gen_pol()
    gen_doppler()
    gen_refraction()
    gen_interp()
    gen_refl()
    gen_cut()
    gen_quadrature()
    gen_profiles()
    gen_lutpack()
    gen_refraction_ode()
