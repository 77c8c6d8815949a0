"""GPU: the HIP path, called through the C ABI, against the oracle and the golden vectors.

Tolerances (stated by the task): ZH / ZDR within 0.01 dB, KDP within 1e-3 deg/km, linear
observables within 1e-5 relative, gate / level indexing bit-exact.  The f32 interpolation is
compared bit for bit."""
import numpy as np
import pytest

from cosmo_pol_amd import _lib, engine
from cosmo_pol_amd.lut import Lookup_table
from oracle import cosmo_oracle as O
from tests import synthetic as syn, util

pytestmark = pytest.mark.gpu

OBS = _lib.OBS_NAMES


def make_engine(conf, n_e=util.LUT_NE, n_t=util.LUT_NT):
    eng = engine.Engine(conf, device=0)
    tabs = util.synthetic_lut_arrays(conf['microphysics']['scheme'], conf['radar']['frequency'], n_e, n_t)
    eng.upload_luts({h: Lookup_table.from_arrays(*t) for h, t in tabs.items()})
    return eng


def check_phidp(phidp, kdp, want_phidp, want_kdp, res, tag='', unseen=None):
    """PHIDP against the reference, explained by the KDP differences actually present.  Both sides accumulate their
    own float32 terms 2 KDP gate by gate (scattering_sz.py:118, utilities.py:22-24), so at gate g they can differ by
    the differences of the terms, sum_{i <= g} 2 |KDP_i - KDP_ref,i| res / 1000, plus the roundings of the two
    accumulations and of the final scaling: (g + 3) ulps of the largest partial sum there can have been.
    unseen: gates whose KDP the outputs no longer show (sensitivity cut) although it entered the sums; for those the
    difference is taken to be the bar KDP itself is held to."""
    ok = ~np.isnan(np.asarray(want_phidp))
    k = np.nan_to_num(np.asarray(kdp, dtype=np.float64))
    kw = np.nan_to_num(np.asarray(want_kdp, dtype=np.float64))
    res_km = res / 1000.
    dk = np.abs(k - kw)
    if unseen is not None:
        dk = np.where(unseen, 1e-5 * (np.abs(kw) + np.abs(kw).max(initial=0.)), dk)
    explained = np.cumsum(2 * dk) * res_km
    partial = np.cumsum(np.abs(2 * kw)) * res_km
    rounding = (np.arange(len(k)) + 3) * 2.0 ** -23 * partial
    err = np.abs(np.asarray(phidp, dtype=np.float64) - np.asarray(want_phidp, dtype=np.float64))
    over = (err - explained - rounding)[ok]
    assert over.max(initial=-1.) <= 0, "%s PHIDP: %.3e beyond what the KDP differences explain at gate %d" % (
        tag, over.max(), int(np.nonzero(ok)[0][np.argmax(over)]))


def check_obs(obs, want, tag, res, rtol=1e-5, exact_scan=True):
    """want: dict name -> f32 array (oracle or golden); res: radial resolution in metres.  Unless a sensitivity cut
    removed gates (exact_scan=False), PHIDP must also equal, bit for bit, the sequential float32 cumsum of the GPU's
    own 2 KDP times res / 1000 - the reference's accumulation (scattering_sz.py:118, utilities.py:22-24)."""
    for i, n in enumerate(OBS):
        w = np.asarray(want[n])
        g = obs[i]
        assert np.array_equal(np.isnan(g), np.isnan(w)), "%s %s: NaN pattern" % (tag, n)
        ok = ~np.isnan(w)
        if n in ('ZH', 'ZV', 'ZDR'):
            db = np.abs(10 * np.log10(g[ok].astype(np.float64) / w[ok].astype(np.float64)))
            assert db.max() < 0.01, "%s %s: %.3e dB" % (tag, n, db.max())
        if n == 'KDP':
            assert np.abs(g[ok] - w[ok]).max() < 1e-3, (tag, n)
        if n == 'PHIDP':
            # no tolerance of its own: the error must be explained by the KDP errors of the gates before
            check_phidp(g, obs[OBS.index('KDP')], w, want['KDP'], res, tag)
            continue
        scale = np.abs(w[ok]).astype(np.float64)
        atol = 1e-5 * (np.nanmax(np.abs(np.asarray(want['KDP'], dtype=np.float64)), initial=0.) + 1e-30) \
            if n in ('KDP', 'DELTA_HV') else 0.
        err = np.abs(g[ok].astype(np.float64) - w[ok]) - rtol * scale - atol
        assert err.max() <= 0, "%s %s: rel err %.3e at %d" % (
            tag, n, np.max(np.abs(g[ok] - w[ok]) / (scale + 1e-300)), np.argmax(err))
    if exact_scan:
        two_k = 2 * obs[OBS.index('KDP')]
        two_k[np.isnan(two_k)] = 0
        np.testing.assert_array_equal(obs[OBS.index('PHIDP')], np.cumsum(two_k) * res / 1000., err_msg=tag)


def beams_to_arrays(beams):
    vals = {k: np.stack([b['values'][k] for b in beams]) for k in beams[0]['values']}
    return (vals, np.stack([b['elev'] for b in beams]), np.stack([b['mask'] for b in beams]),
            [b['GH_weight'] for b in beams], [b['GH_pt'][0] for b in beams])


@pytest.mark.parametrize('name', ['obs_1mom_b1', 'obs_1mom_b9', 'obs_2mom_b9', 'obs_2mom_b3_x'])
def test_observables_vs_golden(name):
    """cpol_observables_from_beams == the reference's get_radar_observables (golden)."""
    g = util.golden(name)
    conf = util.cfg_from_meta(g['meta'], doppler={'scheme': 0})
    eng = make_engine(conf)
    beams = util.beams_from_golden(g)
    obs, mask, _ = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    check_obs(obs, {n: g['out_' + n] for n in OBS}, name, float(g['meta'][5]))
    np.testing.assert_array_equal(mask, g['out_mask'])


@pytest.mark.parametrize('scheme,n_gates,n_sub,att,dscheme', [
    (1, 667, 1, 1, 1), (1, 667, 9, 1, 1), (2, 500, 9, 1, 1), (2, 300, 27, 0, 1), (2, 1, 1, 1, 1), (2, 65, 3, 1, 1),
    (1, 667, 9, 1, 2), (1, 130, 1, 0, 2), (2, 200, 9, 1, 2)])
def test_observables_vs_oracle(scheme, n_gates, n_sub, att, dscheme):
    conf = O.make_config(microphysics={'scheme': scheme}, attenuation={'correction': att}, doppler={'scheme': dscheme},
                         radar={'radial_resolution': 75. if scheme == 1 else 500., 'frequency': 5.6})
    eng = make_engine(conf)
    beams = syn.synthetic_beams(n_gates, n_sub, scheme, seed=100 + n_gates, res=conf['radar']['radial_resolution'],
                                nan_tail=min(7, n_gates // 3), below_head=min(2, n_gates // 4))
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    luts = util.oracle_luts(scheme, 5.6)
    with np.errstate(all='ignore'):
        want_v = O.get_doppler_velocity(util.oracle_beams(beams), luts, conf)
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf)
    check_obs(obs, want.values, 'oracle', conf['radar']['radial_resolution'])
    np.testing.assert_array_equal(mask, want.mask)
    util.assert_close(rvel, want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')


@pytest.mark.parametrize('n_units', [1, 31, 33, 1023, 1024, 1025, 2049])
def test_one_bucket_with_chunk_boundary_counts(n_units):
    """All units of a class in a single (elevation, temperature) node, their number straddling the integration
    kernels' pass (32) and chunk (1024) sizes: every gate has its own lambda, so the chunk-local cutoff order
    permutes them, and the last chunk is a single unit or one short of full."""
    # Attenuation correction on: over 2049 gates of heavy precipitation the running float32 product and sum must
    # follow the reference's sequential accumulation (np.cumprod / np.cumsum on float32) to stay inside the bar.
    conf = O.make_config(microphysics={'scheme': 2}, attenuation={'correction': 1}, doppler={'scheme': 2},
                         radar={'frequency': 5.6})
    eng = make_engine(conf)
    beams = syn.synthetic_beams(n_units, 1, 2, seed=900 + n_units, zero_frac=0., nan_tail=0, below_head=0)
    rng = np.random.default_rng(n_units)
    b = beams[0]
    b['values']['T'][:] = 268.3
    b['elev'][:] = 3.1
    for h in 'RSGH':
        q = (10. ** rng.uniform(-6., -3., n_units)).astype('float32')
        lo, hi = syn.X_RANGE[h]
        b['values']['Q%s_v' % h][:] = q
        b['values']['QN%s_v' % h][:] = (q / np.exp(rng.uniform(np.log(lo), np.log(hi), n_units))).astype('float32')
    assert all((b['values'][k] > 0).all() for k in ('QR_v', 'QS_v', 'QG_v', 'QH_v'))
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    luts = util.oracle_luts(2, 5.6)
    with np.errstate(all='ignore'):
        # The attenuation factors 10 ** x as correctly rounded float32 values, which the kernel uses: NumPy's
        # AVX-512 float32 power is one ulp off that for 10-20 % of arguments, and over 2049 gates those ulps add
        # up to more than the bar (the reference's own result moves by as much with the CPU it runs on).
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf, pow10=O.pow10_rounded)
        want_np = O.get_radar_observables(util.oracle_beams(beams), luts, conf)
        want_v = O.get_doppler_velocity(util.oracle_beams(beams), luts, conf)
    # where the running product has left float32's normal range its bits are few: such gates are compared as zeros
    for w in (want, want_np):
        for zn in ('ZH', 'ZV', 'ZDR'):
            tiny = np.abs(w.values[zn]) < 1e-30
            w.values[zn][tiny] = obs[OBS.index(zn)][tiny]
    check_obs(obs, want.values, 'bucket of %d' % n_units, conf['radar']['radial_resolution'])
    # against NumPy's own power: one ulp per factor that differs from the correctly rounded one, counted here on
    # the reference's own attenuations (tests/test_host_cpu.py checks that such factors differ by one ulp at most)
    res_km = np.float32(conf['radar']['radial_resolution'] / 1000.)
    for zn, an in (('ZH', 'AH'), ('ZV', 'AV')):
        x = -0.1 * want_np.values[an] * res_km
        differing = np.cumsum(np.nan_to_num(10 ** x) != np.nan_to_num(O.pow10_rounded(x)))
        g_, w_ = obs[OBS.index(zn)].astype(np.float64), want_np.values[zn].astype(np.float64)
        assert np.all(np.abs(g_ / w_ - 1) <= 1e-5 + 2.0 ** -23 * differing), zn
    util.assert_close(rvel, want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')
    assert np.isfinite(obs[0]).all()


@pytest.mark.parametrize('scheme,dscheme', [(1, 0), (1, 2), (2, 2)])
def test_observables_with_extreme_contents(scheme, dscheme):
    """Mass and number densities scaled over 10 decades: the slope parameter of the size distributions then
    runs from values for which N(D) is a float32 zero beyond the first few diameters to values for which it
    never vanishes, all inside the same workgroup (the integration stops a wavefront at the last diameter any
    of its units needs, and must give what the full 1024-term sum gives)."""
    conf = O.make_config(microphysics={'scheme': scheme}, attenuation={'correction': 1}, doppler={'scheme': dscheme},
                         radar={'radial_resolution': 500., 'frequency': 5.6})
    eng = make_engine(conf)
    beams = syn.synthetic_beams(260, 9, scheme, seed=4242, res=500., zero_frac=0.1)
    rng = np.random.default_rng(99)
    for b in beams:
        for k, v in b['values'].items():
            if k.startswith('Q') and k.endswith('_v'):
                v *= (10. ** rng.uniform(-8, 2, v.shape)).astype(np.float32)
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    luts = util.oracle_luts(scheme, 5.6)
    with np.errstate(all='ignore'):
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf)
        if dscheme:
            want_v = O.get_doppler_velocity(util.oracle_beams(beams), luts, conf)
    check_obs(obs, want.values, 'extreme', 500.)
    np.testing.assert_array_equal(mask, want.mask)
    if dscheme:
        util.assert_close(rvel, want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')


def test_certified_term_cutoff():
    """The PSD integration stops a unit where its N(D) falls below 2^-E N(D_0); E < 150 only for classes whose tables
    carry the library's certificate that the omitted terms stay below 2^-28 of the sum.  Contents over 10 decades:
    the certified run equals the exact-zero run to float32 rounding and both meet the oracle; a table that grows too
    fast along the diameter axis gets no certificate and keeps the exact-zero rule."""
    conf = O.make_config(microphysics={'scheme': 2}, attenuation={'correction': 0}, doppler={'scheme': 0},
                         radar={'radial_resolution': 500., 'frequency': 5.6})
    eng = make_engine(conf)
    cut = eng.ctx.get_term_cutoff()
    assert sorted(cut) == ['G', 'H', 'R', 'S'] and all(48. <= e <= 128. for e in cut.values()), cut
    beams = syn.synthetic_beams(300, 9, 2, seed=31337, res=500., zero_frac=0.1)
    rng = np.random.default_rng(5)
    for b in beams:
        for k, v in b['values'].items():
            if k.startswith('Q') and k.endswith('_v'):
                v *= (10. ** rng.uniform(-8, 2, v.shape)).astype(np.float32)
    obs1, _, _ = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    eng.ctx.set_term_cutoff(False)
    assert all(e == 150. for e in eng.ctx.get_term_cutoff().values())
    obs0, _, _ = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    eng.ctx.set_term_cutoff(True)
    assert eng.ctx.get_term_cutoff() == cut
    for n in ('ZH', 'ZV', 'AH', 'AV', 'RHOHV'):
        a, b = obs1[OBS.index(n)].astype(np.float64), obs0[OBS.index(n)].astype(np.float64)
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(b)
        assert np.abs(a[ok] / b[ok] - 1).max() <= 2.5e-7, n          # two float32 roundings
    with np.errstate(all='ignore'):
        want = O.get_radar_observables(util.oracle_beams(beams), util.oracle_luts(2, 5.6), conf)
    check_obs(obs1, want.values, 'certified cutoff', 500.)
    check_obs(obs0, want.values, 'exact-zero cutoff', 500.)
    # snow table growing by 28 more decades along its diameter axis (41 in all): no exponent up to 128 is safe for
    # that class, the exact-zero rule still is
    tabs = dict(util.synthetic_lut_arrays(2, 5.6))          # a copy: the helper's dictionary is shared between tests
    e, t, d, tab = tabs['S']
    tabs['S'] = (e, t, d, (tab * 1e-20 * (10. ** (1.4 * d.astype(np.float64)))[None, None, :, None]).astype(np.float32))
    assert np.isfinite(tabs['S'][3]).all()
    eng2 = engine.Engine(conf, device=0)
    eng2.upload_luts({h: Lookup_table.from_arrays(*v) for h, v in tabs.items()})
    cut2 = eng2.ctx.get_term_cutoff()
    assert cut2['S'] == 150. and cut2['R'] == cut['R'], cut2
    obs2, _, _ = eng2.ctx.observables_from_beams(*beams_to_arrays(beams))
    with np.errstate(all='ignore'):
        want2 = O.get_radar_observables(util.oracle_beams(beams), {h: O.Lut(*v) for h, v in tabs.items()}, conf)
    # (values of this table are astronomically large: the relative bar only)
    for n in ('ZH', 'ZV', 'AH', 'AV', 'RHOHV'):
        a, b = obs2[OBS.index(n)].astype(np.float64), np.asarray(want2.values[n], dtype=np.float64)
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(b)
        assert np.abs(a[ok] / b[ok] - 1).max() <= 1e-5, n


@pytest.mark.parametrize('scheme,dscheme,decades', [(2, 2, (-40, -24)), (1, 0, (-12, -8)), (1, 2, (-12, -8))])
def test_combined_columns_against_all_twelve(scheme, dscheme, decades):
    """Units whose contributions to all twelve components are of normal float32 size are integrated over the seven
    combinations of columns the observables are made of, the others over all twelve columns
    (cpol_set_column_combination).  Contents over 10 decades, and a stretch of gates scaled down much further: with
    2-moment input until the components are subnormal float32 numbers, with 1-moment input (where the slope grows as the
    content falls) until they underflow one after the other and the reference's rule 'an exact zero becomes NaN'
    (scattering_sz.py:110) bites component by component.  The run with the combined columns and the run that takes
    all twelve columns for every unit have the same NaN pattern, agree to float32 rounding where nothing cancels, and
    both meet the oracle."""
    conf = O.make_config(microphysics={'scheme': scheme}, attenuation={'correction': 0}, doppler={'scheme': dscheme},
                         radar={'radial_resolution': 500., 'frequency': 5.6})
    eng = make_engine(conf)
    nodes = eng.ctx.get_column_combination()
    assert sorted(nodes) == sorted(syn.hydrom_types(scheme)) and all(n > 0 for n in nodes.values()), nodes
    beams = syn.synthetic_beams(300, 9, scheme, seed=777, res=500., zero_frac=0.1)
    rng = np.random.default_rng(11)
    tiny = slice(40, 110)
    for b in beams:
        for k, v in b['values'].items():
            if k.startswith('Q') and k.endswith('_v'):
                v *= (10. ** rng.uniform(-8, 2, v.shape)).astype(np.float32)
                v[tiny] *= (10. ** rng.uniform(decades[0], decades[1], v[tiny].shape)).astype(np.float32)
    obs1, mask1, rvel1 = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    eng.ctx.set_column_combination(False)
    assert eng.ctx.get_column_combination() == {}
    obs0, mask0, rvel0 = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    eng.ctx.set_column_combination(True)
    assert eng.ctx.get_column_combination() == nodes
    np.testing.assert_array_equal(mask1, mask0)
    if dscheme and scheme == 2:
        np.testing.assert_array_equal(rvel1, rvel0)            # Doppler units of their own do not depend on the form
    elif dscheme:
        # 1-moment input: the two Doppler moments ride on the polarimetric units - columns 12, 13 of the 14-column
        # image, or the spare combined column and 2 pi times the first one
        util.assert_close(rvel1, rvel0, rtol=2e-6, atol=1e-6, name='RVEL on / off')
        with np.errstate(all='ignore'):
            want_v = O.get_doppler_velocity(util.oracle_beams(beams), util.oracle_luts(scheme, 5.6), conf)
        util.assert_close(rvel1, want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')
        util.assert_close(rvel0, want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL all columns')
    with np.errstate(all='ignore'):
        want = O.get_radar_observables(util.oracle_beams(beams), util.oracle_luts(scheme, 5.6), conf)
    n_nan = {}
    # components that have left float32's normal range have few bits: values are compared where the reflectivity
    # (and with it the cross terms, four decades below) and the attenuation are well inside it
    with np.errstate(invalid='ignore'):
        normal = (np.abs(np.asarray(want.values['ZH'])) > 1e-26) & (np.abs(np.asarray(want.values['AH'])) > 1e-30)
    for i, n in enumerate(OBS):
        a, b, w = obs1[i].astype(np.float64), obs0[i].astype(np.float64), np.asarray(want.values[n], dtype=np.float64)
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b), err_msg=n)
        np.testing.assert_array_equal(np.isnan(a), np.isnan(w), err_msg=n)
        n_nan[n] = int(np.isnan(w[tiny]).sum())
        ok = ~np.isnan(w) & normal
        if n == 'RHOHV':               # made of float32 squares of components (scattering_sz.py:188), which are subnormal
            with np.errstate(invalid='ignore'):          # below 1e-19: a reflectivity of 1e-15 and less
                ok &= np.abs(np.asarray(want.values['ZH'])) > 1e-12
        if n in ('ZH', 'ZV', 'ZDR', 'AH', 'AV', 'RHOHV'):
            assert np.abs(a[ok] / b[ok] - 1).max() <= 1e-6, n
            assert np.abs(a[ok] / w[ok] - 1).max() <= 1e-5, n
            assert np.abs(b[ok] / w[ok] - 1).max() <= 1e-5, n
    # the regime is in the test: tiny but present components everywhere (2-moment), components lost one by one (1-moment)
    if scheme == 2:
        assert np.nanmax(np.abs(np.asarray(want.values['AH'])[tiny])) < 1e-29 and n_nan['AH'] == 0, n_nan
    else:
        assert 10 < n_nan['AH'] < n_nan['ZH'] < n_nan['RHOHV'] < 70, n_nan
    # the whole bar on the gates of ordinary contents
    keep = np.r_[0:40, 110:300]
    check_obs(obs1[:, keep], {n: np.asarray(v)[keep] for n, v in want.values.items()}, 'combined', 500., exact_scan=False)
    check_obs(obs0[:, keep], {n: np.asarray(v)[keep] for n, v in want.values.items()}, 'all twelve', 500., exact_scan=False)


def test_combined_columns_and_a_table_with_a_zero_column():
    """Hail whose cross terms s1, s2 are exact zeros (spheres): a gate that holds nothing but hail has those two
    components zero, which the reference turns into NaN (scattering_sz.py:110) and with them ZH, ZV, ZDR and RHOHV, while
    a gate that also holds another class is finite.  No node of that table qualifies for the combined columns, the
    other classes' nodes do, so the gates of the beam mix both forms of unit."""
    conf = O.make_config(microphysics={'scheme': 2}, attenuation={'correction': 1}, doppler={'scheme': 1},
                         radar={'radial_resolution': 500., 'frequency': 5.6})
    tabs = dict(util.synthetic_lut_arrays(2, 5.6))
    e, t, d, tab = tabs['H']
    tab = tab.copy()
    tab[..., 1] = 0.
    tab[..., 2] = 0.
    tabs['H'] = (e, t, d, tab)
    eng = engine.Engine(conf, device=0)
    eng.upload_luts({h: Lookup_table.from_arrays(*v) for h, v in tabs.items()})
    nodes = eng.ctx.get_column_combination()
    assert sorted(nodes) == ['G', 'R', 'S'], nodes
    beams = syn.synthetic_beams(240, 9, 2, seed=4711, res=500., zero_frac=0.2, nan_tail=5, below_head=2)
    for b in beams:
        inside = np.isfinite(b['values']['T'])          # gates outside the model carry NaN in every variable
        for k in ('QH_v', 'QNH_v'):
            v = b['values'][k]
            v[inside & ~(v > 0)] = v[v > 0].mean()
        for h in 'RSG':                       # gates 60..119: hail only
            b['values']['Q%s_v' % h][60:120] = 0.
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    with np.errstate(all='ignore'):
        luts = {h: O.Lut(*v) for h, v in tabs.items()}
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf)
    zh = np.asarray(want.values['ZH'])
    assert np.isnan(zh[60:120]).all() and np.isfinite(np.asarray(want.values['AH'])[60:120]).all()
    assert np.isfinite(zh[130:200]).any()
    check_obs(obs, want.values, 'zero column', 500.)
    np.testing.assert_array_equal(mask, want.mask)


def omitted_share_of_observable_combinations(tab, d_axis, mu, nu, cut_e, rng, n_samples=400):
    """What the certificate of a term cutoff claims, checked by brute force in float64: for random table nodes and
    random slopes, the share of each quantity the observables are made of (scattering_sz.py:168-194) that lies beyond
    the unit's cutoff - omitted terms taken at their worst, all of one sign, against the signed value kept; the two
    parts of (s4 + s7) + i (s6 - s5) against its modulus.  Returns the largest share found."""
    D = d_axis.astype(np.float64)
    x = D ** nu - (D ** nu).min()
    xq = x[::16]
    worst = 0.
    n_e, n_t = tab.shape[:2]
    for _ in range(n_samples):
        z = tab[rng.integers(n_e), rng.integers(n_t)].astype(np.float64)
        q = int(rng.integers(1, 64))                                   # the unit keeps q quads
        lo, hi = (cut_e + 2.) / xq[q], (cut_e + 2.) / xq[q - 1] if q > 1 else 8. * (cut_e + 2.) / xq[q]
        l2 = np.float32(lo * (hi / lo) ** rng.uniform(0.001, 0.999))
        nq = int((xq <= np.float32(cut_e + 2.) / l2).sum())          # the kernel's own count for this slope
        w = (D ** mu) * np.exp2(-float(l2) * x)
        g = np.stack([z[:, 0] - z[:, 1] - z[:, 2] + z[:, 3], z[:, 0] + z[:, 1] + z[:, 2] + z[:, 3],
                      z[:, 10] - z[:, 8], z[:, 4] + z[:, 7], z[:, 6] - z[:, 5], z[:, 9], z[:, 11]]) * w
        kept = np.abs(g.sum(axis=1))
        kept[3] = kept[4] = np.hypot(kept[3], kept[4])
        tail = np.abs(g[:, 16 * nq:]).sum(axis=1)
        share = np.where(tail > 0, tail / np.maximum(kept, 1e-300), 0.)
        worst = max(worst, float(share.max()))
    return worst


@pytest.mark.parametrize('case', ['nearly_spherical', 'oscillating'])
def test_term_cutoff_certifies_what_the_observables_use(case):
    """KDP is a difference of two columns, RHOHV and DELTA_HV come from (s4 + s7) + i (s6 - s5): a cutoff that is
    harmless for every column can still be felt by such a combination.  Two tables made to stress that:
    'nearly_spherical' - rain whose s10 and s8 agree to 1e-4, the difference growing with the fourth power of the
    diameter (the omitted tail holds far more of the difference than of either column); 'oscillating' - snow whose
    co-polar correlation turns through several periods along the diameter axis (sign changes in s4 .. s7, Mie-like).
    Whatever exponent the library certifies for such a class (or none: 150), the omitted share of every combination
    stays below 2^-28 for random units, and all observables meet the oracle at the usual bars."""
    from cosmo_pol_amd import hydrometeors
    conf = O.make_config(microphysics={'scheme': 2}, attenuation={'correction': 1}, doppler={'scheme': 0},
                         radar={'radial_resolution': 500., 'frequency': 5.6})
    tabs = dict(util.synthetic_lut_arrays(2, 5.6))
    if case == 'nearly_spherical':
        e, t, d, tab = tabs['R']
        tab = tab.copy()
        dn = (d.astype(np.float64) / d.max())[None, None, :]
        tab[..., 10] = (tab[..., 8].astype(np.float64) * (1. + 1e-4 * dn ** 4)).astype(np.float32)
        tabs['R'] = (e, t, d, tab)
        changed = 'R'
    else:
        e, t, d, tab = tabs['S']
        tab = tab.copy()
        dn = (d.astype(np.float64) / d.max())[None, None, :]
        cross = (tab[..., 4].astype(np.float64) + 1j * tab[..., 5]) * np.exp(1j * 5 * np.pi * dn)
        tab[..., 4] = tab[..., 7] = cross.real.astype(np.float32)
        tab[..., 5] = cross.imag.astype(np.float32)
        tab[..., 6] = -tab[..., 5]
        tabs['S'] = (e, t, d, tab)
        changed = 'S'
    eng = engine.Engine(conf, device=0)
    eng.upload_luts({h: Lookup_table.from_arrays(*v) for h, v in tabs.items()})
    cut = eng.ctx.get_term_cutoff()
    base = make_engine(conf).ctx.get_term_cutoff()
    print('term cutoff exponents, %s table: %s (plain synthetic tables: %s)' % (case, cut, base))
    assert cut[changed] >= base[changed]                      # never laxer than for the smooth table
    rng = np.random.default_rng(17)
    for h in 'RSGH':
        if cut[h] < 150.:
            hy = hydrometeors.create_hydrometeor(h, '2mom')
            share = omitted_share_of_observable_combinations(tabs[h][3], tabs[h][2], hy.mu, float(hy.nu), cut[h], rng)
            assert share <= 2.0 ** -28, (h, cut[h], share)
    beams = syn.synthetic_beams(300, 9, 2, seed=4711, res=500., zero_frac=0.1)
    rs = np.random.default_rng(6)
    for b in beams:
        for k, v in b['values'].items():
            if k.startswith('Q') and k.endswith('_v'):
                v *= (10. ** rs.uniform(-3, 1, v.shape)).astype(np.float32)
    luts = {h: O.Lut(*v) for h, v in tabs.items()}
    with np.errstate(all='ignore'):
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf)
    for certified in (True, False):
        eng.ctx.set_term_cutoff(certified)
        obs, _, _ = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
        tag = '%s, %s cutoff' % (case, 'certified' if certified else 'exact-zero')
        for n in ('ZH', 'ZV', 'ZDR'):
            g_, w_ = obs[OBS.index(n)].astype(np.float64), np.asarray(want.values[n], dtype=np.float64)
            np.testing.assert_array_equal(np.isnan(g_), np.isnan(w_), err_msg=tag)
            ok = ~np.isnan(w_)
            assert np.abs(10 * np.log10(g_[ok] / w_[ok])).max() < 0.01, (tag, n)
        for n in ('RHOHV', 'AH', 'AV'):
            util.assert_close(obs[OBS.index(n)], want.values[n], rtol=1e-5, name=tag + ' ' + n)
        # KDP (deg / km, with DELTA_HV in it as the reference adds it) and DELTA_HV (rad) on their absolute bars: the
        # difference of two columns that agree to 1e-4 keeps three digits of a float32 result row
        util.assert_close(obs[OBS.index('KDP')], want.values['KDP'], rtol=0, atol=1e-3, name=tag + ' KDP')
        util.assert_close(obs[OBS.index('DELTA_HV')], want.values['DELTA_HV'], rtol=1e-5, atol=1e-6, name=tag + ' DELTA_HV')


def test_observables_with_descending_diameter_axis():
    """A table whose diameter axis runs from large to small: no cutoff ordering applies (the library only uses it on
    ascending axes) and the exponent reference is the smallest diameter wherever it sits; dD = D[1] - D[0] is
    negative, as the reference would compute it (scattering_sz.py:50)."""
    conf = O.make_config(microphysics={'scheme': 2}, attenuation={'correction': 0}, doppler={'scheme': 0},
                         radar={'radial_resolution': 500., 'frequency': 5.6})
    tabs = {h: (e, t, d[::-1].copy(), tab[:, :, ::-1, :].copy())
            for h, (e, t, d, tab) in util.synthetic_lut_arrays(2, 5.6).items()}
    eng = engine.Engine(conf, device=0)
    eng.upload_luts({h: Lookup_table.from_arrays(*v) for h, v in tabs.items()})
    beams = syn.synthetic_beams(150, 3, 2, seed=77, res=500.)
    obs, mask, _ = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    with np.errstate(all='ignore'):
        want = O.get_radar_observables(util.oracle_beams(beams), {h: O.Lut(*v) for h, v in tabs.items()}, conf)
    check_obs(obs, want.values, 'descending axis', 500.)
    np.testing.assert_array_equal(mask, want.mask)


@pytest.mark.parametrize('ds', [1, 2, 3])
def test_doppler_vs_golden(ds):
    """get_doppler_velocity (schemes 1, 2 and 3) against the reference's output."""
    g = util.golden('doppler_s%d' % ds)
    conf = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': ds},
                         radar={'frequency': 9.41, 'radial_resolution': 75., 'PRI': 700, 'FFT_length': 256})
    eng = make_engine(conf)
    beams = util.beams_from_golden(g)
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    rv_atol = 0.
    if ds == 3:
        np.testing.assert_allclose(eng.varray, g['varray'], rtol=0, atol=0)
        rv_atol = check_spectrum(eng.ctx.last_spectrum, g['out_DSPECTRUM'], g['varray'],
                                 O.spectrum_open_bins(util.oracle_beams(beams), conf))
    check_rvel(rvel, g['out_RVEL'], rv_atol)
    np.testing.assert_array_equal(mask, g['out_mask'])


def check_spectrum(got, want, varray, open_bins=None):
    """Doppler spectrum [gates, velocity bins] against the reference / oracle.  Returns, per gate, how far RVEL may
    move because of table terms that sit in the neighbouring bin (see below).

    Every bin is a float32 sum of float64 terms (doppler_c.c:24-26), scaled and GH-weighted.  Which terms a bin
    takes is decided by its edges' fall speed W + wind/tan(el) - v/sin(el) (doppler_sz.py:341), evaluated with
    float32 sin and tan: a few m/s left over from tens or hundreds, so the last bit of sin / tan moves all edges
    of a gate by about 1e-6 relative, and with ~1000 diameters per class about one edge per gate and sub-beam
    then falls on the other side of a table node.  NumPy's float32 sin / tan are not correctly rounded (5 % /
    25 % of arguments differ from the rounded exact value with its AVX512 kernels, which also differ from
    libm's), so the reference's own spectrum changes in this way with the CPU it runs on.  The kernel uses the
    correctly rounded values.

    An edge is shared by two neighbouring bins, so a term that crosses it leaves one bin and enters the other;
    only the outer edges of a sub-beam's first and last spectrum rows (open_bins, from oracle.spectrum_open_bins)
    have no neighbour, there the term leaves or joins the spectrum.  Nothing else can differ.  Asserted: identical
    zero pattern; every bin to 2e-6 relative + 1e-6 of the gate's peak, EXCEPT runs of neighbouring bins that
    exchanged terms, and for each such run (with the bin on either side) the sum over the run is conserved to the
    same bar unless it holds an open bin (then it may change by single terms, each below 5 % of the peak); fewer
    than 1 % of the bins are in such runs."""
    assert got.shape == want.shape
    np.testing.assert_array_equal(got == 0, want == 0)
    peak = np.abs(want).max(axis=1, keepdims=True)
    tol = 2e-6 * np.abs(want) + 1e-6 * peak
    bad = np.abs(got - want) > tol
    assert bad.mean() < 0.01, "%d of %d bins differ" % (bad.sum(), bad.size)
    vres = float(varray[1] - varray[0])
    rv_atol = np.zeros(got.shape[0])
    for g in np.nonzero(bad.any(axis=1))[0]:
        idx = np.nonzero(bad[g])[0]
        runs = np.split(idx, np.nonzero(np.diff(idx) > 1)[0] + 1)
        for run in runs:
            lo, hi = max(run[0] - 1, 0), min(run[-1] + 2, got.shape[1])
            moved = np.abs(got[g, lo:hi] - want[g, lo:hi]).sum() / 2.
            n_open = int(open_bins[g, lo:hi].sum()) if open_bins is not None else 0
            assert abs(got[g, lo:hi].sum() - want[g, lo:hi].sum()) <= tol[g, lo:hi].sum() + 0.05 * peak[g, 0] * n_open, \
                "gate %d bins %d..%d: terms were not just exchanged between neighbours" % (g, lo, hi - 1)
            if n_open:      # a term that left the spectrum moves RVEL by up to its share of the power times the bin's velocity
                rv_atol[g] += abs(got[g, lo:hi].sum() - want[g, lo:hi].sum()) * 2 * np.abs(varray).max() / abs(want[g].sum())
            # the exchanged terms are single table terms, small against the gate's power
            assert moved <= 0.05 * peak[g, 0] * len(run)
            rv_atol[g] += moved * vres / abs(want[g].sum())
    return rv_atol


def check_rvel(rvel, want, rv_atol, name='RVEL'):
    """RVEL = sum v S / sum S (doppler_sz.py:99) to 1e-5 relative + 1e-6 m/s, plus the shift of the terms that
    check_spectrum found in the neighbouring bin (each moves by one bin width)."""
    rvel, want = np.asarray(rvel, dtype=np.float64), np.asarray(want, dtype=np.float64)
    np.testing.assert_array_equal(np.isnan(rvel), np.isnan(want), err_msg=name)
    ok = ~np.isnan(want)
    err = np.abs(rvel[ok] - want[ok]) - (1e-5 * np.abs(want[ok]) + 1e-6 + 1.01 * np.broadcast_to(rv_atol, want.shape)[ok])
    assert err.max(initial=-1.) <= 0, "%s: %.3e m/s beyond the bar at %d" % (name, err.max(), int(np.argmax(err)))


@pytest.mark.parametrize('n_gates,n_sub', [(40, 1), (24, 9)])
def test_doppler_spectrum_vs_oracle(n_gates, n_sub):
    """Scheme 3 on other seeds / sub-beam counts against the CPU restatement, with the observables alongside."""
    conf = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': 3},
                         radar={'frequency': 5.6, 'radial_resolution': 250., 'PRI': 1000, 'FFT_length': 128})
    eng = make_engine(conf)
    beams = syn.synthetic_beams(n_gates, n_sub, 1, seed=77 + n_sub, res=250., elevation=12.0, nan_tail=4, below_head=2)
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    luts = util.oracle_luts(1, 5.6)
    with np.errstate(all='ignore'):
        want_v = O.get_doppler_velocity(util.oracle_beams(beams), luts, conf, get_refl=O.get_refl_c)
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf)
    check_obs(obs, want.values, 'oracle', 250.)
    rv_atol = check_spectrum(eng.ctx.last_spectrum, want_v.values['DSPECTRUM'], eng.varray,
                             O.spectrum_open_bins(util.oracle_beams(beams), conf))
    check_rvel(rvel, want_v.values['RVEL'], rv_atol)


def test_spectrum_chain_cut_is_exact():
    """doppler_c.c:24-26 adds a bin's terms to a float accumulator one after the other.  Where a bin holds 256
    diameters or more (low elevations: the whole size range falls into one or two velocity bins) the kernel ends
    the accumulation once every term still to come is below 2^-27 of the sum.  Such terms cannot change a float
    sum, so the spectra with and without the shortcut are the same bits."""
    conf = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': 3},
                         radar={'frequency': 5.6, 'radial_resolution': 250., 'PRI': 1000, 'FFT_length': 128})
    eng = make_engine(conf)
    for elevation, seed in ((0.7, 5), (2.0, 6), (12.0, 7)):
        beams = syn.synthetic_beams(48, 9, 1, seed=seed, res=250., elevation=elevation, nan_tail=3, below_head=1)
        rng = np.random.default_rng(seed)
        for b in beams:                     # contents over six decades: short and long tails of N(D)
            for k, v in b['values'].items():
                if k.startswith('Q') and k.endswith('_v'):
                    v *= (10. ** rng.uniform(-5, 1, v.shape)).astype(np.float32)
        arrays = beams_to_arrays(beams)
        eng.ctx.set_term_cutoff(True)
        _, _, rvel1 = eng.ctx.observables_from_beams(*arrays)
        spec1 = np.array(eng.ctx.last_spectrum)
        eng.ctx.set_term_cutoff(False)
        _, _, rvel0 = eng.ctx.observables_from_beams(*arrays)
        spec0 = np.array(eng.ctx.last_spectrum)
        eng.ctx.set_term_cutoff(True)
        assert np.isfinite(spec0).any() and (spec0[np.isfinite(spec0)] > 0).any()
        np.testing.assert_array_equal(spec1.view(np.int64), spec0.view(np.int64))
        np.testing.assert_array_equal(np.asarray(rvel1).view(np.int64), np.asarray(rvel0).view(np.int64))


def _beams_with_edr(n_gates, n_sub, seed):
    beams = syn.synthetic_beams(n_gates, n_sub, 1, seed=seed, res=250., elevation=12.0, nan_tail=4, below_head=2)
    rng = np.random.default_rng(seed)
    for b in beams:
        edr = (10. ** rng.uniform(-5., -1., n_gates)).astype('float32')           # m^2 s^-3
        edr[rng.uniform(size=n_gates) < 0.1] = 0.                                 # no turbulence: no spread
        edr[np.isnan(b['values']['T'])] = np.nan
        b['values']['EDR'] = edr
    return beams


def test_turbulence_spread_of_one_subbeam():
    """With one sub-beam the spectrum without broadening is the sub-beam's own, so broadening it on the host
    (the oracle's gaussian_filter1d per gate + power rescaling) must reproduce the kernel's broadened spectrum
    to rounding: this isolates the convolution from the float32 sin / tan sensitivity of the bin edges.
    Parity of the broadening itself is unpinned: the reference cannot reach it (doppler_sz.py:36,89)."""
    n_gates = 48
    base = dict(microphysics={'scheme': 1}, radar={'frequency': 5.6, 'radial_resolution': 250., 'PRI': 1000,
                                                   'FFT_length': 128, '3dB_beamwidth': 1.45})
    beams = _beams_with_edr(n_gates, 9, seed=404)[5:6]          # one off-axis sub-beam with its weight
    conf3 = O.make_config(doppler={'scheme': 3}, **base)
    eng = make_engine(conf3)
    eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    plain = eng.ctx.last_spectrum.copy()
    conf4 = O.make_config(doppler={'scheme': 4}, **base)
    eng.configure(conf4)
    _, _, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    got = eng.ctx.last_spectrum
    consts = O.radar_constants(conf4)
    ranges = np.arange(125., 125. + 250. * n_gates, 250.)
    std = O.get_turb_std(ranges, beams[0]['values']['EDR'], conf4)
    assert np.nanmax(std) / (consts['VARRAY'][2] - consts['VARRAY'][1]) > 3.          # several bins wide somewhere
    want = O.turb_spectrum_spread(plain, std, consts['VARRAY'])
    assert (plain != 0).sum() > 100 and (want != plain).any()
    peak = np.abs(want).max(axis=1, keepdims=True)
    err = np.abs(got - want) / np.maximum(peak, 1e-300)
    assert err.max() < 1e-12, "gate %d: %.3e of the peak" % (int(err.max(axis=1).argmax()), err.max())
    np.testing.assert_allclose(got.sum(axis=1), plain.sum(axis=1), rtol=1e-12)          # power is kept
    with np.errstate(all='ignore'):
        rv = (consts['VARRAY'][None, :] * want).sum(axis=1) / want.sum(axis=1)
    util.assert_close(rvel, rv, rtol=1e-9, atol=1e-9, name='RVEL')
    # doppler.turbulence_correction on top of scheme 3 is the same request
    conf3t = O.make_config(doppler={'scheme': 3, 'turbulence_correction': 1}, **base)
    eng.configure(conf3t)
    eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    np.testing.assert_array_equal(eng.ctx.last_spectrum, got)


def test_turbulence_spread_vs_oracle():
    """Scheme 4 (nine sub-beams, each spread with its own eddy dissipation rates) against the CPU restatement;
    the bins a table term enters from the neighbouring bin (see check_spectrum) are spread out with the rest, so
    the bar is 2e-6 relative + 2e-4 of the gate's peak."""
    conf = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': 4},
                         radar={'frequency': 5.6, 'radial_resolution': 250., 'PRI': 1000, 'FFT_length': 128,
                                '3dB_beamwidth': 1.45})
    eng = make_engine(conf)
    beams = _beams_with_edr(24, 9, seed=405)
    _, _, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    got = eng.ctx.last_spectrum
    luts = util.oracle_luts(1, 5.6)
    with np.errstate(all='ignore'):
        want_v = O.get_doppler_velocity(util.oracle_beams(beams), luts, conf, get_refl=O.get_refl_c)
    want = want_v.values['DSPECTRUM']
    peak = np.abs(want).max(axis=1, keepdims=True)
    assert (peak > 0).sum() > 5
    err = (np.abs(got - want) - 2e-6 * np.abs(want)) / np.maximum(peak, 1e-300)
    assert err.max() < 2e-4, "gate %d: %.3e of the peak" % (int(err.max(axis=1).argmax()), err.max())
    np.testing.assert_allclose(got.sum(axis=1), want.sum(axis=1), rtol=2e-4)
    util.assert_close(rvel, want_v.values['RVEL'], rtol=1e-5, atol=0.02, name='RVEL')


def test_doppler_scheme3_needs_one_moment():
    conf = O.make_config(microphysics={'scheme': 2}, doppler={'scheme': 3})
    with pytest.raises(RuntimeError):
        engine.Engine(conf, device=0)


def test_observables_all_invalid():
    """No hydrometeors anywhere: every observable is NaN, PHIDP is 0 (nan_cumsum)."""
    conf = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': 0})
    eng = make_engine(conf)
    beams = syn.synthetic_beams(40, 3, 1, seed=3)
    for b in beams:
        for k in ('QR_v', 'QS_v', 'QG_v'):
            b['values'][k][:] = 0
    obs, mask, _ = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    with np.errstate(all='ignore'):
        want = O.get_radar_observables(util.oracle_beams(beams), util.oracle_luts(1, 5.6), conf)
    for i, n in enumerate(OBS):
        np.testing.assert_array_equal(np.isnan(obs[i]), np.isnan(want.values[n]))
    np.testing.assert_array_equal(obs[OBS.index('PHIDP')], want.values['PHIDP'])


def small_model(scheme=2):
    return syn.synthetic_model(nz=12, ny=40, nx=48, scheme=scheme, seed=99, llc=(-1.0, -0.8), res=(0.05, 0.04), ztop=9000.)


def test_interp_points_bit_exact():
    g = util.golden('interp')
    m = small_model()
    conf = O.make_config(microphysics={'scheme': 2})
    eng = engine.Engine(conf, device=0)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    out = eng.ctx.interp_points(g['coords'], g['H'])
    names = list(m['fields'].keys())
    for name in ('T', 'QR_v', 'U', 'QNS_v'):
        got, want = out[names.index(name)], g['out_' + name]
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want)
        np.testing.assert_array_equal(got[ok].view(np.int32), want[ok].view(np.int32))
    # every variable against the C restatement on fresh random points
    rng = np.random.default_rng(8)
    n = 5000
    coords = np.empty((n, 2), dtype='float32')
    coords[:, 0] = -0.8 + 0.04 * rng.uniform(0, 38.99, n)
    coords[:, 1] = -1.0 + 0.05 * rng.uniform(0, 46.99, n)
    H = rng.uniform(100., 9500., n).astype('float32')
    out = eng.ctx.interp_points(coords, H)
    llc = np.array([-1.0, -0.8], dtype='float32')
    res = np.array([0.05, 0.04], dtype='float32')
    for i, name in enumerate(names):
        want = O.interp_c(coords, H, m['fields'][name], m['zlev'], llc, res)
        np.testing.assert_array_equal(np.isnan(out[i]), np.isnan(want))
        ok = ~np.isnan(want)
        np.testing.assert_array_equal(out[i][ok].view(np.int32), want[ok].view(np.int32))


@pytest.mark.parametrize('case', ['in_range', 'wide', 'tiny', 'nonfinite', 'thin_level'])
def test_interp_division_forms(case):
    """interpolation_c.c:151 divides by the level thickness once per variable.  Inside the range that
    cpol_upload_model certifies the kernel shares one refined reciprocal per level pair; outside it every quotient
    is a full division.  Either way the bits are the C restatement's: variables spread over the certified range's
    40 decades, values below it (down to subnormals), non-finite values, a level one float32 step thick."""
    m = small_model(scheme=1)
    rng = np.random.default_rng(5)
    fields = {k: np.array(v, dtype='float32') for k, v in m['fields'].items()}
    zlev = np.array(m['zlev'], dtype='float32')
    want_certified = True
    if case == 'wide':
        for k, e in (('U', 24.), ('V', -22.), ('QR_v', -15.), ('T', 20.)):
            fields[k] = (fields[k] * np.float32(10.) ** rng.uniform(e - 1, e, fields[k].shape).astype('float32')).astype('float32')
        assert 0 < np.abs(fields['V'][fields['V'] != 0]).min() and np.abs(fields['V'][fields['V'] != 0]).min() > 2. ** -78
    elif case == 'tiny':
        fields['QR_v'] = (fields['QR_v'] * np.float32(10.) ** rng.uniform(-36, -18, fields['QR_v'].shape).astype('float32')).astype('float32')
        assert (np.abs(fields['QR_v'][fields['QR_v'] != 0]) < 2. ** -126).any()         # subnormals among them
        want_certified = False
    elif case == 'nonfinite':
        fields['W'][3, 7, 9] = np.inf
        fields['QS_v'][5, 11, 13] = np.nan
        want_certified = False
    elif case == 'thin_level':
        zlev[4] = np.nextafter(zlev[5], np.float32(np.inf))          # one float32 step: 2.4e-4 or 4.9e-4 m
        want_certified = False
    conf = O.make_config(microphysics={'scheme': 1})
    eng = engine.Engine(conf, device=0)
    eng.upload_model(fields, zlev, m['proj_info'], m['resolution'])
    assert eng.ctx.model_range_certified() == want_certified
    n = 20000
    coords = np.empty((n, 2), dtype='float32')
    coords[:, 0] = -0.8 + 0.04 * rng.uniform(0, 38.99, n)
    coords[:, 1] = -1.0 + 0.05 * rng.uniform(0, 46.99, n)
    H = rng.uniform(100., 9500., n).astype('float32')
    out = eng.ctx.interp_points(coords, H)
    llc = np.array([-1.0, -0.8], dtype='float32')
    res = np.array([0.05, 0.04], dtype='float32')
    with np.errstate(all='ignore'):
        for i, name in enumerate(fields.keys()):
            want = O.interp_c(coords, H, fields[name], zlev, llc, res)
            np.testing.assert_array_equal(np.isnan(out[i]), np.isnan(want), err_msg=name)
            ok = ~np.isnan(want)
            np.testing.assert_array_equal(out[i][ok].view(np.int32), want[ok].view(np.int32), err_msg=name)


@pytest.mark.parametrize('nz', [3, 4, 7, 8, 9, 15, 33, 64, 80, 113])
def test_interp_level_search_any_level_count(nz):
    """The level search starts from eight coarse heights at levels k (nz - 1) / 7 and narrows four ways at a
    time: every level count from the smallest the library accepts, points on exact level heights, points above
    the top, below the lowest level and below the terrain, bit for bit against the C restatement
    (interpolation_c.c:108-135)."""
    m = syn.synthetic_model(nz=nz, ny=24, nx=28, scheme=1, seed=1000 + nz, llc=(-1.0, -0.8), res=(0.05, 0.04), ztop=9000.)
    conf = O.make_config(microphysics={'scheme': 1})
    eng = engine.Engine(conf, device=0)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    rng = np.random.default_rng(nz)
    n = 6000
    coords = np.empty((n, 2), dtype='float32')
    coords[:, 0] = -0.8 + 0.04 * rng.uniform(0, 22.99, n)
    coords[:, 1] = -1.0 + 0.05 * rng.uniform(0, 26.99, n)
    H = rng.uniform(-200., 9800., n).astype('float32')
    # a third of the points sit exactly on a level height of their lower-left column
    zlev = np.asarray(m['zlev'], dtype='float32')
    llc = np.array([-1.0, -0.8], dtype='float32')
    res = np.array([0.05, 0.04], dtype='float32')
    iy = np.floor((coords[:, 0] - llc[1]) / res[1]).astype(int)
    ix = np.floor((coords[:, 1] - llc[0]) / res[0]).astype(int)
    lev = rng.integers(0, nz, n)
    hit = rng.uniform(size=n) < 1. / 3
    H[hit] = zlev[lev[hit], iy[hit], ix[hit]]
    out = eng.ctx.interp_points(coords, H)
    seen = 0
    for i, name in enumerate(m['fields'].keys()):
        want = O.interp_c(coords, H, m['fields'][name], zlev, llc, res)
        np.testing.assert_array_equal(np.isnan(out[i]), np.isnan(want), err_msg=name)
        ok = ~np.isnan(want)
        np.testing.assert_array_equal(out[i][ok].view(np.int32), want[ok].view(np.int32), err_msg=name)
        seen = max(seen, int((ok & (want != -9999.)).sum()))
    assert seen > n // 4


def test_get_all_radar_pts_dropin():
    """The reference's C symbol, same signature, served by the GPU."""
    import ctypes as C
    g = util.golden('interp')
    m = small_model()
    lib = _lib.load()
    n = len(g['H'])
    for name in ('T', 'QNS_v'):
        data = np.ascontiguousarray(m['fields'][name])
        out = np.zeros(n, dtype='float32')
        coords = np.ascontiguousarray(g['coords'])
        H = np.ascontiguousarray(g['H'])
        llc, res = np.ascontiguousarray(g['llc']), np.ascontiguousarray(g['res'])
        r = lib.get_all_radar_pts(_lib.fptr(out), n, _lib.fptr(coords), n, 2, _lib.fptr(H), n, _lib.fptr(data), 12, 40, 48,
                                  _lib.fptr(m['zlev']), 12, 40, 48, _lib.fptr(llc), 2, _lib.fptr(res), 2)
        assert r
        want = g['out_' + name]
        np.testing.assert_array_equal(np.isnan(out), np.isnan(want))
        ok = ~np.isnan(want)
        np.testing.assert_array_equal(out[ok].view(np.int32), want[ok].view(np.int32))


def test_get_refl_dropin():
    g = util.golden('get_refl')
    lib = _lib.load()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    Da, Db, D, rcs = f(g['Da']), f(g['Db']), f(g['D']), f(g['rcs'])
    N0, mu, lam, nu, step, dmin = f(g['N0']), f(g['mu']), f(g['lam']), f(g['nu']), f(g['step']), f(g['Dmin'])
    out = np.zeros(len(g['out']), dtype=np.float32)
    p = _lib.fptr
    r = lib.get_refl(p(out), len(out), p(Da), Da.shape[0], 3, p(Db), Db.shape[0], 3, p(D), 1024, 3, p(rcs), 1024, 3,
                     p(N0), 3, p(mu), 3, p(lam), 3, p(nu), 3, p(step), 3, p(dmin), 3)
    assert r
    np.testing.assert_allclose(out, g['out'], rtol=2e-6, atol=0)


def pipeline_conf(scheme, nh, nv, res=500., rng=20000., dscheme=1):
    return O.make_config(microphysics={'scheme': scheme}, doppler={'scheme': dscheme},
                         radar={'coords': [46.0, 8.5, 1500.], 'radial_resolution': res, 'range': rng,
                                'frequency': 5.6, '3dB_beamwidth': 1.0},
                         integration={'scheme': 1, 'nh_GH': nh, 'nv_GH': nv, 'weight_threshold': 1.})


def pipeline_model(scheme):
    # rotated pole chosen so that the radar sits near the middle of the small grid
    m = syn.synthetic_model(nz=16, ny=60, nx=64, scheme=scheme, seed=5, llc=(-1.6, -1.5), res=(0.05, 0.05),
                            pole=(-44.0, 8.5), ztop=12000.)
    return m


def oracle_model_vars(m):
    attrs = {'z-levels': m['zlev'], 'proj_info': m['proj_info'], 'resolution': m['resolution']}
    return {n: O.ModelVar(a, attrs) for n, a in m['fields'].items()}


def test_geolocate_vs_oracle():
    conf = pipeline_conf(2, 3, 3)
    m = pipeline_model(2)
    eng = engine.Engine(conf, device=0)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    az_off, el_off, w = eng.set_gauss_hermite()
    az = np.array([0., 33.3, 181.0, 359.5])
    el = np.array([1.0, 5.5, 20.0, 89.0])
    geo = eng.ctx.geolocate(az, el)
    rr = O.radar_ranges(conf)
    p = m['proj_info']
    n_bad = 0
    for r in range(len(az)):
        for b in range(len(w)):
            S, H, E = O.ref_4_3(rr, el_off[b] + el[r], conf['radar']['coords'])
            for got, want in ((geo['dist'][r, b], S), (geo['height'][r, b], H), (geo['elev'][r, b], E)):
                d = np.abs(got.view(np.int32) - want.view(np.int32))
                assert d.max() <= 1
                n_bad += int((d != 0).sum())
            lon, lat = O.geod_fwd(conf['radar']['coords'][1], conf['radar']['coords'][0], az_off[b] + az[r],
                                  geo['dist'][r, b].astype(np.float64))
            np.testing.assert_allclose(geo['lat'][r, b], lat, rtol=0, atol=1e-12)
            np.testing.assert_allclose(geo['lon'][r, b], lon, rtol=0, atol=1e-12)
            rot = O.wgs_to_rotated(geo['lat'][r, b], geo['lon'][r, b], p['Latitude_of_southern_pole'],
                                   p['Longitude_of_southern_pole']).astype('float32')
            d = np.abs(geo['coords'][r, b].view(np.int32) - rot.view(np.int32))
            assert d.max() <= 1
            n_bad += int((d != 0).sum())
    # float64 libm differences may flip a float32 rounding now and then (measured on 900 000 positions of the
    # benchmark volume: one value in 4.5 million), never more
    assert n_bad <= 2


def test_turbulence_whole_path_and_fallback(capsys):
    """Scheme 4 through cpol_run_rays.  Without the eddy dissipation rate among the model variables the engine
    says so and runs the plain spectrum, as the reference's message promises (doppler_sz.py:40-41); with it,
    EDR is interpolated along the sub-beams like every other variable and each sub-beam's spectrum is spread."""
    import copy
    conf = pipeline_conf(1, 3, 3, dscheme=4)
    conf['radar'].update({'PRI': 1000, 'FFT_length': 128})
    m = pipeline_model(1)
    az, el = np.array([10., 200.]), np.array([3.0, 15.0])
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    assert 'using doppler_scheme == 3 instead' in capsys.readouterr().out
    eng.set_gauss_hermite()
    plain = eng.ctx.run_rays(az, el)['dspectrum'].copy()
    conf3 = copy.deepcopy(conf)
    conf3['doppler']['scheme'] = 3
    eng.configure(conf3)
    np.testing.assert_array_equal(eng.ctx.run_rays(az, el)['dspectrum'], plain)
    # now with an eddy dissipation rate field
    eng.configure(conf)
    fields = dict(m['fields'])
    z = np.asarray(m['zlev'], dtype=np.float64)
    fields['EDR'] = (2e-3 * np.exp(-z / 4000.) * (1.5 + np.sin(z / 700.))).astype('float32')
    eng.upload_model(fields, m['zlev'], m['proj_info'], m['resolution'])
    out = eng.ctx.run_rays(az, el)
    assert (out['ray_status'] == 0).all()
    luts = util.oracle_luts(1, 5.6)
    attrs = {'z-levels': m['zlev'], 'proj_info': m['proj_info'], 'resolution': m['resolution']}
    dic = {n: O.ModelVar(a, attrs) for n, a in fields.items()}
    changed = 0
    for r in range(len(az)):
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(dic, az[r], el[r], conf)
            assert 'EDR' in beams[0].values
            want_v = O.get_doppler_velocity(beams, luts, conf, get_refl=O.get_refl_c)
        got, want = out['dspectrum'][r], want_v.values['DSPECTRUM']
        peak = np.abs(want).max(axis=1, keepdims=True)
        err = (np.abs(got - want) - 2e-6 * np.abs(want)) / np.maximum(peak, 1e-300)
        assert err.max() < 2e-4, "ray %d gate %d: %.3e of the peak" % (r, int(err.max(axis=1).argmax()), err.max())
        # a table term that enters from the neighbouring bin may leave or join the gate's spectrum at its edge
        np.testing.assert_allclose(got.sum(axis=1), want.sum(axis=1), rtol=2e-4)
        util.assert_close(out['rvel'][r], want_v.values['RVEL'], rtol=1e-5, atol=0.02, name='RVEL')
        changed += int((got != plain[r]).sum())
    assert changed > 100


@pytest.mark.parametrize('scheme,nh,nv', [(1, 1, 1), (2, 3, 3)])
def test_run_rays_vs_oracle(scheme, nh, nv):
    """Whole path (get_profiles_GH -> get_doppler_velocity -> get_radar_observables) for a few rays."""
    conf = pipeline_conf(scheme, nh, nv)
    m = pipeline_model(scheme)
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    az = np.array([10., 100., 200., 300., 45.])
    el = np.array([1.0, 3.0, 8.0, 15.0, 40.0])
    out = eng.ctx.run_rays(az, el)
    assert (out['ray_status'] == 0).all()
    luts = util.oracle_luts(scheme, 5.6)
    dic = oracle_model_vars(m)
    for r in range(len(az)):
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(dic, az[r], el[r], conf)
            want_v = O.get_doppler_velocity(beams, luts, conf)
            want = O.get_radar_observables(beams, luts, conf)
        check_obs(out['obs'][:, r], want.values, 'ray %d' % r, conf['radar']['radial_resolution'])
        np.testing.assert_array_equal(out['mask'][r], want.mask)
        util.assert_close(out['rvel'][r], want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')
        np.testing.assert_array_equal(out['dist'][r], want.dist_profile)
        np.testing.assert_array_equal(out['height'][r], want.heights_profile)
        np.testing.assert_allclose(out['lat'][r], want.lats_profile, rtol=0, atol=1e-12)
        np.testing.assert_allclose(out['lon'][r], want.lons_profile, rtol=0, atol=1e-12)


def test_host_delivery_schedules_and_async_calls():
    """cpol_run_rays delivers the same bits whatever its batch schedule (copies of one batch beside the kernels of the
    next), without a Doppler scheme it leaves the caller's rvel alone in every batch, and cpol_run_rays_async - three
    scans queued before the first wait, two sets of device buffers - delivers what the blocking call delivers."""
    import ctypes
    conf = pipeline_conf(2, 3, 3, dscheme=2)
    m = pipeline_model(2)
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    ctx = eng.ctx
    rng = np.random.default_rng(3)
    scans = [(rng.uniform(0, 360, n), rng.uniform(0.5, 30, n)) for n in (37, 11, 23)]
    ctx.set_host_batching(1, 0.)
    want = [ctx.run_rays(az, el) for az, el in scans]
    names = ('obs', 'rvel', 'mask', 'lat', 'lon', 'dist', 'height', 'ray_status')
    try:
        ctx.set_batch_units(5 * 9 * eng.n_gates)            # at most five rays per batch
        for nb, lf in ((4, 0.25), (3, 0.05), (8, 0.5)):
            ctx.set_host_batching(nb, lf)
            got = ctx.run_rays(*scans[0])
            for k in names:
                assert got[k].tobytes() == want[0][k].tobytes(), (nb, lf, k)
        ctx.set_batch_units(32 << 20)
        ctx.set_host_batching(2, 0.5)
        G = eng.n_gates
        bufs, structs = [], []
        for az, el in scans:
            R = len(az)
            b = {'obs': np.full((9, R, G), 7, np.float32), 'ray_status': np.full(R, -1, np.int32)}
            b.update({k: np.full((R, G), 7, np.float64) for k in ('rvel', 'mask', 'lat', 'lon')})
            b.update({k: np.full((R, G), 7, np.float32) for k in ('dist', 'height')})
            bufs.append(b)
            structs.append(_lib.CpolOutputs(_lib.fptr(b['obs']), _lib.dptr(b['rvel']), _lib.dptr(b['mask']), _lib.dptr(b['lat']),
                                            _lib.dptr(b['lon']), _lib.fptr(b['dist']), _lib.fptr(b['height']),
                                            _lib.iptr(b['ray_status']), None))
        keep = [(np.ascontiguousarray(az), np.ascontiguousarray(el)) for az, el in scans]
        for (az, el), st in zip(keep, structs):
            ctx.run_rays_async(az.ctypes.data, el.ctypes.data, len(az), st)
        ctx.synchronize()
        for b, w in zip(bufs, want):
            for k in names:
                assert b[k].tobytes() == w[k].tobytes(), k
    finally:
        ctx.set_batch_units(32 << 20)
        ctx.set_host_batching(4, 0.25)
    # no Doppler scheme: rvel is not an output, whatever the number of batches
    conf0 = pipeline_conf(2, 3, 3, dscheme=2)
    conf0['doppler']['scheme'] = 0
    eng.configure(conf0, doppler=0)
    az, el = scans[0]
    R, G = len(az), eng.n_gates
    rv = np.full((R, G), 123.0)
    o = {'obs': np.empty((9, R, G), np.float32), 'mask': np.empty((R, G)), 'st': np.zeros(R, np.int32)}
    st = _lib.CpolOutputs(_lib.fptr(o['obs']), _lib.dptr(rv), _lib.dptr(o['mask']), None, None, None, None, _lib.iptr(o['st']), None)
    ctx.set_batch_units(5 * 9 * G)
    try:
        rc = ctx.lib.cpol_run_rays(ctx.h, R, np.ascontiguousarray(az).ctypes.data, np.ascontiguousarray(el).ctypes.data, ctypes.byref(st))
        assert rc == 0
    finally:
        ctx.set_batch_units(32 << 20)
    assert (rv == 123.0).all()
    assert o['obs'][0].tobytes() == want[0]['obs'][0].tobytes()


def test_run_rays_out_of_domain():
    conf = pipeline_conf(1, 1, 1, rng=400000.)
    m = pipeline_model(1)
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    with pytest.raises(IndexError):
        eng.ctx.run_rays(np.array([90.]), np.array([1.0]))
    out = eng.ctx.run_rays(np.array([90.]), np.array([1.0]), raise_on_domain=False)
    assert out['ray_status'][0] == _lib.ERR_DOMAIN


# ---------------------------------------------------------------------------------------------
# the reference-named Python API on top of the C ABI
# ---------------------------------------------------------------------------------------------
def _api_setup(scheme, nh, nv, dscheme=1, sens=None):
    from cosmo_pol_amd import cfg as pcfg, model as pmodel, runtime
    from cosmo_pol_amd.constants import constants
    conf = pipeline_conf(scheme, nh, nv, dscheme=dscheme)
    if sens is not None:
        conf['radar']['sensitivity'] = sens
    runtime.reset()
    pcfg.set_config(conf)
    constants.update()
    m = pipeline_model(scheme)
    dic = pmodel.variables_from_arrays(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    tabs = util.synthetic_lut_arrays(scheme, 5.6)
    luts = {h: Lookup_table.from_arrays(*t) for h, t in tabs.items()}
    return conf, m, dic, luts


def test_api_functions_vs_oracle():
    """interpolation.get_profiles_GH -> doppler_sz.get_doppler_velocity -> scattering_sz.get_radar_observables,
    called like radar_operator.py:241-243 does."""
    from cosmo_pol_amd import interpolation, scattering_sz, doppler_sz
    conf, m, dic, luts = _api_setup(2, 3, 3)
    az, el = 123.0, 4.0
    beams = interpolation.get_profiles_GH(dic, az, el)
    with np.errstate(all='ignore'):
        want_b = O.get_profiles_GH(oracle_model_vars(m), az, el, conf)
    assert len(beams) == len(want_b) == 9
    for b, w in zip(beams, want_b):
        np.testing.assert_array_equal(b.mask, w.mask)
        np.testing.assert_array_equal(b.dist_profile, w.dist_profile)
        np.testing.assert_array_equal(b.heights_profile, w.heights_profile)
        assert b.GH_weight == w.GH_weight and list(b.GH_pt) == list(w.GH_pt)
        for k in w.values:
            np.testing.assert_array_equal(b.values[k].view(np.int32), w.values[k].view(np.int32))
    dop = doppler_sz.get_doppler_velocity(beams, luts)
    pol = scattering_sz.get_radar_observables(beams, luts)
    olut = util.oracle_luts(2, 5.6)
    with np.errstate(all='ignore'):
        want_v = O.get_doppler_velocity(want_b, olut, conf)
        want = O.get_radar_observables(want_b, olut, conf)
    check_obs(np.stack([pol.values[n] for n in OBS]), want.values, 'api', conf['radar']['radial_resolution'])
    util.assert_close(dop.values['RVEL'], want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')
    np.testing.assert_array_equal(pol.mask, want.mask)
    # the in-place folding of the reference is reproduced
    np.testing.assert_array_equal(beams[0].elev_profile, want_b[0].elev_profile)


def test_radar_operator_ppi_with_sensitivity_cut():
    from cosmo_pol_amd import RadarOperator
    conf, m, dic, luts = _api_setup(1, 1, 1, sens=[10, 10000])
    op = RadarOperator(conf)
    op.load_model_file(dic)
    assert op.get_config()['microphysics']['scheme'] == 1
    op.set_lut(1, luts)
    radar = op.get_PPI([2.0, 6.0], az_step=90, az_start=0, az_stop=270)
    assert radar.nrays == 8 and radar.fields['ZH']['data'].shape == (8, 40)
    olut = util.oracle_luts(1, 5.6)
    cut = 0
    for s, el in enumerate([2.0, 6.0]):
        zh = radar.get_field(s, 'ZH')
        for i, az in enumerate([0., 90., 180., 270.]):
            with np.errstate(all='ignore'):
                beams = O.get_profiles_GH(oracle_model_vars(m), az, el, conf)
                v = O.get_doppler_velocity(beams, olut, conf)
                b = O.get_radar_observables(beams, olut, conf)
                b.values.update(v.values)
                pre = np.isfinite(b.values['ZH']).sum()
                O.cut_at_sensitivity([b], conf)
                cut += pre - np.isfinite(b.values['ZH']).sum()
                want_db = 10 * np.log10(b.values['ZH'])
            got = zh[i].filled(np.nan)
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want_db))
            ok = ~np.isnan(want_db)
            assert np.abs(got[ok] - want_db[ok]).max() < 0.01
            rv = radar.get_field(s, 'RVEL')[i].filled(np.nan)
            util.assert_close(rv, b.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')
    assert cut > 0          # the cut was exercised


def test_radar_operator_three_element_sensitivity():
    """radar.sensitivity = [sensitivity (dBm), gain (dBm), SNR (dB)]: the cut follows the radar equation
    (utilities.py:74-76 with constants.RADAR_CONSTANT_DB, constants.py:79-82)."""
    from cosmo_pol_amd import RadarOperator
    from cosmo_pol_amd.constants import constants
    sens = [-85., 42., 8.]
    conf, m, dic, luts = _api_setup(1, 1, 1, sens=sens)
    op = RadarOperator(conf)
    op.load_model_file(dic)
    op.set_lut(1, luts)
    radar = op.get_PPI([3.0], azimuths=[20., 140., 260.])
    assert np.isfinite(constants.RADAR_CONSTANT_DB)
    olut = util.oracle_luts(1, 5.6)
    cut = kept = 0
    for i, az in enumerate([20., 140., 260.]):
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(oracle_model_vars(m), az, 3.0, conf)
            b = O.get_radar_observables(beams, olut, conf)
            pre = np.isfinite(b.values['ZH']).sum()
            O.cut_at_sensitivity([b], conf)
            want_db = 10 * np.log10(b.values['ZH'])
        cut += pre - np.isfinite(want_db).sum()
        kept += np.isfinite(want_db).sum()
        got = radar.get_field(0, 'ZH')[i].filled(np.nan)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want_db))
        ok = ~np.isnan(want_db)
        assert np.abs(got[ok] - want_db[ok]).max() < 0.01
    assert cut > 0 and kept > 0          # the threshold bites somewhere along the rays, not everywhere


def test_radar_operator_diagnostic_mode():
    """diagnostic_mode=True (radar_operator.py:245-246): the sub-beam averages of the interpolated model variables
    (interpolation.integrate_GH_pts, interpolation.py:21-51) join the output fields."""
    from cosmo_pol_amd import RadarOperator
    conf, m, dic, luts = _api_setup(2, 3, 3)
    op = RadarOperator(conf, diagnostic_mode=True)
    op.load_model_file(dic)
    op.set_lut(2, luts)
    els, azs = [1.5, 12.0], [15., 200., 310.]
    radar = op.get_PPI(els, azimuths=azs)
    plain = RadarOperator(conf)
    plain.load_model_file(dic)
    plain.set_lut(2, luts)
    assert 'T' not in plain.get_PPI(els, azimuths=azs).fields
    seen = 0
    for s, el in enumerate(els):
        for i, az in enumerate(azs):
            with np.errstate(all='ignore'):
                want = O.integrate_GH_pts(O.get_profiles_GH(oracle_model_vars(m), az, el, conf))
            for k in ('T', 'U', 'QR_v', 'QNS_v', 'RHO'):
                got = radar.get_field(s, k)[i].filled(np.nan)
                util.assert_close(got, np.asarray(want.values[k], dtype=np.float64), rtol=1e-6, atol=1e-30, name=k)
                seen += int((got != 0).sum())
    assert seen > 100


def test_radar_operator_rhi_doppler_spectrum():
    """get_RHI with doppler.scheme 3: RVEL and DSPECTRUM per ray, rows removed by the sensitivity cut are NaN."""
    from cosmo_pol_amd import RadarOperator
    conf, m, dic, luts = _api_setup(1, 1, 3, dscheme=3, sens=[5, 10000])
    op = RadarOperator(conf)
    op.load_model_file(dic)
    op.set_lut(1, luts)
    els = [3.0, 12.0, 35.0]
    radar = op.get_RHI(77.0, elevations=els)
    spec = radar.fields['DSPECTRUM']['data']
    assert spec.shape[:2] == (3, 40)
    olut = util.oracle_luts(1, 5.6)
    cut = 0
    for i, el in enumerate(els):
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(oracle_model_vars(m), 77.0, el, conf)
            v = O.get_doppler_velocity(beams, olut, conf, get_refl=O.get_refl_c)
            b = O.get_radar_observables(beams, olut, conf)
            b.values.update(v.values)
            pre = np.isfinite(b.values['ZH']).sum()
            O.cut_at_sensitivity([b], conf)
            cut += pre - np.isfinite(b.values['ZH']).sum()
        want = b.values['DSPECTRUM']
        got = spec[i].filled(np.nan)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
        ok = ~np.isnan(want).any(axis=1)
        rv_atol = np.zeros(len(ok))
        rv_atol[ok] = check_spectrum(got[ok], want[ok], O.radar_constants(conf)['VARRAY'],
                                     O.spectrum_open_bins(beams, conf)[ok])
        rv = radar.get_field(0, 'RVEL')[i].filled(np.nan)
        check_rvel(rv, b.values['RVEL'], rv_atol)
    assert cut > 0


# ---------------------------------------------------------------------------------------------
# the five reference configurations, shaped like their option files, on a small model
# ---------------------------------------------------------------------------------------------
REFERENCE_CONFIGS = {
    # option_files/test_strat.yml:1-25: C-band PPI, 50 km at 75 m (667 gates), 1-moment, one sub-beam
    'test_strat': dict(kind='ppi', mp=1, freq=5.6, rng=50000., res=75., nh=1, nv=1, dop=1, bw=1.45,
                       rays=[(1.0, 30.), (1.0, 211.)]),
    # option_files/MXPOL_RHI.yml: RHI, 50 km at 75 m, 1-moment, 3 x 3 Gauss-Hermite
    'MXPOL_RHI': dict(kind='rhi', mp=1, freq=5.6, rng=50000., res=75., nh=3, nv=3, dop=1, bw=1.45,
                      rays=[(2.0, 47.), (35.0, 47.)]),
    # option_files/MXPOL_RHI_new_scheme.yml:5-6 (500 m -> 100 gates), 2-moment, nh = 3, nv = 9 (cfg.py:39-40)
    'MXPOL_RHI_new_scheme': dict(kind='rhi', mp=2, freq=5.6, rng=50000., res=500., nh=3, nv=9, dop=1, bw=1.45,
                                 rays=[(5.0, 140.), (60.0, 140.)]),
    # option_files/CH_PPI_lema.yml: 150 km at 500 m (300 gates), 9 x 9 sub-beams, 2-moment input
    'CH_PPI_lema': dict(kind='ppi', mp=2, freq=5.6, rng=150000., res=500., nh=9, nv=9, dop=2, bw=1.0,
                        rays=[(1.6, 300.)]),
    # option_files/MXPOL_RHI_PAYERNE_DOPPLER.yml:4-23: X band, 11 km at 75 m (147 gates), FFT 256, full spectrum
    'MXPOL_RHI_PAYERNE_DOPPLER': dict(kind='rhi', mp=1, freq=9.41, rng=11000., res=75., nh=3, nv=3, dop=3, bw=1.45,
                                      rays=[(10.0, 90.), (45.0, 90.)]),
}


@pytest.mark.parametrize('name', sorted(REFERENCE_CONFIGS))
def test_reference_configuration(name):
    """RadarOperator.get_PPI / get_RHI under each of the reference's option files (settings restated), a few
    rays each, every field against the CPU restatement of radar_operator.py:241-244."""
    from cosmo_pol_amd import RadarOperator, cfg as pcfg, model as pmodel, runtime
    from cosmo_pol_amd.constants import constants
    c = REFERENCE_CONFIGS[name]
    conf = O.make_config(microphysics={'scheme': c['mp']}, doppler={'scheme': c['dop']},
                         radar={'coords': [46.0, 8.5, 1500.], 'radial_resolution': c['res'], 'range': c['rng'],
                                'frequency': c['freq'], '3dB_beamwidth': c['bw'], 'PRI': 700, 'FFT_length': 256,
                                'sensitivity': [-30, 10000]},
                         integration={'scheme': 1, 'nh_GH': c['nh'], 'nv_GH': c['nv'], 'weight_threshold': 1.})
    runtime.reset()
    pcfg.set_config(conf)
    constants.update()
    m = syn.synthetic_model(nz=20, ny=90, nx=90, scheme=c['mp'], seed=11, llc=(-2.25, -2.25), res=(0.05, 0.05),
                            pole=(-44.0, 8.5), ztop=15000.)
    dic = pmodel.variables_from_arrays(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    tabs = util.synthetic_lut_arrays(c['mp'], c['freq'])
    op = RadarOperator(conf)
    op.load_model_file(dic)
    op.set_lut(c['mp'], {h: Lookup_table.from_arrays(*t) for h, t in tabs.items()})
    olut = util.oracle_luts(c['mp'], c['freq'])
    for el, az in c['rays']:
        radar = op.get_PPI([el], azimuths=[az]) if c['kind'] == 'ppi' else op.get_RHI(az, elevations=[el])
        n_gates = int(round(c['rng'] / c['res']))
        assert radar.fields['ZH']['data'].shape == (1, n_gates)
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(oracle_model_vars(m), az, el, conf)
            assert len(beams) == c['nh'] * c['nv']
            get_refl = O.get_refl_c if c['dop'] == 3 else O.get_refl_np
            want = O.get_radar_observables(beams, olut, conf)
            want.values.update(O.get_doppler_velocity(beams, olut, conf, get_refl=get_refl).values)
            kdp_before_cut = want.values['KDP'].copy()
            O.cut_at_sensitivity([want], conf)
        got = {}
        for n in OBS:
            if n in radar.fields:
                a = radar.fields[n]['data'][0].filled(np.nan).astype(np.float64)
                got[n] = 10 ** (0.1 * a) if n in ('ZH', 'ZV', 'ZDR') else a          # back from dB
        for n in ('ZH', 'ZV', 'ZDR'):
            w = want.values[n]
            ok = ~np.isnan(w)
            np.testing.assert_array_equal(np.isnan(got[n]), ~ok, err_msg=n)
            assert np.abs(10 * np.log10(got[n][ok] / w[ok])).max() < 0.01, n
        assert np.nanmax(np.abs(got['KDP'] - want.values['KDP']), initial=0.) < 1e-3
        for n in ('KDP', 'RHOHV', 'AH', 'AV', 'DELTA_HV'):
            util.assert_close(got[n], want.values[n], rtol=1e-5,
                              atol=1e-5 * np.nanmax(np.abs(want.values['KDP']), initial=0.) if n != 'RHOHV' else 0., name=n)
        # PHIDP: a gate removed by the sensitivity cut holds NaN on both sides, but its KDP fed the sums of the gates
        # behind it on both sides before it was removed
        np.testing.assert_array_equal(np.isnan(got['PHIDP']), np.isnan(want.values['PHIDP']))
        cut = np.isnan(want.values['KDP']) & ~np.isnan(kdp_before_cut)
        check_phidp(got['PHIDP'], got['KDP'], want.values['PHIDP'], kdp_before_cut, c['res'], name, unseen=cut)
        rv_atol = np.zeros(n_gates)
        if c['dop'] == 3:
            sp = radar.fields['DSPECTRUM']['data'][0].filled(np.nan)
            ws = want.values['DSPECTRUM']
            np.testing.assert_array_equal(np.isnan(sp), np.isnan(ws))
            okr = ~np.isnan(ws).any(axis=1)
            rv_atol[okr] = check_spectrum(sp[okr], ws[okr], O.radar_constants(conf)['VARRAY'],
                                          O.spectrum_open_bins(beams, conf)[okr])
        rv = radar.fields['RVEL']['data'][0].filled(np.nan)
        check_rvel(rv, want.values['RVEL'], rv_atol)


@pytest.mark.parametrize('name', ['gh_1x1', 'gh_3x3', 'gh_3x9_thr', 'multigauss', 'legendre_antenna', 'gautschi'])
def test_profiles_vs_reference_run(name, tmp_path):
    """interpolation.get_profiles_GH against the beams the reference's own get_profiles_GH assembled (golden
    'profiles': integration schemes 1, 2, 3, 5; geodesy served by stand-ins there, so positions are not pinned by
    it, but the sub-beam set, refraction, masks and the interpolated values along the beams are)."""
    from cosmo_pol_amd import interpolation, cfg as pcfg, model as pmodel, runtime
    from cosmo_pol_amd.constants import constants
    from tests.util import PROFILE_CASES
    g = util.golden('profiles')
    conf = O.make_config(radar={'coords': [46.0, 8.5, 1500.], 'radial_resolution': 500., 'range': 20000.,
                                'frequency': 5.6, '3dB_beamwidth': 1.45},
                         integration=dict(PROFILE_CASES[name]['integration']))
    if conf['integration']['scheme'] in (3, 5):
        path = str(tmp_path / 'antenna.csv')
        np.savetxt(path, np.column_stack((g['antenna_angles'], g['antenna_db'])), delimiter=',')
        conf['integration']['antenna_diagram'] = path
    runtime.reset()
    pcfg.set_config(conf)
    constants.update()
    m = pipeline_model(1)
    dic = pmodel.variables_from_arrays(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    beams = interpolation.get_profiles_GH(dic, float(g['azimuth']), float(g['elevation']))
    assert len(beams) == len(g[name + '_GH_weight'])
    n_ulp = n_val = 0
    for b, bm in enumerate(beams):
        np.testing.assert_allclose(bm.GH_pt, g[name + '_GH_pt'][b], rtol=0, atol=1e-9)
        np.testing.assert_allclose(bm.GH_weight, g[name + '_GH_weight'][b], rtol=1e-9)
        np.testing.assert_array_equal(bm.mask, g[name + '_mask'][b])
        for got, want in ((bm.dist_profile, g[name + '_dist'][b]), (bm.heights_profile, g[name + '_heights'][b]),
                          (bm.elev_profile, g[name + '_elev'][b])):
            d = np.abs(got.view(np.int32) - want.view(np.int32))
            assert d.max() <= 1
            n_ulp += int((d != 0).sum())
        for v in ('T', 'QR_v', 'U'):
            # the interpolation itself is bit-exact (test_interp_points_bit_exact); what can move a value here is the
            # last bit of a position (counted above), and then only by the field's change over that ulp
            gv, wv = bm.values[v], g['%s_val_%s' % (name, v)][b]
            util.assert_close(gv, wv, rtol=2e-5, atol=1e-12, name=v)
            n_val += int((gv.view(np.int32) != np.asarray(wv, dtype=np.float32).view(np.int32)).sum())
    n_pos = len(beams) * len(beams[0].dist_profile)
    assert n_ulp <= 0.002 * 3 * n_pos + 1
    assert n_val <= 0.002 * 3 * n_pos + 1, "%d of %d interpolated values differ in their bits" % (n_val, 3 * n_pos)


def _refractivity_field(m):
    z = m['zlev'].astype(np.float64)
    agl = z - z[-1][None]
    return (320. * np.exp(-z / 7500.) + 60. * np.exp(-agl / 150.)).astype('float32')


def test_refraction_scheme_2_table():
    """refraction.scheme 2 (atm_refraction.py:45-84): profiles solved on the host (pinned against the reference by
    the 'refraction_ode' golden), looked up on the GPU by total elevation; everything downstream against the
    oracle fed the same profiles through the reference's list_refraction parameter (interpolation.py:247-258)."""
    from cosmo_pol_amd import RadarOperator, refraction, cfg as pcfg, model as pmodel, runtime
    from cosmo_pol_amd.constants import constants
    conf = pipeline_conf(1, 3, 3)
    conf['refraction'] = {'scheme': 2}
    runtime.reset()
    pcfg.set_config(conf)
    constants.update()
    m = pipeline_model(1)
    fields = dict(m['fields'])
    fields['N'] = _refractivity_field(m)
    dic = pmodel.variables_from_arrays(fields, m['zlev'], m['proj_info'], m['resolution'])
    op = RadarOperator(conf)
    op.load_model_file(dic)
    assert 'N' not in op.dic_vars and op.N != 0
    op.set_lut(1, {h: Lookup_table.from_arrays(*t) for h, t in util.synthetic_lut_arrays(1, 5.6).items()})
    els = [0.5, 4.0, 25.0]
    radar = op.get_RHI(200.0, elevations=els)
    olut = util.oracle_luts(1, 5.6)
    rr = O.radar_ranges(conf)
    subs, pts_ver = O.gh_quadrature(conf)
    for i, el in enumerate(els):
        refr = [refraction.ref_ODE_s(rr, pt + el, conf['radar']['coords'], op.N) for pt in pts_ver]
        # the table differs from the 4/3 model, or the test would prove nothing
        assert np.abs(refr[1][1] - O.ref_4_3(rr, pts_ver[1] + el, conf['radar']['coords'])[1]).max() > 1.0
        with np.errstate(all='ignore'):
            beams = O.get_profiles_GH(oracle_model_vars(m), 200.0, el, conf, list_refraction=refr)
            want = O.get_radar_observables(beams, olut, conf)
            want.values.update(O.get_doppler_velocity(beams, olut, conf).values)
            O.cut_at_sensitivity([want], conf)
        np.testing.assert_array_equal(radar.gate_altitude['data'][i], want.heights_profile)
        zh = radar.fields['ZH']['data'][i].filled(np.nan)
        wdb = 10 * np.log10(want.values['ZH'])
        np.testing.assert_array_equal(np.isnan(zh), np.isnan(wdb))
        ok = ~np.isnan(wdb)
        assert np.abs(zh[ok] - wdb[ok]).max() < 0.01
        util.assert_close(radar.fields['KDP']['data'][i].filled(np.nan), want.values['KDP'], rtol=1e-5,
                          atol=1e-5 * np.nanmax(np.abs(want.values['KDP']), initial=0.), name='KDP')
        util.assert_close(radar.fields['RVEL']['data'][i].filled(np.nan), want.values['RVEL'], rtol=1e-5, atol=1e-6,
                          name='RVEL')
    # a ray whose elevation has no profile is reported, not silently mis-refracted
    st = runtime.get_engine(0, doppler=1, sensitivity_cut=True)
    with pytest.raises(_lib.CosmoPolError):
        st.eng.ctx.run_rays(np.array([200.0]), np.array([7.77]))
    st.eng.set_refraction(None)
    out = st.eng.ctx.run_rays(np.array([200.0]), np.array([7.77]))
    assert (out['ray_status'] == 0).all()


# ---------------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------------
def test_empty_scan():
    conf = pipeline_conf(1, 1, 1)
    m = pipeline_model(1)
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    eng.set_gauss_hermite()
    out = eng.ctx.run_rays(np.zeros(0), np.zeros(0))
    assert out['obs'].shape == (9, 0, 40) and out['rvel'].shape == (0, 40)


def test_out_of_table_temperatures_and_folded_elevations():
    """T beyond both ends of the table axis clamps to the end nodes (lut.py:240-241), elevations above 90 and
    below 0 fold back (scattering_sz.py:66-68); the unfolded angle still projects the wind (doppler_sz.py:76)."""
    conf = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': 2}, radar={'radial_resolution': 500., 'frequency': 5.6})
    eng = make_engine(conf)
    beams = syn.synthetic_beams(64, 3, 1, seed=41, res=500.)
    rng = np.random.default_rng(2)
    for b in beams:
        b['values']['T'][:] = rng.choice([150., 199.9, 230., 262., 275.5, 315., 350.], 64).astype('float32')
        b['elev'][:] = rng.choice([-3.5, -0.2, 0., 1., 45., 89.9, 90., 90.1, 135., 179.], 64).astype('float32')
        for k in ('QR_v', 'QS_v', 'QG_v'):
            b['values'][k][:] = np.abs(b['values'][k]) + np.float32(1e-5)
    obs, mask, rvel = eng.ctx.observables_from_beams(*beams_to_arrays(beams))
    luts = util.oracle_luts(1, 5.6)
    with np.errstate(all='ignore'):
        want_v = O.get_doppler_velocity(util.oracle_beams(beams), luts, conf)
        want = O.get_radar_observables(util.oracle_beams(beams), luts, conf)
    check_obs(obs, want.values, 'clamp/fold', 500.)
    util.assert_close(rvel, want_v.values['RVEL'], rtol=1e-5, atol=1e-6, name='RVEL')


def test_longer_subbeams_are_truncated_to_the_central_one():
    """scattering_sz.py:54-57 (GPM: sub-beams longer than the central beam) through the Python API."""
    from cosmo_pol_amd import scattering_sz, doppler_sz, cfg as pcfg, runtime
    from cosmo_pol_amd.beam import Beam
    from cosmo_pol_amd.constants import constants
    conf = O.make_config(microphysics={'scheme': 1}, doppler={'scheme': 1}, radar={'radial_resolution': 500., 'frequency': 5.6})
    runtime.reset()
    pcfg.set_config(conf)
    constants.update()
    raw = syn.synthetic_beams(50, 3, 1, seed=8, res=500.)
    lens = [50, 44, 47]          # the central beam (index 1) is the shortest

    def cut(b, n):
        return dict(values={k: v[:n].copy() for k, v in b['values'].items()}, mask=b['mask'][:n].copy(),
                    elev=b['elev'][:n].copy(), dist=b['dist'][:n].copy(), heights=b['heights'][:n].copy(),
                    GH_pt=b['GH_pt'], GH_weight=b['GH_weight'])
    beams = [cut(b, n) for b, n in zip(raw, lens)]
    luts = {h: Lookup_table.from_arrays(*t) for h, t in util.synthetic_lut_arrays(1, 5.6).items()}
    mine = [Beam(dict(b['values']), b['mask'].copy(), None, None, b['dist'], b['heights'], b['elev'].copy(),
                 list(b['GH_pt']), b['GH_weight']) for b in cut_all(beams)]
    pol = scattering_sz.get_radar_observables(mine, luts)
    with np.errstate(all='ignore'):
        want = O.get_radar_observables(util.oracle_beams(cut_all(beams)), util.oracle_luts(1, 5.6), conf)
    assert len(pol.values['ZH']) == 44
    check_obs(np.stack([pol.values[n] for n in OBS]), want.values, 'ragged', 500.)
    np.testing.assert_array_equal(pol.mask, want.mask)
    assert all(len(v) == 44 for b in mine for v in b.values.values())        # visible truncation of the inputs


def cut_all(beams):
    import copy
    return copy.deepcopy(beams)


def test_gpm_swath_vs_oracle():
    """A small GPM swath (2 scan lines x 5 beams, 3 x 3 sub-beams whose profiles differ in length) through
    gpm.simulate_swath - one call of the ray kernels per scan line with the straight-line profiles uploaded as a
    refraction table - against the reference's per-beam sequence restated on the CPU (radar_operator.py:379-421:
    get_profiles_GH with get_GPM_refraction, get_radar_observables, cut_at_sensitivity), then the swath
    container (GPM_simulator.py:31-88)."""
    import copy
    from cosmo_pol_amd import gpm
    from tests.test_host_cpu import synthetic_dpr
    d, _ = synthetic_dpr(n_lines=2, n_beams=5, lat0=46.0, lon0=8.5, half_swath=60e3)
    az, elev, rang, coords = gpm.get_GPM_angles(d, 'Ku')
    conf = pipeline_conf(1, 3, 3, res=125.)
    conf['radar'].update({'type': 'GPM', 'frequency': 13.6, 'sensitivity': 12, '3dB_beamwidth': 0.7})
    m = pipeline_model(1)
    eng = make_engine(conf)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    beams = gpm.simulate_swath(eng, conf, az, elev, rang, coords, 'Ku')
    luts = util.oracle_luts(1, 13.6)
    dic = oracle_model_vars(m)
    want = []
    lens = set()
    for i in range(az.shape[0]):
        for j in range(az.shape[1]):
            c = copy.deepcopy(conf)
            c['radar']['range'] = float(rang[i, j])
            c['radar']['coords'] = [float(v) for v in coords[i]]
            with np.errstate(all='ignore'):
                sub = O.get_profiles_GH(dic, az[i, j], elev[i, j], c)
                lens.update(len(b.dist_profile) for b in sub)
                want.append(O.get_radar_observables(sub, luts, c))
    assert len(lens) > 3                                            # sub-beams of unequal length are exercised
    with np.errstate(all='ignore'):
        want = O.cut_at_sensitivity(want, conf)
    n_finite = 0
    for idx, (g, w) in enumerate(zip(beams, want)):
        assert len(g.dist_profile) == len(w.dist_profile), idx
        obs = np.stack([g.values[n] for n in OBS])
        check_obs(obs, w.values, 'beam %d' % idx, 125., exact_scan=False)
        np.testing.assert_array_equal(g.mask, w.mask)
        np.testing.assert_array_equal(g.dist_profile, w.dist_profile)
        np.testing.assert_array_equal(g.heights_profile, w.heights_profile)
        np.testing.assert_allclose(g.lats_profile, w.lats_profile, rtol=0, atol=1e-12)
        np.testing.assert_allclose(g.lons_profile, w.lons_profile, rtol=0, atol=1e-12)
        n_finite += int(np.isfinite(w.values['ZH']).sum())
    assert n_finite > 200
    sw = gpm.SimulatedGPM(beams, az.shape, 'Ku')
    bs, lats, lons, data = O.simulated_gpm(want, az.shape, OBS)
    np.testing.assert_array_equal(sw.bin_surface, bs)
    np.testing.assert_allclose(sw.lats, lats, rtol=0, atol=1e-12)
    for n in ('ZH', 'KDP', 'AH'):
        a, b = sw.data[n], data[n]
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(b)
        np.testing.assert_allclose(a[ok], b[ok], rtol=2.5e-3 if n == 'ZH' else 1e-5, atol=1e-3 if n == 'KDP' else 0)


def test_radar_operator_gpm_swath():
    """RadarOperator.get_GPM_swath (radar_operator.py:350-424) sets the DPR band's frequency, resolution and
    sensitivity itself and returns the swath container; its content equals a direct gpm.simulate_swath run."""
    from cosmo_pol_amd import RadarOperator, gpm, cfg as pcfg
    from tests.test_host_cpu import synthetic_dpr
    conf, m, dic, _ = _api_setup(1, 3, 3)
    tabs = util.synthetic_lut_arrays(1, 13.6)
    luts = {h: Lookup_table.from_arrays(*t) for h, t in tabs.items()}
    d, _ = synthetic_dpr(n_lines=1, n_beams=4, lat0=46.0, lon0=8.5, half_swath=50e3)
    op = RadarOperator(conf)
    op.load_model_file(dic)
    op.set_lut(1, luts)
    sw = op.get_GPM_swath(d, 'Ku')
    c = pcfg.CONFIG
    assert c['radar']['type'] == 'GPM' and c['radar']['frequency'] == 13.6 and c['radar']['radial_resolution'] == 125
    assert sw.data['ZH'].shape[:2] == (1, 4) and sw.band == 'Ku'
    az, elev, rang, coords = gpm.get_GPM_angles(d, 'Ku')
    eng = make_engine(c)
    eng.upload_luts(luts)
    eng.upload_model(m['fields'], m['zlev'], m['proj_info'], m['resolution'])
    ref = gpm.SimulatedGPM(gpm.simulate_swath(eng, c, az, elev, rang, coords, 'Ku'), az.shape, 'Ku')
    np.testing.assert_array_equal(sw.bin_surface, ref.bin_surface)
    for n in OBS:
        np.testing.assert_array_equal(sw.data[n], ref.data[n])
    assert np.isfinite(sw.data['ZH']).any() and (sw.bin_surface >= 0).all()
    # a ground-radar scan afterwards is refused as in the reference (radar_operator.py:159-162)
    assert op.get_PPI([2.0], az_step=90) == []
