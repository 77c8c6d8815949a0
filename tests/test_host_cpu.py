"""CPU: host logic of the product package and the C-ABI surface (no compute calls)."""
import os
import re

import numpy as np

from cosmo_pol_amd import _lib, cfg, engine, hydrometeors
from oracle import cosmo_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


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
