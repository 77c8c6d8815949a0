"""Shared helpers for the tests: golden loading, oracle-side LUT / beam builders."""
import copy
import os

import numpy as np

from tests import synthetic as syn
from oracle import cosmo_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LUT_NE, LUT_NT = 8, 12     # must match tests/golden/make_golden.py


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


_lut_cache = {}


def synthetic_lut_arrays(scheme, frequency, n_e=LUT_NE, n_t=LUT_NT):
    key = (scheme, frequency, n_e, n_t)
    if key not in _lut_cache:
        _lut_cache[key] = {h: syn.synthetic_lut(h, frequency, n_e=n_e, n_t=n_t) for h in syn.hydrom_types(scheme)}
    return _lut_cache[key]


def oracle_luts(scheme, frequency, n_e=LUT_NE, n_t=LUT_NT):
    return {h: O.Lut(*v) for h, v in synthetic_lut_arrays(scheme, frequency, n_e, n_t).items()}


def beams_from_golden(g, prefix="in_"):
    names = [k[len(prefix) + 4:] for k in g.files if k.startswith(prefix + "val_")]
    nb = g[prefix + "mask"].shape[0]
    out = []
    for b in range(nb):
        out.append({"values": {n: g[prefix + "val_" + n][b].copy() for n in names},
                    "mask": g[prefix + "mask"][b].copy(), "elev": g[prefix + "elev"][b].copy(),
                    "dist": g[prefix + "dist"][b].copy(), "heights": g[prefix + "heights"][b].copy(),
                    "GH_pt": list(g[prefix + "GH_pt"][b]), "GH_weight": float(g[prefix + "GH_weight"][b])})
    return out


def oracle_beams(beams):
    out = []
    for b in beams:
        lat = np.linspace(46, 46.5, len(b["dist"]))
        out.append(O.Beam(copy.deepcopy(b["values"]), b["mask"].copy(), lat, lat + 1, b["dist"].copy(),
                          b["heights"].copy(), b["elev"].copy(), list(b["GH_pt"]), b["GH_weight"]))
    return out


def cfg_from_meta(meta, **extra):
    scheme, n_gates, n_sub, att, freq, res, seed = meta
    cfg = O.make_config(microphysics={"scheme": int(scheme)}, attenuation={"correction": int(att)},
                        radar={"frequency": float(freq), "radial_resolution": float(res)})
    for sec, kv in extra.items():
        cfg[sec].update(kv)
    return cfg


def assert_close(a, b, rtol, atol=0.0, name=""):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    assert np.array_equal(nan_a, nan_b), "%s: NaN pattern differs at %s" % (name, np.where(nan_a != nan_b)[0][:10])
    m = ~nan_a
    inf_ok = np.isinf(a[m]) == np.isinf(b[m])
    assert inf_ok.all(), name
    fin = m & np.isfinite(a) & np.isfinite(b)
    err = np.abs(a[fin] - b[fin]) - (atol + rtol * np.abs(b[fin]))
    if err.size and err.max() > 0:
        i = np.argmax(err)
        raise AssertionError("%s: max violation at %d: got %r want %r (rtol %g atol %g)" %
                             (name, i, a[fin][i], b[fin][i], rtol, atol))


# integration settings of the 'profiles' golden (tests/golden/make_golden.py: gen_profiles), shared by the generator
# and the GPU test that reads the fixture
PROFILE_CASES = {
    'gh_1x1': dict(integration={'scheme': 1, 'nh_GH': 1, 'nv_GH': 1, 'weight_threshold': 1.}),
    'gh_3x3': dict(integration={'scheme': 1, 'nh_GH': 3, 'nv_GH': 3, 'weight_threshold': 1.}),
    'gh_3x9_thr': dict(integration={'scheme': 1, 'nh_GH': 3, 'nv_GH': 9, 'weight_threshold': 0.999}),
    'multigauss': dict(integration={'scheme': 2, 'nr_GH': 3, 'na_GL': 5, 'weight_threshold': 1.,
                                    'antenna_params': np.array([[0., 0., 0.62], [-26., 2.8, 0.4], [-26., -2.8, 0.4]])}),
    'legendre_antenna': dict(integration={'scheme': 3, 'nh_GH': 5, 'nv_GH': 3, 'weight_threshold': 1.}),
    'gautschi': dict(integration={'scheme': 5, 'nh_GH': 3, 'nv_GH': 5, 'weight_threshold': 0.9999}),
}
