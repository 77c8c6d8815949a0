"""Physical, radar-derived and microphysical constants of the forward operator.

Values follow constants/constants.py:20-98, constants/constants_1mom.py:15-65 and
constants/constants_2mom.py:16-149 of the reference (including the overrides of the
2-moment snow / graupel mass-diameter law by the 1-moment one, constants_2mom.py:27,39,61,74).
They are evaluated with the same expressions so that the float64 values are identical.
"""
import numpy as np
import scipy.special as spe

from . import cfg

EPS = np.finfo(np.float32).eps
C = 299792458.
RHO_W = 1000. / (1000 ** 3)
RHO_I = 916. / (1000 ** 3)
KW = 0.93
A = 1.6                    # Kolmogorov constant of the turbulence spread (constants.py:40)
RHO_0 = 1.225
MAX_HEIGHT_COSMO = 35000
# GPM dual-frequency precipitation radar (constants.py:91-95)
GPM_SENSITIVITY = 12
GPM_RADIAL_RES_KA = 250
GPM_RADIAL_RES_KU = 125
GPM_NO_BINS_KA = 88 + 1
GPM_NO_BINS_KU = 176 + 1
SIMULATED_VARIABLES = ['ZH', 'DSPECTRUM', 'RVEL', 'ZV', 'PHIDP', 'ZDR', 'RHOHV', 'KDP']


class Constant_class(object):
    """Radar-dependent constants; call update() after the configuration changes
    (constants/constants.py:70-98)."""

    def __init__(self):
        self.EPS, self.C, self.KW, self.RHO_W, self.RHO_I, self.RHO_0 = EPS, C, KW, RHO_W, RHO_I, RHO_0
        self.A = A
        self.SIMULATED_VARIABLES = SIMULATED_VARIABLES
        self.MAX_HEIGHT_COSMO = MAX_HEIGHT_COSMO
        self.GPM_SENSITIVITY, self.GPM_RADIAL_RES_KA, self.GPM_RADIAL_RES_KU = GPM_SENSITIVITY, GPM_RADIAL_RES_KA, GPM_RADIAL_RES_KU
        self.GPM_NO_BINS_KA, self.GPM_NO_BINS_KU = GPM_NO_BINS_KA, GPM_NO_BINS_KU
        self.update()

    def update(self):
        conf = cfg.CONFIG
        if not conf:
            return
        r = conf['radar']
        self.NVEL = C / (4 * 1E-6 * r['PRI'] * r['frequency'] * 1E09)
        self.VRES = 2 * self.NVEL / r['FFT_length']
        self.VARRAY = np.arange(-self.NVEL, self.NVEL + self.VRES, self.VRES)
        self.WAVELENGTH = C / (r['frequency'] * 1E09) * 1000
        self.PULSE_WIDTH = 2 * r['radial_resolution']
        sens = r.get('sensitivity')
        if isinstance(sens, (list, tuple)) and len(sens) == 3:
            # radar equation after Probert-Jones (constants.py:79-82)
            self.RADAR_CONSTANT_DB = 180 - 10 * np.log10(np.pi ** 3 * r['3dB_beamwidth'] ** 2
                                                         * self.KW ** 2 * self.PULSE_WIDTH / 1000) - 2 * sens[1] \
                + 10 * np.log10(1024 * np.log(2) * (self.WAVELENGTH / 1000.) ** 2)
        if r['type'] == 'ground':
            self.RANGE_RADAR = np.arange(r['radial_resolution'] / 2, r['range'], r['radial_resolution'])


constants = Constant_class()

# -----------------------------------------------------------------------------------------
# microphysics: one table per moment scheme, keyed by class letter
# -----------------------------------------------------------------------------------------
D_MIN = {'R': 0.2, 'S': 0.2, 'G': 0.2, 'H': 0.2}
D_MAX = {'R': 8, 'S': 20, 'G': 15, 'H': 15}


def _one_moment():
    t = {}
    b_g = 3.1
    g = {'N0': 4 * 1E3, 'b': b_g, 'a': 169.6 * (1000 ** -b_g), 'alpha': 442.0 * (1000 ** -0.89),
         'beta': 0.89, 'mu': 0.0}
    g['lambda_factor'] = g['a'] * g['N0'] * spe.gamma(g['b'] + 1)
    t['G'] = g
    b_s = 2
    s = {'N0': None, 'b': b_s, 'a': 0.038 / (1000 ** b_s), 'alpha': 4.9 * (1000 ** -0.25), 'beta': 0.25, 'mu': 0.0}
    s['lambda_factor'] = spe.gamma(s['b'] + 1)
    t['S'] = s
    mu_r = 0.5
    n00 = 8E6 / (1000 ** (1 + mu_r)) * (0.01) ** (-mu_r)
    r = {'N0': 0.1 * n00 * np.exp(3.2 * mu_r), 'a': np.pi / 6. * RHO_W, 'b': 3., 'alpha': 130 / np.sqrt(1000),
         'beta': 0.5, 'mu': mu_r}
    r['lambda_factor'] = r['a'] * r['N0'] * spe.gamma(1. + r['b'] + r['mu'])
    t['R'] = r
    for p in t.values():
        p['nu'] = 1
        p['vel_factor'] = spe.gamma(p['mu'] + p['beta'] + 1)
        p['ntot_factor'] = spe.gamma(p['mu'] + 1)
    return t


ONE_MOMENT = _one_moment()

# massic generalized-gamma parameters of the 2-moment scheme: D = a_ x^b_, v = alpha_ x^beta_
_TWO_MOMENT_MASSIC = {
    'G': dict(a_=0.142, b_=0.314, alpha_=86.89371, beta_=0.268325, nu_=1. / 3., mu_=1.0, x_min=1E-09, x_max=5E-04),
    'S': dict(a_=2.4, b_=0.455, alpha_=4.2, beta_=0.092, nu_=0.5, mu_=0.0, x_min=1E-10, x_max=2E-05),
    'R': dict(a_=0.124, b_=1. / 3., alpha_=114.0137, beta_=0.23437, nu_=1. / 3., mu_=0.0, x_min=2.6E-10, x_max=3E-06),
    'H': dict(a_=0.1366, b_=1. / 3., alpha_=39.3, beta_=1. / 6., nu_=1. / 3., mu_=1.0, x_min=2.6E-09, x_max=5E-04),
}


def _two_moment():
    t = {}
    for k, m in _TWO_MOMENT_MASSIC.items():
        b = 1. / m['b_']
        a = m['a_'] ** (-1 / m['b_'])
        if k in ('G', 'S'):
            b = ONE_MOMENT[k]['b']
        beta = m['beta_'] / m['b_']
        alpha = m['alpha_'] * m['a_'] ** (-m['beta_'] / m['b_'])
        nu = m['nu_'] / m['b_']
        mu = (m['mu_'] + 1) / m['b_'] - 1
        lam_f = 1. / a * spe.gamma((mu + 1) / nu) / spe.gamma((mu + b + 1) / nu)
        a = a * 1000 ** (-b)
        if k in ('G', 'S'):
            a = ONE_MOMENT[k]['a']
        alpha = alpha * 1000 ** (-beta)
        t[k] = {'N0': ONE_MOMENT['G']['N0'] if k == 'G' else None, 'a': a, 'b': b, 'alpha': alpha, 'beta': beta,
                'mu': mu, 'nu': nu, 'lambda_factor': lam_f,
                'vel_factor': spe.gamma((mu + beta + 1) / nu), 'ntot_factor': spe.gamma((mu + 1) / nu),
                'x_min': m['x_min'], 'x_max': m['x_max']}
    return t


TWO_MOMENT = _two_moment()
