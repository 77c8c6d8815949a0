"""Hydrometeor classes (parameters only) and their C-ABI descriptors.

The per-gate PSD evaluation (set_psd / get_N of hydrometeors/hydrometeors.py:58-64,127-155,
247-257,311-324,379-389) runs inside the HIP kernels; this module supplies the class-level
constants and which set_psd branch applies.
"""
from . import constants as cst
from ._lib import VAR_SLOTS

PSD_1MOM_FIXED_N0, PSD_1MOM_SNOW, PSD_2MOM = 0, 1, 2


class Hydrometeor(object):
    def __init__(self, kind, scheme):
        if kind == 'H':
            scheme = '2mom'                  # no 1-moment hail (hydrometeors.py:413)
        if scheme not in ('1mom', '2mom'):
            scheme = '1mom'                  # hydrometeors.py:27-28,180-181
        self.kind, self.scheme = kind, scheme
        p = (cst.ONE_MOMENT if scheme == '1mom' else cst.TWO_MOMENT)[kind]
        self.a, self.b, self.alpha, self.beta = p['a'], p['b'], p['alpha'], p['beta']
        self.mu, self.nu, self.N0 = p['mu'], p['nu'], p['N0']
        self.lambda_factor, self.vel_factor, self.ntot_factor = p['lambda_factor'], p['vel_factor'], p['ntot_factor']
        self.x_min, self.x_max = p.get('x_min', 0.), p.get('x_max', 0.)
        self.d_min, self.d_max = cst.D_MIN[kind], cst.D_MAX[kind]
        self.lambda_ = None

    def get_V(self, D):
        return self.alpha * D ** self.beta

    def get_D_from_V(self, V):
        return (1.0 / self.alpha * V) ** (1.0 / self.beta)

    def psd_descriptor(self):
        if self.scheme == '2mom':
            kind = PSD_2MOM
        elif self.kind == 'S':
            kind = PSD_1MOM_SNOW
        else:
            kind = PSD_1MOM_FIXED_N0
        return dict(psd_kind=kind, a=self.a, b=self.b, alpha=self.alpha, beta=self.beta, mu=self.mu,
                    nu=float(self.nu), lambda_factor=self.lambda_factor, vel_factor=self.vel_factor,
                    ntot_factor=self.ntot_factor, n0_fixed=float(self.N0) if self.N0 is not None else 0.,
                    x_min=self.x_min, x_max=self.x_max, d_min=self.d_min, d_max=self.d_max,
                    q_var=VAR_SLOTS['Q' + self.kind + '_v'],
                    qn_var=VAR_SLOTS['QN' + self.kind + '_v'] if kind == PSD_2MOM else -1,
                    t_var=VAR_SLOTS['T'])

    def doppler_descriptor(self, microphysics_scheme):
        """Descriptor of the class as the Doppler path feeds it.  With 2-moment model input the
        reference calls set_psd(QN, Q) on the 1-moment classes it instantiated (doppler_sz.py:57,143):
        rain / graupel then take QN as the mass density (hydrometeors.py:252,384) and snow takes QN
        as the temperature (hydrometeors.py:317)."""
        d = self.psd_descriptor()
        if microphysics_scheme == 2 and self.scheme == '1mom':
            if self.kind == 'S':
                d['t_var'] = VAR_SLOTS['QNS_v']
            else:
                d['q_var'] = VAR_SLOTS['QN' + self.kind + '_v']
        return d


class Rain(Hydrometeor):
    def __init__(self, scheme, sedi=False):
        Hydrometeor.__init__(self, 'R', scheme)


class Snow(Hydrometeor):
    def __init__(self, scheme):
        Hydrometeor.__init__(self, 'S', scheme)


class Graupel(Hydrometeor):
    def __init__(self, scheme):
        Hydrometeor.__init__(self, 'G', scheme)


class Hail(Hydrometeor):
    def __init__(self, scheme='2mom'):
        Hydrometeor.__init__(self, 'H', scheme)


def create_hydrometeor(hydrom_type, scheme, sedi=False):
    """hydrometeors.py:13-21."""
    return {'R': Rain, 'S': Snow, 'G': Graupel, 'H': Hail}[hydrom_type](scheme)


def hydrom_types(scheme):
    return ['R', 'S', 'G'] if scheme == 1 else ['R', 'S', 'G', 'H']
