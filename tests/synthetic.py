"""Deterministic synthetic inputs: COSMO-like model fields, scattering lookup
tables and per-sub-beam profiles.

The reference needs COSMO GRIB files and T-matrix lookup tables that are not
available offline; benchmarks and parity tests use these stand-ins instead.  Only
shapes, axes, dtypes and smooth / positive behaviour follow the real inputs:
  * model fields [nz, ny, nx] float32, level 0 = model top, terrain-following
    heights descending with the level index (what interpolation_c.c:108-135 assumes);
  * lookup tables [e, t, d, 12] float32 with the axes of lookup/compute_lut_sz.py:30-38,104
    (e = 0..90 step 2, t = 262..314 / 200..276 step 2, 1024 diameters, 12 components in
    the order Z11 Z12 Z21 Z22 Z33 Z34 Z43 Z44 ReS00 ImS00 ReS11 ImS11).
"""
import numpy as np

BASE_VARIABLES = ['U', 'V', 'W', 'QR_v', 'QS_v', 'QG_v', 'RHO', 'T']       # radar_operator.py:32
BASE_VARIABLES_2MOM = ['QH_v', 'QNH_v', 'QNR_v', 'QNS_v', 'QNG_v']         # radar_operator.py:33

D_RANGE = {'R': (0.2, 8.), 'S': (0.2, 20.), 'G': (0.2, 15.), 'H': (0.2, 15.)}
X_RANGE = {'R': (2.6E-10, 3E-06), 'S': (1E-10, 2E-05), 'G': (1E-09, 5E-04), 'H': (2.6E-09, 5E-04)}
C_LIGHT = 299792458.


def lut_axes(kind, n_e=46, n_t=None, n_d=1024):
    e_axis = np.arange(0, 2 * n_e, 2, dtype=np.float64)
    if kind == 'R':
        t_full = np.arange(262, 316, 2, dtype=np.float64)
    else:
        t_full = np.arange(200, 278, 2, dtype=np.float64)
    t_axis = t_full if n_t is None else t_full[len(t_full) - n_t:] if kind != 'R' else t_full[:n_t]
    d_axis = np.linspace(D_RANGE[kind][0], D_RANGE[kind][1], n_d).astype('float32')
    return e_axis, t_axis, d_axis


def synthetic_lut(kind, frequency=5.6, n_e=46, n_t=None, n_d=1024):
    """Smooth Rayleigh-like scattering table -> (e_axis, t_axis, d_axis, table[e,t,d,12] f32)."""
    e_axis, t_axis, d_axis = lut_axes(kind, n_e, n_t, n_d)
    lam = C_LIGHT / (frequency * 1e9) * 1000.
    k0 = (np.pi ** 2 / lam ** 2) * 0.48
    cls = {'R': (1.0, 0.12, 3e-3), 'S': (0.08, 0.04, 2e-4), 'G': (0.35, 0.03, 6e-4), 'H': (0.8, 0.06, 1.5e-3)}[kind]
    dens, obl, absb = cls
    e = np.deg2rad(e_axis)[:, None, None]
    t = ((t_axis - 273.15) / 40.)[None, :, None]
    d = d_axis.astype(np.float64)[None, None, :]
    dn = d / D_RANGE[kind][1]
    amp = k0 * dens * d ** 3 * (1 + 0.05 * t)
    # Mie-like roll-off so that large diameters do not dominate without bound
    amp = amp / (1 + (d / (0.35 * lam)) ** 2)
    ani = obl * dn * np.cos(e) ** 2
    shh = amp * (1 + ani) * np.exp(1j * (0.02 * dn + 0.01 * t))
    svv = amp * (1 - ani) * np.exp(1j * (0.015 * dn))
    phh, pvv = np.abs(shh) ** 2, np.abs(svv) ** 2
    cross = shh * np.conj(svv)
    fwd_h = amp * (1 + 1.2 * ani) * (1 + 1j * absb * (1 + dn) * (1 + 0.3 * t))
    fwd_v = amp * (1 - 1.2 * ani) * (1 + 1j * absb * (1 + 0.9 * dn) * (1 + 0.3 * t))
    tab = np.empty(np.broadcast(e, t, d).shape + (12,), dtype=np.float32)
    tab[..., 0] = 0.5 * (phh + pvv)
    tab[..., 1] = 0.5 * (pvv - phh)
    tab[..., 2] = 0.5 * (pvv - phh)
    tab[..., 3] = 0.5 * (phh + pvv)
    tab[..., 4] = cross.real
    tab[..., 5] = cross.imag
    tab[..., 6] = -cross.imag
    tab[..., 7] = cross.real
    tab[..., 8] = fwd_v.real
    tab[..., 9] = fwd_v.imag
    tab[..., 10] = fwd_h.real
    tab[..., 11] = fwd_h.imag
    return e_axis, t_axis, d_axis, tab


def hydrom_types(scheme):
    return ['R', 'S', 'G'] if scheme == 1 else ['R', 'S', 'G', 'H']


def variable_names(scheme):
    return BASE_VARIABLES + (BASE_VARIABLES_2MOM if scheme == 2 else [])


def _hydro_fields(z, T, rng, scheme, zero_frac):
    """Mass / number densities on an array of heights z (any shape)."""
    shp = z.shape
    f32 = np.float32
    u = lambda lo, hi: (lo + (hi - lo) * rng.random(shp, dtype=f32))
    keep = lambda: (rng.random(shp, dtype=f32) >= zero_frac).astype(f32)
    out = {}
    warm = (T > 273.15).astype(f32)
    near = np.exp(-((T - 271.) / 6.) ** 2).astype(f32)
    out['QR_v'] = f32(1e-4) * u(0.5, 1.5) * warm * keep()
    out['QS_v'] = f32(2e-4) * u(0.5, 1.5) * (1 - warm) * keep()
    out['QG_v'] = f32(2e-5) * u(0.5, 1.5) * near * (near > 0.05) * keep()
    if scheme == 2:
        cell = np.exp(-((T - 265.) / 10.) ** 2).astype(f32)
        out['QH_v'] = f32(1e-3) * u(0.0, 1.0) * cell * (cell > 0.1) * keep()
        for h in ('R', 'S', 'G', 'H'):
            lo, hi = X_RANGE[h]
            xbar = np.exp(u(np.log(lo), np.log(hi)))
            # a few gates with number density but zero mass and vice versa
            qn = out['Q%s_v' % h] / xbar
            out['QN%s_v' % h] = (qn * (rng.random(shp, dtype=f32) >= 0.02)).astype(f32)
    return {k: v.astype(f32) for k, v in out.items()}


def synthetic_model(nz=80, ny=774, nx=1158, scheme=2, seed=20161111, zero_frac=0.4,
                    llc=(-6.8, -4.4), res=(0.01, 0.01), pole=(-43.0, 10.0), ztop=22000.):
    """COSMO-like model state.  Returns dict(fields={name: [nz,ny,nx] f32}, zlev, proj_info,
    resolution, names)."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    yy = np.arange(ny, dtype=np.float64)[:, None]
    xx = np.arange(nx, dtype=np.float64)[None, :]
    topo = 1600. + 1400. * np.sin(2 * np.pi * yy / max(ny, 8) * 3.1) * np.cos(2 * np.pi * xx / max(nx, 8) * 2.3)
    topo = np.clip(topo, 200., 3000.)
    k = np.arange(nz, dtype=np.float64)[:, None, None]
    z = topo[None] + (1 - k / (nz - 1)) ** 1.5 * (ztop - topo[None])
    z = z.astype(f32)
    T = (288. - 6.5e-3 * z).astype(f32)
    fields = _hydro_fields(z, T, rng, scheme, zero_frac)
    shp = z.shape
    fields['T'] = T
    fields['U'] = (10 + 5 * (2 * rng.random(shp, dtype=f32) - 1)).astype(f32)
    fields['V'] = (10 + 5 * (2 * rng.random(shp, dtype=f32) - 1)).astype(f32)
    fields['W'] = (0.5 * rng.standard_normal(shp, dtype=f32)).astype(f32)
    fields['RHO'] = (1.225 * np.exp(-z / 8000.)).astype(f32)
    names = variable_names(scheme)
    proj = {'Lo1': llc[0], 'La1': llc[1], 'Lo2': llc[0] + (nx - 1) * res[0], 'La2': llc[1] + (ny - 1) * res[1],
            'Latitude_of_southern_pole': pole[0], 'Longitude_of_southern_pole': pole[1]}
    return {'fields': {n: fields[n] for n in names}, 'zlev': z, 'proj_info': proj,
            'resolution': [res[0], res[1]], 'names': names, 'time': 0}


def synthetic_beams(n_gates, n_sub, scheme, seed=7, elevation=3.0, res=500., zero_frac=0.35,
                    nan_tail=0, below_head=0):
    """Per-sub-beam profiles as produced by get_profiles_GH: list of dicts with keys
    values{name:[G] f32}, mask[G], elev[G] f32, dist[G] f32, heights[G] f32, GH_pt, GH_weight."""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    rr = np.arange(res / 2, res * n_gates, res)[:n_gates]
    w = rng.random(n_sub) + 0.2
    w = w / w.sum()
    beams = []
    for b in range(n_sub):
        el = elevation + 0.4 * (b - n_sub // 2) / max(n_sub, 1)
        h = (1000. + rr * np.sin(np.deg2rad(el)) + rr ** 2 / (2 * 8.5e6)).astype(f32)
        T = (290. - 6.5e-3 * h + 0.3 * rng.standard_normal(n_gates)).astype(f32)
        vals = _hydro_fields(h, T, rng, scheme, zero_frac)
        vals['T'] = T
        vals['U'] = (10 + 5 * (2 * rng.random(n_gates) - 1)).astype(f32)
        vals['V'] = (-4 + 5 * (2 * rng.random(n_gates) - 1)).astype(f32)
        vals['W'] = (0.5 * rng.standard_normal(n_gates)).astype(f32)
        vals['RHO'] = (1.225 * np.exp(-h / 8000.)).astype(f32)
        mask = np.zeros(n_gates)
        if below_head and b % 3 == 0:
            mask[:below_head] = 1
        if nan_tail and b % 2 == 0:
            mask[n_gates - nan_tail - b % 5:] = -1
        for k in vals:
            vals[k] = vals[k].copy()
            vals[k][mask != 0] = np.nan
        elev = (el + rr / 1.7e5).astype(f32)
        beams.append({'values': {k: vals[k] for k in variable_names(scheme)}, 'mask': mask,
                      'elev': elev, 'dist': (rr * np.cos(np.deg2rad(el))).astype(f32), 'heights': h,
                      'GH_pt': [30.0 + 0.3 * (b % 3 - 1), el], 'GH_weight': float(w[b])})
    return beams
