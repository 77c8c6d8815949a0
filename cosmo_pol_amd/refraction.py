"""Beam propagation through the atmosphere (refraction/atm_refraction.py), host side.

Scheme 1, the 4/3 earth-radius model, runs inside the geometry kernel on the GPU; ref_4_3 here is its host
twin for callers of the reference's function.  Scheme 2 integrates the ray equation through the model's own
refractivity profile above the radar (horizontally homogeneous, atm_refraction.py:45-84).  It depends on the
total elevation of a sub-beam only, so the host solves it once per distinct (elevation + vertical offset) and
hands the (S, H, E) profiles to the library as a table (cpol_set_refraction_table).
"""
import numpy as np

RAD2DEG = 180.0 / np.pi
DEG2RAD = 1 / RAD2DEG


def get_earth_radius(latitude):
    """atm_refraction.py:18-21 (the argument is used as radians whatever the caller passes)."""
    a = 6378.1370 * 1000
    b = 6356.7523 * 1000
    return np.sqrt(((a ** 2 * np.cos(latitude)) ** 2 + (b ** 2 * np.sin(latitude)) ** 2) /
                   ((a * np.cos(latitude)) ** 2 + (b * np.sin(latitude)) ** 2))


def ref_4_3(range_vec, elevation_angle, coords_radar):
    """atm_refraction.py:110-123: arc distance S, height H (float32) and elevation E (degrees) of every gate.
    The earth radius is evaluated at coords_radar[1], the longitude, as the reference does."""
    el = elevation_angle * np.pi / 180.
    ke = 4. / 3.
    alt = coords_radar[2]
    R = get_earth_radius(coords_radar[1])
    H = np.sqrt(range_vec ** 2 + (ke * R) ** 2 + 2 * range_vec * ke * R * np.sin(el)) - ke * R + alt
    S = ke * R * np.arcsin((range_vec * np.cos(el)) / (ke * R + H))
    E = el + np.arctan(range_vec * np.cos(el) / (range_vec * np.sin(el) + ke * R + alt))
    return S.astype('float32'), H.astype('float32'), E.astype('float32') * RAD2DEG


def get_GPM_refraction(elevation, conf=None):
    """atm_refraction.py:125-149: gates of a space-borne beam looking down at `elevation` degrees below the
    horizontal (taken positive), straight propagation (ke = 1) from radar.coords, every radial_resolution up to
    radar.range.  Only the gates below MAX_HEIGHT_COSMO are returned, so the profiles of two elevations differ
    in length.  The earth radius is evaluated at coords[0] in degrees used as radians, as the reference does."""
    from . import cfg, constants
    r = (conf if conf is not None else cfg.CONFIG)['radar']
    latitude, altitude_radar = r['coords'][0], r['coords'][2]
    elev_rad = elevation * DEG2RAD
    RE = get_earth_radius(latitude)
    range_vec = np.arange(r['radial_resolution'] / 2., r['range'], r['radial_resolution'])
    H = -(np.sqrt(range_vec ** 2 + RE ** 2 + 2 * range_vec * RE * np.sin(elev_rad)) - RE) + altitude_radar
    S = RE * np.arcsin((range_vec * np.cos(elev_rad)) / (RE + H))
    E = elevation - np.arctan(range_vec * np.cos(elev_rad) / (range_vec * np.sin(elev_rad) + RE + altitude_radar)) * RAD2DEG
    low = H < constants.MAX_HEIGHT_COSMO
    return S[low].astype('float32'), H[low].astype('float32'), E[low].astype('float32')


def wgs_to_rotated(lat, lon, pole_lat_s, pole_lon_s):
    """Geographic -> rotated-pole coordinates (rlat, rlon) in degrees, pole given by its southern pole as in the
    model's metadata.  Standard rotated-pole transform (pycosmo's own is not part of the reference snapshot)."""
    pollat, pollon = -pole_lat_s, pole_lon_s - 180.0
    if pollon < -180.0:
        pollon += 360.0
    sp, cp = np.sin(pollat * DEG2RAD), np.cos(pollat * DEG2RAD)
    ph, dl = np.asarray(lat) * DEG2RAD, (np.asarray(lon) - pollon) * DEG2RAD
    a0 = cp * np.cos(ph) * np.cos(dl) + sp * np.sin(ph)
    a1 = -np.sin(dl) * np.cos(ph)
    a2 = -sp * np.cos(ph) * np.cos(dl) + cp * np.sin(ph)
    return np.arcsin(a0) * RAD2DEG, np.arctan2(a1, a2) * RAD2DEG


def _linear_extrapolating(x, y):
    """Scalar piecewise-linear function through (x, y), continued with the end slopes (atm_refraction.py:90-108).
    The node arrays keep their dtype (float32 when they come from the model): the segment slope is formed in that
    dtype and applied in float64, which is what scipy's interp1d does with such arrays."""
    x = np.asarray(x)
    y = np.asarray(y)
    slope = (y[1:] - y[:-1]) / (x[1:] - x[:-1])
    x64, y64, s64 = x.astype(np.float64), y.astype(np.float64), slope.astype(np.float64)
    d0, d1 = float(y[1] - y[0]), float(y[-1] - y[-2])
    w0, w1 = float(x[1] - x[0]), float(x[-1] - x[-2])
    n = len(x)

    def f(v):
        v = float(v)
        if v < x64[0]:
            return y64[0] + (v - x64[0]) * d0 / w0
        if v > x64[-1]:
            return y64[-1] + (v - x64[-1]) * d1 / w1
        i = min(max(int(np.searchsorted(x64, v)), 1), n - 1) - 1
        return s64[i] * (v - x64[i]) + y64[i]
    return f


def refractive_index_profile(N, coords_radar, radar_type='ground'):
    """Heights and refractive index of the model column used for the radar (atm_refraction.py:47-67).  The
    reference rounds the radar's fractional row index and uses it for BOTH horizontal indices (:57,59)."""
    proj = N.attributes['proj_info']
    rlat, rlon = wgs_to_rotated(coords_radar[0], coords_radar[1], proj['Latitude_of_southern_pole'],
                                proj['Longitude_of_southern_pole'])
    res = N.attributes['resolution']
    row = (float(rlat) - float(proj['La1'])) / res[1]
    i = int(np.round(row))
    n = 1 + np.asarray(N.data)[:, i, i] * 1E-6           # stays in the field's float32, as in the reference
    h = np.asarray(N.attributes['z-levels'])[:, i, i]
    if radar_type == 'ground':
        h, n = h[::-1], n[::-1]
    return h, n


def ref_ODE_s(range_vec, elevation_angle, coords_radar, N, radar_type='ground', profile=None):
    """atm_refraction.py:45-84: h'' from the refractive-index gradient, integrated along range with LSODA.
    As in the reference, E is the integrated dh/dr (the sine of the elevation) scaled by 180/pi, and the arc
    distance is accumulated with cos of that quantity."""
    from scipy.integrate import odeint
    h, n = profile if profile is not None else refractive_index_profile(N, coords_radar, radar_type)
    RE = get_earth_radius(coords_radar[0])
    n_of_h = _linear_extrapolating(h, n)
    dn_of_h = _linear_extrapolating(h[0:-1], np.diff(n) / np.diff(h))

    def rhs(z, r):
        hh, u = z
        k = (1. / n_of_h(hh)) * dn_of_h(hh) + 1. / (RE + hh)
        return [u, -u ** 2 * k + k]

    range_vec = np.asarray(range_vec, dtype=np.float64)
    Z = odeint(rhs, [coords_radar[2], np.sin(elevation_angle / 180. * np.pi)], range_vec)
    H, E = Z[:, 0], Z[:, 1]
    dR = range_vec[1] - range_vec[0]
    step = RE * np.arcsin((np.cos(E[:-1]) * dR) / (RE + H[1:]))
    S = np.concatenate(([0.], np.cumsum(step)))
    return S.astype('float32'), H.astype('float32'), E.astype('float32') * RAD2DEG


def get_radar_refraction(range_vec, elevation_angle, coords_radar, refraction_method, N=0):
    """atm_refraction.py:24-36."""
    if refraction_method == 1:
        return ref_4_3(range_vec, elevation_angle, coords_radar)
    if refraction_method == 2:
        return ref_ODE_s(range_vec, elevation_angle, coords_radar, N)
    raise ValueError('refraction scheme must be 1 (4/3 earth radius) or 2 (ray equation)')
