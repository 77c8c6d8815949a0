"""Space-borne (GPM DPR) swaths: viewing geometry, the swath run and the swath container
(gpm/GPM_simulator.py, radar_operator.py:350-424).

The reference simulates every beam of the swath as a ray from the satellite: azimuth towards the beam's
footprint, elevation from the triangle earth centre - satellite - footprint, straight propagation
(atm_refraction.get_GPM_refraction), and only the gates below 35 km.  Here one scan line (all beams share the
satellite position) is one call of the ray kernels: the host computes the (S, H, E) profile of every distinct
total elevation (beam + vertical sub-beam offset) and hands them to the library as a refraction table, the
profiles padded to a common length with NaN heights.  A padded gate interpolates to "below the terrain"
(mask 1), which is what the reference's handling of sub-beams of unequal length amounts to
(scattering_sz.py:60-62,144-151), and every beam's output is cut at the length of its central sub-beam.

The DPR file itself is HDF5; h5py is optional: get_GPM_angles takes either a file name (h5py needed) or a
dictionary with the six arrays it reads.  pyproj is not used: the azimuth comes from Vincenty's inverse
formula on WGS84 (sub-millimetre against Karney's for these distances) and the ECEF position from the closed
form.
"""
import copy

import numpy as np

from . import constants as cst

ANG2RAD = np.pi / 180.
WGS84_A = 6378137.0
WGS84_F = 1 / 298.257223563


def get_group(band):
    """GPM_simulator.py:22-29."""
    return {'Ku': 'NS', 'Ka': 'HS', 'both': 'MS'}[band]


def geodetic_to_ecef(lat, lon, h=0.):
    """WGS84 latitude / longitude (degrees), height (m) -> earth-centred earth-fixed x, y, z (m)."""
    lat, lon = np.asarray(lat, dtype=np.float64) * ANG2RAD, np.asarray(lon, dtype=np.float64) * ANG2RAD
    e2 = WGS84_F * (2 - WGS84_F)
    n = WGS84_A / np.sqrt(1 - e2 * np.sin(lat) ** 2)
    return (n + h) * np.cos(lat) * np.cos(lon), (n + h) * np.cos(lat) * np.sin(lon), (n * (1 - e2) + h) * np.sin(lat)


def geodesic_inverse_azimuth(lat1, lon1, lat2, lon2):
    """Forward azimuth (degrees, clockwise from north, in (-180, 180]) and length (m) of the WGS84 geodesic
    from point 1 to point 2 (Vincenty 1975, iterated to 1e-14 rad); arrays broadcast."""
    f, a = WGS84_F, WGS84_A
    b = a * (1 - f)
    p1, p2 = np.asarray(lat1, dtype=np.float64) * ANG2RAD, np.asarray(lat2, dtype=np.float64) * ANG2RAD
    L = (np.asarray(lon2, dtype=np.float64) - np.asarray(lon1, dtype=np.float64)) * ANG2RAD
    L = (L + np.pi) % (2 * np.pi) - np.pi
    U1, U2 = np.arctan((1 - f) * np.tan(p1)), np.arctan((1 - f) * np.tan(p2))
    sU1, cU1, sU2, cU2 = np.sin(U1), np.cos(U1), np.sin(U2), np.cos(U2)
    lam = L * np.ones(np.broadcast(p1, p2, L).shape)
    for _ in range(100):
        sl, cl = np.sin(lam), np.cos(lam)
        ss = np.hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)
        cs = sU1 * sU2 + cU1 * cU2 * cl
        sig = np.arctan2(ss, cs)
        with np.errstate(invalid='ignore', divide='ignore'):
            sa = np.where(ss == 0, 0., cU1 * cU2 * sl / ss)
            c2a = 1 - sa ** 2
            c2m = np.where(c2a == 0, 0., cs - 2 * sU1 * sU2 / c2a)
        C = f / 16 * c2a * (4 + f * (4 - 3 * c2a))
        new = L + (1 - C) * f * sa * (sig + C * ss * (c2m + C * cs * (-1 + 2 * c2m ** 2)))
        done = np.all(np.abs(new - lam) < 1e-14)
        lam = new
        if done:
            break
    sl, cl = np.sin(lam), np.cos(lam)
    u2 = c2a * (a * a - b * b) / (b * b)
    A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
    B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
    ds = B * ss * (c2m + B / 4 * (cs * (-1 + 2 * c2m ** 2) - B / 6 * c2m * (-3 + 4 * ss ** 2) * (-3 + 4 * c2m ** 2)))
    dist = b * A * (sig - ds)
    az = np.arctan2(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl) / ANG2RAD
    return az, dist


def _read_dpr(gpm, band):
    if isinstance(gpm, dict):
        return gpm
    try:
        import h5py
    except ImportError:
        raise ImportError('reading a DPR file needs h5py; pass a dictionary with Latitude, Longitude, scLat, '
                          'scLon, dprAlt and scPos instead')
    group = get_group(band)
    with h5py.File(gpm, 'r') as f:
        nav = f[group]['navigation']
        return {'Latitude': f[group]['Latitude'][:], 'Longitude': f[group]['Longitude'][:], 'scLat': nav['scLat'][:],
                'scLon': nav['scLon'][:], 'dprAlt': nav['dprAlt'][:], 'scPos': nav['scPos'][:]}


def get_GPM_angles(gpm, band='Ku'):
    """GPM_simulator.py:186-235.  gpm: DPR file name or {'Latitude', 'Longitude': [N, M] footprints, 'scLat',
    'scLon', 'dprAlt': [N] sub-satellite point and altitude, 'scPos': [N, 3] ECEF position}.  Returns azimuths,
    elevations (positive downwards), ranges to the footprint on the ellipsoid, all [N, M], and coords_GPM
    [N, 3] = (latitude, longitude, altitude) of the satellite per scan line."""
    d = _read_dpr(gpm, band)
    lat_2D, lon_2D = np.asarray(d['Latitude'], dtype=np.float64), np.asarray(d['Longitude'], dtype=np.float64)
    clat, clon = np.asarray(d['scLat'], dtype=np.float64), np.asarray(d['scLon'], dtype=np.float64)
    alt, pos = np.asarray(d['dprAlt'], dtype=np.float64), np.asarray(d['scPos'], dtype=np.float64)
    azimuths, _ = geodesic_inverse_azimuth(clat[:, None], clon[:, None], lat_2D, lon_2D)
    sx, sy, sz = geodetic_to_ecef(lat_2D, lon_2D, 0.)
    ranges = np.sqrt((sx - pos[:, None, 0]) ** 2 + (sy - pos[:, None, 1]) ** 2 + (sz - pos[:, None, 2]) ** 2)
    H = np.sqrt((pos ** 2).sum(axis=1))[:, None]
    RE = H - alt[:, None]
    with np.errstate(invalid='ignore'):
        theta = -np.arcsin((H ** 2 + ranges ** 2 - RE ** 2) / (2 * H * ranges)) / np.pi * 180.
    theta = np.where(np.isnan(theta), -90., theta)          # rounding just past the nadir (:230-231)
    return azimuths, -theta, ranges, np.vstack((clat, clon, alt)).T


class SimulatedGPM(object):
    """GPM_simulator.py:31-88: the beams of a swath [N scan lines, M beams] on a common range axis that starts
    at the ground: only the gates above the model terrain (mask < 1), flipped, zero padded at the far end;
    bin_surface = number of gates from the first one below the terrain to the end of the beam."""

    def __init__(self, list_beams, dim, band):
        N, M = dim
        names = list(list_beams[0].values.keys())
        max_len = max(len(b.dist_profile) for b in list_beams)
        self.bin_surface = np.zeros((N, M))
        self.data = {k: np.zeros((N, M, max_len)) for k in names}
        self.lats = np.full((N, M, max_len), np.nan)
        self.lons = np.full((N, M, max_len), np.nan)
        for idx, b in enumerate(list_beams):
            i, j = divmod(idx, M)
            mask = np.asarray(b.mask)
            below = np.flatnonzero(mask >= 1)
            self.bin_surface[i, j] = len(mask) - below[0] if len(below) else 0
            keep = mask < 1
            n = int(keep.sum())
            self.lats[i, j, :n] = np.asarray(b.lats_profile)[keep][::-1]
            self.lons[i, j, :n] = np.asarray(b.lons_profile)[keep][::-1]
            for k in names:
                self.data[k][i, j, :n] = np.asarray(b.values[k])[keep][::-1]
        self.band = band


def simulate_swath(eng, conf, az, elev, rang, coords_GPM, band='Ku'):
    """The beams of a swath as a list of Beam (row-major over [N, M]) after the sensitivity cut, i.e. what
    radar_operator.py:399-421 hands to SimulatedGPM.  eng: an Engine with model and lookup tables uploaded;
    conf: the option dictionary (its radar section is overridden per scan line)."""
    from .beam import Beam
    from .refraction import get_GPM_refraction
    from .engine import gauss_hermite_subbeams
    from ._lib import OBS_NAMES
    res = float(conf['radar']['radial_resolution'])
    N, M = az.shape
    _, el_off, _ = gauss_hermite_subbeams(conf)
    el_off = np.asarray(el_off, dtype=np.float64)
    central = len(el_off) // 2                       # the sub-beam whose length the outputs take (scattering_sz.py:40-41)
    beams = [None] * (N * M)
    for i in range(N):
        c = copy.deepcopy(conf)
        c['radar']['coords'] = [float(v) for v in coords_GPM[i]]
        # profiles per beam and distinct total elevation; a beam whose elevations coincide with those of a beam
        # already in the group goes to the next group (the table is keyed by elevation, and the range is not)
        todo = list(range(M))
        while todo:
            group, table, rest = [], {}, []
            for j in todo:
                cj = copy.deepcopy(c)
                cj['radar']['range'] = float(rang[i, j])
                keys_j = el_off + float(elev[i, j])
                rows = {float(k): get_GPM_refraction(float(k), cj) for k in np.unique(keys_j)}
                if any(k in table for k in rows):
                    rest.append(j)
                    continue
                table.update(rows)
                group.append((j, len(rows[float(keys_j[central])][0])))
            todo = rest
            keys = np.array(sorted(table))
            n_gates = max(max(len(table[k][0]) for k in keys), 1)
            S, H, E = (np.full((len(keys), n_gates), np.nan, dtype=np.float32) for _ in range(3))
            for r, k in enumerate(keys):
                s_, h_, e_ = table[k]
                S[r, :len(s_)], H[r, :len(s_)], E[r, :len(s_)] = s_, h_, e_
                # a padded gate keeps the last position and elevation (inside the domain) and a NaN height
                S[r, len(s_):] = s_[-1] if len(s_) else 0.
                E[r, len(s_):] = e_[-1] if len(s_) else 90.
            c['radar']['range'] = n_gates * res
            eng.configure(c, doppler=0, sensitivity_cut=True)
            eng.set_gauss_hermite()
            eng.ctx.set_refraction_table(keys, S, H, E)
            js = [j for j, _ in group]
            out = eng.ctx.run_rays(np.asarray(az[i, js], dtype=np.float64), np.asarray(elev[i, js], dtype=np.float64))
            for r, (j, n) in enumerate(group):
                vals = {name: out['obs'][k][r, :n].copy() for k, name in enumerate(OBS_NAMES)}
                beams[i * M + j] = Beam(vals, out['mask'][r, :n].copy(), out['lat'][r, :n].copy(),
                                        out['lon'][r, :n].copy(), out['dist'][r, :n].copy(), out['height'][r, :n].copy())
    eng.ctx.set_refraction_table()
    return beams
