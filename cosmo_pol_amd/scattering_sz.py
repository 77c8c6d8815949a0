"""get_radar_observables / get_pol_from_sz with the reference signatures
(scatter/scattering_sz.py:17,164); the arithmetic runs in the HIP kernels."""
import numpy as np

from . import runtime
from ._lib import OBS_NAMES
from .beam import Beam
from .constants import constants


def _pack(list_beams, len_beams):
    names = list(list_beams[0].values.keys())
    vals = {}
    for k in names:
        vals[k] = np.stack([_fit(np.asarray(b.values[k], dtype=np.float32), len_beams) for b in list_beams])
    elev = np.stack([_fit(np.asarray(b.elev_profile, dtype=np.float32), len_beams) for b in list_beams])
    mask = np.stack([_fit(np.asarray(b.mask, dtype=np.float64), len_beams, pad=1.) for b in list_beams])
    w = np.array([b.GH_weight for b in list_beams], dtype=np.float64)
    return vals, elev, mask, w


def _fit(a, n, pad=np.nan):
    if len(a) >= n:
        return a[:n]
    return np.concatenate([a, np.full(n - len(a), pad, dtype=a.dtype)])


def get_radar_observables(list_beams, lut_sz, device=0):
    """list[Beam] of one ray (from get_profiles_GH) -> Beam of ZH, ZDR, ZV, KDP, DELTA_HV, PHIDP, RHOHV,
    AH, AV in linear units.  Like the reference it truncates the sub-beam arrays to the central
    beam's length and folds beam.elev_profile into [0, 90] in place (scattering_sz.py:56-68)."""
    st = runtime.get_engine(device, doppler=0)
    runtime.ensure_luts(st, lut_sz)
    idx_0 = int(len(list_beams) / 2)
    len_beams = len(list_beams[idx_0].dist_profile)
    for b in list_beams:
        for k in b.values.keys():
            b.values[k] = b.values[k][0:len_beams]
    vals, elev, mask, w = _pack(list_beams, len_beams)
    obs, mask_out, _ = st.eng.ctx.observables_from_beams(vals, elev, mask, w)
    for b in list_beams:                      # visible side effect of the reference
        e = b.elev_profile
        e[e > 90] = 180 - e[e > 90]
        e[e < 0] = -e[e < 0]
    c = list_beams[idx_0]
    rad_obs = {n: obs[i].copy() for i, n in enumerate(OBS_NAMES)}
    return Beam(rad_obs, mask_out, c.lats_profile, c.lons_profile, c.dist_profile, c.heights_profile)


def get_pol_from_sz(sz):
    """scattering_sz.py:164-196 for an [n, 12] array of integrated scattering components (host side;
    the batched path evaluates the same expressions in float32 inside k_gate_reduce)."""
    wl = constants.WAVELENGTH
    sz = np.asarray(sz)
    b = sz[:, 0] - sz[:, 1] - sz[:, 2] + sz[:, 3]
    c = sz[:, 0] + sz[:, 1] + sz[:, 2] + sz[:, 3]
    xs_h, xs_v = 2 * np.pi * b, 2 * np.pi * c
    zfac = wl ** 4 / (np.pi ** 5 * constants.KW ** 2)
    with np.errstate(all='ignore'):
        rhohv = np.sqrt(((sz[:, 4] + sz[:, 7]) ** 2 + (sz[:, 6] - sz[:, 5]) ** 2) / (b * c))
        zdr = xs_h / xs_v
    kdp = 1e-3 * (180.0 / np.pi) * wl * (sz[:, 10] - sz[:, 8])
    ah = 4.343e-3 * (2 * wl * sz[:, 11])
    av = 4.343e-3 * (2 * wl * sz[:, 9])
    delta_hv = np.arctan2(sz[:, 5] - sz[:, 6], -sz[:, 4] - sz[:, 7])
    return zfac * xs_h, zfac * xs_v, zdr, rhohv, kdp, ah, av, delta_hv
