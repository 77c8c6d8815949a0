"""Helpers with the reference names (utilities/utilities.py:18-117)."""
import numpy as np

from . import cfg
from .constants import SIMULATED_VARIABLES


def nan_cumprod(x):
    x[np.isnan(x)] = 1
    return np.cumprod(x)


def nan_cumsum(x):
    x[np.isnan(x)] = 0
    return np.cumsum(x)


def _pad_to_common(x, y, cst):
    x, y = np.array(x), np.array(y)
    px = [(0, max(0, b - a)) for a, b in zip(x.shape, y.shape)]
    py = [(0, max(0, a - b)) for a, b in zip(x.shape, y.shape)]
    return np.pad(x, px, 'constant', constant_values=cst), np.pad(y, py, 'constant', constant_values=cst)


def sum_arr(x, y, cst=0):
    return np.sum(_pad_to_common(x, y, cst), axis=0)


def nansum_arr(x, y, cst=0):
    return np.nansum(_pad_to_common(x, y, cst), axis=0)


def sensitivity_threshold(ranges):
    """Threshold in dBZ per gate for the configured radar.sensitivity (utilities.py:70-81);
    None when the configuration has no usable form."""
    sens = cfg.CONFIG['radar']['sensitivity']
    if not isinstance(sens, list):
        sens = [sens]
    with np.errstate(divide='ignore'):
        if len(sens) == 3:
            from .constants import constants
            return sens[0] + constants.RADAR_CONSTANT_DB + sens[2] + 20 * np.log10(ranges / 1000.)
        if len(sens) == 2:
            return (sens[0] - 20 * np.log10(sens[1] / 1000.)) + 20 * np.log10(ranges / 1000.)
        if len(sens) == 1:
            return np.full(len(ranges), float(sens[0]))
    return None


def cut_at_sensitivity(list_beams):
    """NaN-out the simulated variables where 10 log10(ZH) is below the threshold; accepts a list of
    beams or a list of lists (sweeps), like utilities.py:68-105.  Note r = res * arange(n): the first
    gate sits at r = 0 (threshold -inf), as in the reference."""
    def one(b):
        rr = cfg.CONFIG['radar']['radial_resolution'] * np.arange(len(b.dist_profile))
        thr = sensitivity_threshold(rr)
        if thr is None:
            print('Sensitivity parameters are invalid, cannot cut at specified sensitivity')
            return
        with np.errstate(all='ignore'):
            m = 10 * np.log10(b.values['ZH']) < thr
        for k in b.values.keys():
            if k in SIMULATED_VARIABLES:
                b.values[k][m] = np.nan
    for item in list_beams:
        if isinstance(item, list):
            for b in item:
                one(b)
        else:
            one(item)
    return list_beams


def combine_beams(list_of_beams):
    x = list_of_beams[0]
    for l in list_of_beams:
        if (l.dist_profile == x.dist_profile).all():
            x.values.update(l.values)
        else:
            print('Beams are not defined on the same set of coordinates, aborting')
            return
    return x
