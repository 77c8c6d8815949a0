"""Global configuration (same YAML schema, defaults and silent fall-backs as utilities/cfg.py)."""
import copy
import numbers

import numpy as np
import yaml

FLAG_NUMBER = 'ANY_NUMBER'
CONFIG = {}

DEFAULTS = {
    'radar': {'coords': [46.42563, 6.01, 1672.7], 'type': 'ground', 'frequency': 5.6, 'range': 150000,
              'radial_resolution': 500, 'PRI': 700, 'FFT_length': 256, 'sensitivity': [-5, 10000],
              '3dB_beamwidth': 1.},
    'attenuation': {'correction': 1},
    'refraction': {'scheme': 1},
    'integration': {'scheme': 1, 'nv_GH': 9, 'nh_GH': 3, 'n_gaussians': 7, 'antenna_diagram': '',
                    'weight_threshold': 1., 'nr_GH': 7, 'na_GL': 7},
    'doppler': {'scheme': 1, 'turbulence_correction': 0},
    'microphysics': {'scheme': 1},
}

VALID_VALUES = {
    'radar': {'coords': FLAG_NUMBER, 'type': ['ground', 'GPM'], 'frequency': [2, 'to', 35.6],
              'range': [5000, 'to', 500000], 'radial_resolution': [25, 'to', 5000], 'PRI': [10, 'to', 3000],
              'FFT': [16, 'to', 2048], 'sensitivity': FLAG_NUMBER, '3dB_beamwidth': [0.1, 'to', 10]},
    'attenuation': {'correction': [0, 1]},
    'refraction': {'scheme': [1, 2]},
    'integration': {'scheme': [1, 2, 3, 4], 'nv_GH': np.arange(1, 31, 2), 'nh_GH': np.arange(1, 31, 2),
                    'n_gaussians': np.arange(1, 13, 2), 'weight_threshold': [0., 'to', 1.],
                    'nr_GH': np.arange(1, 31, 2), 'na_GL': np.arange(1, 31, 2)},
    'doppler': {'scheme': [1, 2, 3, 4], 'turbulence_correction': [0, 1]},      # 4: doppler_sz.py:36 (3 + turbulence)
    'microphysics': {'scheme': [1, 2]},
}


def check_valid(section, key, value):
    """(ok, value-or-default, message) like utilities/cfg.py:85-128; an out-of-range
    number falls back to the default."""
    valid = VALID_VALUES.get(section, {}).get(key, None)
    if valid is None:
        return True, value, ''
    if isinstance(valid, str) and valid == FLAG_NUMBER:
        vals = value if hasattr(value, '__len__') else [value]
        if all(isinstance(v, numbers.Number) for v in vals):
            return True, value, ''
        return False, DEFAULTS[section][key], 'Invalid value for "%s: %s": all values must be numbers' % (section, key)
    if isinstance(valid, list) and 'to' in valid:
        if float(valid[0]) <= value <= float(valid[2]):
            return True, value, ''
        return False, DEFAULTS[section][key], 'Invalid value for "%s: %s", using default %s' % (
            section, key, DEFAULTS[section][key])
    if value in valid:
        return True, value, ''
    return False, value, 'Invalid value for "%s: %s"' % (section, key)


def init(options_file):
    """Read a YAML option file into CONFIG; unknown keys are ignored and missing ones take the
    defaults (utilities/cfg.py:130-152)."""
    global CONFIG
    try:
        with open(options_file, 'r') as fh:
            conf = yaml.safe_load(fh)
        if not isinstance(conf, dict):
            raise ValueError
    except Exception:
        CONFIG = copy.deepcopy(DEFAULTS)
        print('Could not find or read ' + str(options_file) + ' file, using default options')
        return
    for section in DEFAULTS:
        conf.setdefault(section, {})
        if conf[section] is None:
            conf[section] = {}
        for key in DEFAULTS[section]:
            conf[section].setdefault(key, DEFAULTS[section][key])
            ok, value, message = check_valid(section, key, conf[section][key])
            if message:
                print(message)
            conf[section][key] = value
    if conf['integration']['antenna_diagram'] != '':
        # utilities/cfg.py:154-163: fit a sum of Gaussians to the antenna diagram (used by integration scheme 2)
        from .quadrature import optimize_gaussians
        data = np.genfromtxt(conf['integration']['antenna_diagram'], delimiter=',')
        y = 10 * np.log10((10 ** (0.1 * data[:, 1])) ** 2)
        conf['integration']['antenna_params'] = optimize_gaussians(data[:, 0], y, conf['integration']['n_gaussians'])
    CONFIG = conf


def set_config(conf):
    """Install a ready-made configuration dictionary (deep-copied over the defaults)."""
    global CONFIG
    merged = copy.deepcopy(DEFAULTS)
    for sec, kv in (conf or {}).items():
        merged.setdefault(sec, {}).update(kv)
    CONFIG = merged
