"""Model-state container standing where pycosmo's variable objects stand in the reference
(radar_operator.py:138: `.data`, `.attributes['z-levels' | 'proj_info' | 'resolution' | 'time']`).
GRIB ingest (pycosmo) is outside the hot path; states come from memory or from .npz files."""
import numpy as np


class ModelVariable(object):
    def __init__(self, name, data, attributes):
        self.name = name
        self.data = data
        self.attributes = attributes


def variables_from_arrays(fields, zlev, proj_info, resolution, time=0):
    """fields: {name: [nz,ny,nx] float32}.  Returns the ordered dict the operator calls dic_vars."""
    attrs = {'z-levels': np.ascontiguousarray(zlev, dtype=np.float32), 'proj_info': dict(proj_info),
             'resolution': list(resolution), 'time': time}
    return {k: ModelVariable(k, np.ascontiguousarray(v, dtype=np.float32), attrs) for k, v in fields.items()}


def save_npz(path, fields, zlev, proj_info, resolution, time=0):
    np.savez(path, zlev=zlev, resolution=np.asarray(resolution, dtype=np.float64), time=np.asarray(time),
             proj_keys=np.array(list(proj_info.keys())), proj_vals=np.array([float(v) for v in proj_info.values()]),
             names=np.array(list(fields.keys())), **{'var_' + k: v for k, v in fields.items()})


def load_npz(path):
    z = np.load(path, allow_pickle=False)
    proj = {str(k): float(v) for k, v in zip(z['proj_keys'], z['proj_vals'])}
    fields = {str(n): z['var_' + str(n)] for n in z['names']}
    return variables_from_arrays(fields, z['zlev'], proj, list(z['resolution']), time=z['time'].item())
