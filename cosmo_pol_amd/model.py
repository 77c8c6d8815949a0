"""Model-state container standing where pycosmo's variable objects stand in the reference
(radar_operator.py:138: `.data`, `.attributes['z-levels' | 'proj_info' | 'resolution' | 'time']`).
GRIB ingest (pycosmo, which needs a GRIB decoder) is not available; states come from memory, from .npz files or
from the model's NetCDF output (load_netcdf, classic format through scipy)."""
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


# ---------------------------------------------------------------------------------------------------
# derived variables the operator works on (radar_operator.py:32-33: 'QR_v', 'QS_v', ..., 'RHO'; 'N' for
# refraction scheme 2).  The reference gets them from pycosmo, which is not part of its repository; these are
# the standard thermodynamic definitions and are not pinned against it.
# ---------------------------------------------------------------------------------------------------
R_DRY = 287.05          # J kg-1 K-1
R_VAPOUR = 461.51

MASS_SPECIES = ('QR', 'QS', 'QG', 'QH')
NUMBER_SPECIES = ('QNR', 'QNS', 'QNG', 'QNH')


def air_density(P, T, QV, condensate=0.):
    """Density of moist air [kg m-3] from pressure [Pa], temperature [K], specific humidity and the specific
    content of all condensate [kg kg-1]."""
    return P / (R_DRY * T * (1. + (R_VAPOUR / R_DRY - 1.) * QV - condensate))


def refractivity(P, T, QV):
    """Radio refractivity N = (n - 1) 1e6 (Smith-Weintraub): 77.6 / T (p + 4810 e / T), pressures in hPa."""
    e = QV * P / (0.622 + 0.378 * QV)
    return 77.6 / T * (P / 100. + 4810. * (e / 100.) / T)


def derive_variables(raw, with_refractivity=False):
    """raw: {name: [nz, ny, nx]} with the model's prognostic fields (P, T, QV, U, V, W, QR, QS, QG and optionally
    QC, QI, QH, QNR, QNS, QNG, QNH as mass / number per kg of air).  Returns the fields the operator interpolates:
    U, V, W, T, RHO and the contents per volume of air Q*_v = Q* RHO, QN*_v = QN* RHO [, N]."""
    f32 = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    P, T, QV = (np.asarray(raw[k], dtype=np.float64) for k in ('P', 'T', 'QV'))
    cond = sum(np.asarray(raw[k], dtype=np.float64) for k in ('QC', 'QI') + MASS_SPECIES if k in raw)
    rho = air_density(P, T, QV, cond)
    out = {k: f32(raw[k]) for k in ('U', 'V', 'W', 'T')}
    out['RHO'] = f32(rho)
    for k in MASS_SPECIES + NUMBER_SPECIES:
        if k in raw:
            out[k + '_v'] = f32(np.asarray(raw[k], dtype=np.float64) * rho)
    if with_refractivity:
        out['N'] = f32(refractivity(P, T, QV))
    return out


# ---------------------------------------------------------------------------------------------------
# NetCDF output of the model (what pycosmo.get_variables(..., assign_heights=True) delivers from GRIB in the
# reference, radar_operator.py:138): prognostic fields on (level, rlat, rlon), half-level heights HHL,
# the rotated pole as attributes of the grid-mapping variable.
# ---------------------------------------------------------------------------------------------------
PROGNOSTIC = ('P', 'T', 'QV', 'U', 'V', 'W', 'QR', 'QS', 'QG', 'QC', 'QI', 'QH', 'QNR', 'QNS', 'QNG', 'QNH', 'EDR')


def _to_full_levels(a, nz):
    """Fields given on the nz + 1 half levels (W, HHL) as the mean of the two half levels around a full level."""
    a = np.asarray(a, dtype=np.float64)
    return 0.5 * (a[:-1] + a[1:]) if a.shape[0] == nz + 1 else a


def load_netcdf(path, with_refractivity=False, time_index=0):
    """Read one time step of a COSMO NetCDF file (classic format) and return the operator's variable dictionary:
    U, V, W, T, RHO, Q*_v, QN*_v [, EDR, N] with heights, rotated-pole metadata and resolution attached.
    Needed in the file: rlon, rlat, a variable with grid_north_pole_latitude / _longitude attributes, HHL (half
    levels) or HFL (full levels), P, T, QV, U, V, W, QR, QS, QG; the other species are taken when present.
    Levels run from the model top to the ground, as in the model's output."""
    from scipy.io import netcdf_file
    with netcdf_file(path, 'r', mmap=False) as f:
        v = f.variables

        def get(name):
            a = np.array(v[name].data, dtype=np.float64)
            for att in ('scale_factor', 'add_offset'):
                if hasattr(v[name], att):
                    a = a * getattr(v[name], att) if att == 'scale_factor' else a + getattr(v[name], att)
            return a[time_index] if a.ndim == 4 else a

        rlon, rlat = np.array(v['rlon'].data, dtype=np.float64), np.array(v['rlat'].data, dtype=np.float64)
        pole = [x for x in v.values() if hasattr(x, 'grid_north_pole_latitude')]
        if not pole:
            raise ValueError('no grid-mapping variable with grid_north_pole_latitude in %s' % path)
        np_lat, np_lon = float(pole[0].grid_north_pole_latitude), float(pole[0].grid_north_pole_longitude)
        missing = [k for k in ('P', 'T', 'QV', 'U', 'V', 'W', 'QR', 'QS', 'QG') if k not in v]
        if missing:
            raise ValueError('Not all necessary variables could be found in file: missing %s' % missing)
        nz = get('T').shape[0]
        if 'HFL' in v:
            z = get('HFL')
        elif 'HHL' in v:
            z = _to_full_levels(get('HHL'), nz)
        else:
            raise ValueError('neither HHL nor HFL (level heights) in %s' % path)
        raw = {k: _to_full_levels(get(k), nz) for k in PROGNOSTIC if k in v}
    fields = derive_variables(raw, with_refractivity=with_refractivity)
    if 'EDR' in raw:
        fields['EDR'] = np.ascontiguousarray(raw['EDR'], dtype=np.float32)
    sp_lon = np_lon + 180.
    if sp_lon > 180.:
        sp_lon -= 360.
    proj = {'Lo1': float(rlon[0]), 'La1': float(rlat[0]), 'Lo2': float(rlon[-1]), 'La2': float(rlat[-1]),
            'Latitude_of_southern_pole': -np_lat, 'Longitude_of_southern_pole': sp_lon}
    res = [float(rlon[1] - rlon[0]), float(rlat[1] - rlat[0])]
    return variables_from_arrays(fields, z, proj, res, time=time_index)
