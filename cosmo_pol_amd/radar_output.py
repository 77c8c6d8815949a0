"""Container returned by RadarOperator.get_PPI / get_RHI / get_VPROF: the data contract of the reference's
PyradRadop (radar/pyart_wrapper.py:107-232) without the PyART dependency.

* fields: ZH, ZV and ZDR converted to dB with 0 -> NaN first (:159-161; the reference's list also names
  'DSpectrum', which matches no key, so the Doppler spectrum stays linear), every field a masked array (mask =
  NaN, :193-195) with long_name / units / valid_min / valid_max (:29-49), rays of all sweeps stacked along axis
  0 (:186-191), plus 'Latitude', 'Longitude' of every gate (:164-167) and 'range' tiled per ray (:223-224);
* time, range, azimuth, elevation, fixed_angle, sweep_number, sweep_mode, sweep_start_ray_index,
  sweep_end_ray_index, latitude, longitude, altitude, instrument_parameters['varray'] as PyART's Radar holds
  them (:196-232), get_field(sweep, name), get_slice, iter_*;
* to_netcdf(path): CF/Radial-like NetCDF-3 file (scipy.io), the closest stand-in for pyart.io.write_cfradial.
"""
import numpy as np

UNITS = {'ZH': 'dBZ', 'KDP': 'deg/km', 'PHIDP': 'deg', 'RHOHV': '-', 'ZDR': 'dB', 'RVEL': 'm/s', 'DSPECTRUM': 'dBZ',
         'ZV': 'dBZ', 'U': 'm/s', 'V': 'm/s', 'W': 'm/s', 'T': 'K', 'RHO': 'kg/m3', 'QR_v': 'kg/m3', 'QS_v': 'kg/m3',
         'QG_v': 'kg/m3', 'QH_v': 'kg/m3', 'DELTA_HV': 'rad', 'AH': 'dB/km', 'AV': 'dB/km'}
LABELS = {'ZH': 'Reflectivity', 'KDP': 'Specific diff. phase', 'PHIDP': 'Diff. phase', 'RHOHV': 'Copolar corr. coeff.',
          'ZDR': 'Diff. reflectivity', 'RVEL': 'Mean doppler velocity', 'DSPECTRUM': 'Doppler spectrum',
          'ZV': 'Vert. reflectivity', 'U': 'U-wind component', 'V': 'V-wind component', 'W': 'Vertical wind component',
          'T': 'Temperature', 'RHO': 'Air density', 'QR_v': 'Mass density of rain', 'QS_v': 'Mass density of snow',
          'QG_v': 'Mass density of graupel', 'QH_v': 'Mass density of hail', 'DELTA_HV': 'Backscatter diff. phase',
          'AH': 'Spec. attenuation (hor.)', 'AV': 'Spec. attenuation (vert.)'}
VMIN = {'ZH': 0., 'KDP': 0., 'PHIDP': 0., 'RHOHV': 0.6, 'ZDR': 0., 'RVEL': -25, 'DSPECTRUM': -50, 'ZV': 0., 'U': -30,
        'V': -30, 'W': -10, 'T': 200, 'RHO': 0.5, 'QR_v': 0., 'QS_v': 0., 'QG_v': 0., 'QH_v': 0.}
VMAX = {'ZH': 55, 'KDP': 1, 'PHIDP': 20, 'RHOHV': 1, 'ZDR': 2, 'RVEL': 25, 'DSPECTRUM': 30, 'ZV': 45, 'U': 30, 'V': 30,
        'W': 10, 'T': 300, 'RHO': 1.4, 'QR_v': 1E-3, 'QS_v': 1E-3, 'QG_v': 1E-3, 'QH_v': 1E-2}
DB_FIELDS = ('ZDR', 'ZV', 'ZH')


class SimulatedRadar(object):
    def __init__(self, scan_type, elevations, azimuths, ranges, fields_linear, mask, lat, lon, height,
                 nsweeps, rays_per_sweep, pos_time=None, fixed_angles=None, varray=None):
        self.scan_type = scan_type
        self.nsweeps, self.rays_per_sweep = int(nsweeps), int(rays_per_sweep)
        self.nrays = len(azimuths)
        self.ngates = len(ranges)
        self.range = {'data': np.asarray(ranges), 'units': 'm'}
        self.azimuth = {'data': np.asarray(azimuths), 'units': 'degrees'}
        self.elevation = {'data': np.asarray(elevations), 'units': 'degrees'}
        self.sweep_number = {'data': np.arange(self.nsweeps)}
        self.sweep_mode = {'data': [scan_type] * self.nsweeps}
        self.sweep_start_ray_index = {'data': np.arange(self.nsweeps) * self.rays_per_sweep}
        self.sweep_end_ray_index = {'data': (np.arange(self.nsweeps) + 1) * self.rays_per_sweep - 1}
        if fixed_angles is None:
            src = self.elevation['data'] if scan_type == 'ppi' else self.azimuth['data']
            fixed_angles = src[self.sweep_start_ray_index['data']] if len(src) else np.zeros(self.nsweeps)
        self.fixed_angle = {'data': np.asarray(fixed_angles, dtype=np.float64)}
        self.pos_time = pos_time or {}
        self.latitude = {'data': np.array(self.pos_time.get('latitude', np.nan))}
        self.longitude = {'data': np.array(self.pos_time.get('longitude', np.nan))}
        self.altitude = {'data': np.array(self.pos_time.get('altitude', np.nan))}
        self.time = {'data': np.zeros(self.rays_per_sweep), 'units': 'seconds since ' + str(self.pos_time.get('time', 0))}
        self.instrument_parameters = {'varray': {'data': varray}} if varray is not None else {}
        self.gate_latitude = {'data': lat}
        self.gate_longitude = {'data': lon}
        self.gate_altitude = {'data': height}
        self.mask = mask
        self.metadata = {}
        self.fields = {}
        for k, v in fields_linear.items():
            data = np.array(v, copy=True)
            if k in DB_FIELDS:
                with np.errstate(all='ignore'):
                    data[data == 0] = np.nan
                    data = 10 * np.log10(data)
            f = {'data': np.ma.array(data, mask=np.isnan(data))}
            for name, table in (('long_name', LABELS), ('units', UNITS), ('valid_min', VMIN), ('valid_max', VMAX)):
                if k in table:
                    f[name] = table[k]
            self.fields[k] = f
        if 'AH' in self.fields and 'AV' in self.fields and 'ADP' not in self.fields:
            # convenience field, not in the reference's output: differential specific attenuation
            adp = self.fields['AH']['data'] - self.fields['AV']['data']
            self.fields['ADP'] = {'data': adp, 'long_name': 'Diff. specific attenuation', 'units': 'dB/km'}
        for k, v in (('Latitude', lat), ('Longitude', lon)):
            if v is not None:
                self.fields[k] = {'data': np.ma.array(v, mask=np.isnan(v)), 'units': ['degrees']}
        self.fields['range'] = {'data': np.tile(self.range['data'], (self.nrays, 1))}

    # -- PyART Radar accessors ----------------------------------------------------------------
    def get_slice(self, sweep):
        return slice(int(self.sweep_start_ray_index['data'][sweep]), int(self.sweep_end_ray_index['data'][sweep]) + 1)

    def get_field(self, sweep, field_name):
        data = self.fields[field_name]['data'][self.get_slice(sweep)]
        return np.squeeze(data) if self.scan_type == 'vprof' else data

    def get_azimuth(self, sweep):
        return self.azimuth['data'][self.get_slice(sweep)]

    def get_elevation(self, sweep):
        return self.elevation['data'][self.get_slice(sweep)]

    def iter_slice(self):
        return (self.get_slice(s) for s in range(self.nsweeps))

    def iter_field(self, field_name):
        return (self.fields[field_name]['data'][s] for s in self.iter_slice())

    # -- file output --------------------------------------------------------------------------
    def to_netcdf(self, path, fill_value=-9999.):
        """Write a CF/Radial-like NetCDF-3 file: dimensions time (rays), range, sweep [, velocity]; coordinate
        and sweep variables under their CF/Radial names; every field as float32 with _FillValue where masked."""
        from scipy.io import netcdf_file
        with netcdf_file(path, 'w', version=2) as nc:
            nc.Conventions = 'CF/Radial instrument_parameters'
            nc.source = 'cosmo_pol_amd forward operator (simulated radar)'
            nc.scan_type = str(self.scan_type)
            nc.createDimension('time', self.nrays)
            nc.createDimension('range', self.ngates)
            nc.createDimension('sweep', self.nsweeps)

            def put(name, dims, data, dtype='f8', **attrs):
                v = nc.createVariable(name, dtype, dims)
                v[...] = np.asarray(data)
                for a, val in attrs.items():
                    setattr(v, a, val)
                return v

            put('time', ('time',), np.zeros(self.nrays), units=self.time['units'])
            put('range', ('range',), self.range['data'], units='meters')
            put('azimuth', ('time',), self.azimuth['data'], units='degrees')
            put('elevation', ('time',), self.elevation['data'], units='degrees')
            put('sweep_number', ('sweep',), self.sweep_number['data'], dtype='i4')
            put('fixed_angle', ('sweep',), self.fixed_angle['data'], units='degrees')
            put('sweep_start_ray_index', ('sweep',), self.sweep_start_ray_index['data'], dtype='i4')
            put('sweep_end_ray_index', ('sweep',), self.sweep_end_ray_index['data'], dtype='i4')
            for n in ('latitude', 'longitude', 'altitude'):
                put(n, (), float(getattr(self, n)['data']))
            if 'varray' in self.instrument_parameters:
                va = np.asarray(self.instrument_parameters['varray']['data'])
                nc.createDimension('velocity', len(va))
                put('varray', ('velocity',), va, units='m/s')
            for k, f in self.fields.items():
                data = np.ma.filled(f['data'].astype(np.float32), np.float32(fill_value))
                dims = ('time', 'range') if data.ndim == 2 else ('time', 'range', 'velocity')
                attrs = {'_FillValue': np.float32(fill_value)}
                for a in ('long_name', 'valid_min', 'valid_max'):
                    if a in f:
                        attrs[a] = f[a]
                if 'units' in f:
                    attrs['units'] = f['units'] if isinstance(f['units'], str) else str(f['units'][0])
                put(k, dims, data, dtype='f4', **attrs)
