"""Container returned by RadarOperator.get_PPI / get_RHI: the data contract of the reference's PyradRadop
(radar/pyart_wrapper.py:107-232) without the PyART dependency -- ZH, ZV and ZDR converted to dB
(:159-161), rays of all sweeps stacked along axis 0 (:170-191), masked where NaN."""
import numpy as np

UNITS = {'ZH': 'dBZ', 'ZV': 'dBZ', 'ZDR': 'dB', 'KDP': 'deg/km', 'PHIDP': 'deg', 'RHOHV': '-', 'RVEL': 'm/s',
         'DELTA_HV': 'rad', 'AH': 'dB/km', 'AV': 'dB/km', 'DSPECTRUM': 'dBZ'}
DB_FIELDS = ('ZDR', 'ZV', 'ZH')        # the reference's list also names 'DSpectrum', which matches no key: the
                                        # spectrum stays linear (pyart_wrapper.py:159)


class SimulatedRadar(object):
    def __init__(self, scan_type, elevations, azimuths, ranges, fields_linear, mask, lat, lon, height,
                 nsweeps, rays_per_sweep, pos_time=None):
        self.scan_type = scan_type
        self.nsweeps, self.rays_per_sweep = nsweeps, rays_per_sweep
        self.nrays = len(azimuths)
        self.ngates = len(ranges)
        self.range = {'data': np.asarray(ranges), 'units': 'm'}
        self.azimuth = {'data': np.asarray(azimuths), 'units': 'degrees'}
        self.elevation = {'data': np.asarray(elevations), 'units': 'degrees'}
        self.sweep_start_ray_index = {'data': np.arange(nsweeps) * rays_per_sweep}
        self.sweep_end_ray_index = {'data': (np.arange(nsweeps) + 1) * rays_per_sweep - 1}
        self.gate_latitude = {'data': lat}
        self.gate_longitude = {'data': lon}
        self.gate_altitude = {'data': height}
        self.mask = mask
        self.pos_time = pos_time or {}
        self.fields = {}
        for k, v in fields_linear.items():
            data = np.array(v, copy=True)
            if k in DB_FIELDS:
                with np.errstate(all='ignore'):
                    data = 10 * np.log10(data)
            self.fields[k] = {'data': np.ma.masked_invalid(data), 'units': UNITS.get(k, '-')}

    def get_field(self, sweep, field_name):
        s = slice(sweep * self.rays_per_sweep, (sweep + 1) * self.rays_per_sweep)
        return self.fields[field_name]['data'][s]
