"""Host-side driver of one GPU context: turns the reference-style configuration
dictionary, lookup tables and model state into C-ABI uploads (cosmo_pol_amd._lib).
"""
import numpy as np

from . import _lib
from . import constants as cst
from . import hydrometeors
from . import quadrature


def earth_radius(latitude):
    """refraction/atm_refraction.py:18-21."""
    a = 6378.1370 * 1000
    b = 6356.7523 * 1000
    return np.sqrt(((a ** 2 * np.cos(latitude)) ** 2 + (b ** 2 * np.sin(latitude)) ** 2) /
                   ((a * np.cos(latitude)) ** 2 + (b * np.sin(latitude)) ** 2))


def radar_ranges(conf):
    r = conf['radar']
    return np.arange(r['radial_resolution'] / 2, r['range'], r['radial_resolution'])


def gauss_hermite_subbeams(conf):
    """Sub-beams of the configured integration scheme as (az_off, el_off, weight), see quadrature.subbeam_set."""
    return quadrature.subbeam_set(conf)


class Engine(object):
    def __init__(self, conf, device=0):
        self.ctx = _lib.Context(device)
        self.conf = None
        self.configure(conf)

    def configure(self, conf, doppler=None, sensitivity_cut=False):
        r = conf['radar']
        rr = radar_ranges(conf)
        wl = cst.C / (r['frequency'] * 1E09) * 1000
        dscheme = conf['doppler']['scheme'] if doppler is None else doppler
        # Scheme 4 is the reference's name for the spectrum with turbulence broadening (doppler_sz.py:36); the
        # option file's doppler.turbulence_correction asks for the same on top of scheme 3.
        self.turbulence = int(dscheme) == 4 or (int(dscheme) == 3 and bool(conf['doppler'].get('turbulence_correction', 0)))
        if int(dscheme) == 4:
            dscheme = 3
        sens = r['sensitivity']
        if not isinstance(sens, (list, tuple)):
            sens = [sens]
        mode = 0
        sens = [float(v) for v in sens]
        if sensitivity_cut:
            if len(sens) not in (1, 2, 3):
                # the reference prints this and returns the beams uncut (utilities.py:83-88)
                print('Sensitivity parameters are invalid, cannot cut at specified sensitivity')
            else:
                mode = len(sens)
            if mode == 3:
                # [sensitivity (dBm), gain (dBm), SNR (dB)]: threshold = sensitivity + RADAR_CONSTANT_DB + SNR
                # + 20 log10(r / 1000) (utilities.py:74-76, constants.py:79-82)
                pulse_width = 2 * r['radial_resolution']
                rc_db = 180 - 10 * np.log10(np.pi ** 3 * r['3dB_beamwidth'] ** 2 * cst.KW ** 2 * pulse_width / 1000) \
                    - 2 * sens[1] + 10 * np.log10(1024 * np.log(2) * (wl / 1000.) ** 2)
                sens = [float(sens[0] + rc_db + sens[2]), 0.]
        self.ctx.set_config(radar_lat=float(r['coords'][0]), radar_lon=float(r['coords'][1]),
                            radar_alt=float(r['coords'][2]), range_res=float(r['radial_resolution']),
                            earth_radius=float(earth_radius(r['coords'][1])), n_gates=len(rr),
                            wavelength_mm=float(wl), kw=cst.KW,
                            attenuation_correction=int(conf['attenuation']['correction']),
                            microphysics_scheme=int(conf['microphysics']['scheme']),
                            doppler_scheme=int(dscheme), sensitivity_mode=mode,
                            sensitivity=(float(sens[0]), float(sens[1]) if len(sens) > 1 else 0.))
        if int(dscheme) == 3:
            nvel = cst.C / (4 * 1E-6 * r['PRI'] * r['frequency'] * 1E09)
            vres = 2 * nvel / r['FFT_length']
            self.varray = np.arange(-nvel, nvel + vres, vres)                    # constants.py:73-75
            self.ctx.set_velocity_bins(self.varray)
        self.beamwidth = float(r.get('3dB_beamwidth', 1.0))
        self._apply_turbulence()
        self.conf = conf
        self.n_gates = len(rr)

    def _apply_turbulence(self):
        """Turbulence broadening needs the eddy dissipation rate among the model variables; without it the
        reference falls back to the plain spectrum with a message (doppler_sz.py:36-41)."""
        on = getattr(self, 'turbulence', False)
        names = getattr(self, 'model_names', None)
        if on and names is not None and 'EDR' not in names:
            print('No eddy dissipitation rate variable found in COSMO file, could not use doppler_scheme == 4, '
                  'using doppler_scheme == 3 instead')
            on = False
        self.ctx.set_turbulence(on, getattr(self, 'beamwidth', 1.0), cst.A)

    def set_gauss_hermite(self):
        az, el, w = gauss_hermite_subbeams(self.conf)
        self.ctx.set_quadrature(az, el, w)
        self.quad = (az, el, w)
        return az, el, w

    def set_refraction(self, N=None, ray_elevations=None):
        """Refraction scheme 2: solve the ray equation through the refractivity field N (a model variable) for
        every distinct total elevation of ray_elevations x the current sub-beams and upload the table; with
        N = None the library returns to the 4/3 model.  Returns the number of profiles."""
        if N is None:
            self.ctx.set_refraction_table()
            return 0
        from . import refraction
        coords = self.conf['radar']['coords']
        el_off = np.asarray(self.quad[1], dtype=np.float64)
        keys = np.unique((el_off[None, :] + np.asarray(ray_elevations, dtype=np.float64)[:, None]).ravel())
        rr = radar_ranges(self.conf)
        prof = refraction.refractive_index_profile(N, coords, self.conf['radar']['type'])
        S, H, E = zip(*[refraction.ref_ODE_s(rr, k, coords, N, profile=prof) for k in keys])
        self.ctx.set_refraction_table(keys, np.stack(S), np.stack(H), np.stack(E))
        return len(keys)

    def upload_luts(self, lut_sz):
        """lut_sz: {class letter: Lookup_table-like with axes e, t, d, sz}."""
        scheme = self.conf['microphysics']['scheme']
        for h in hydrometeors.hydrom_types(scheme):
            lut = lut_sz[h]
            ie, it, idd = lut.axes_names['e'], lut.axes_names['t'], lut.axes_names['d']
            tab = lut.float_table() if getattr(lut, 'int_transform', False) else np.asarray(lut.value_table)
            order = (ie, it, idd, lut.axes_names['sz'])
            if order != (0, 1, 2, 3):
                tab = np.transpose(tab, order)
            pol = hydrometeors.create_hydrometeor(h, '2mom' if scheme == 2 else '1mom')
            dop = hydrometeors.create_hydrometeor(h, scheme)      # doppler_sz.py:57 passes the integer
            self.ctx.upload_lut(h, lut.axes_limits[ie][0], lut.axes_step[ie], lut.axes_limits[it][0],
                                lut.axes_step[it], lut.axes[idd], tab, pol.psd_descriptor(), dop.doppler_descriptor(scheme))

    def upload_model(self, fields, zlev, proj_info, resolution):
        llc = (float(proj_info['Lo1']), float(proj_info['La1']))
        urc = (float(proj_info['Lo2']), float(proj_info['La2']))
        pole = (float(proj_info['Latitude_of_southern_pole']), float(proj_info['Longitude_of_southern_pole']))
        self.ctx.upload_model(fields, zlev, llc, urc, resolution, pole)
        self.model_names = list(fields.keys())
        self._apply_turbulence()
