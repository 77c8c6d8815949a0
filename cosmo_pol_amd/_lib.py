"""ctypes binding of libcosmopol_hip.so (the C ABI declared in include/cosmopol.h).

This replaces the reference's SWIG modules interpolation_c / doppler_c
(interpolation/interpolation_c_wrap.c, doppler/doppler_c.i).  There is no CPU
fallback: if the shared library is missing or fails to load, importing a
function from here raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcosmopol_hip.so")

OK, ERR_ARG, ERR_HIP, ERR_STATE, ERR_DOMAIN, ERR_REFRACTION = 0, 1, 2, 3, 4, 5
N_OBS = 9
STATIC_MAX_POSITIONS = 65536
OBS_NAMES = ['ZH', 'ZDR', 'ZV', 'KDP', 'DELTA_HV', 'PHIDP', 'RHOHV', 'AH', 'AV']
VAR_SLOTS = {'U': 0, 'V': 1, 'W': 2, 'QR_v': 3, 'QS_v': 4, 'QG_v': 5, 'RHO': 6, 'T': 7,
             'QH_v': 8, 'QNH_v': 9, 'QNR_v': 10, 'QNS_v': 11, 'QNG_v': 12, 'EDR': 13}
V_COUNT = 14
CLASS_INDEX = {'R': 0, 'S': 1, 'G': 2, 'H': 3}

c_float_p = C.POINTER(C.c_float)
c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)


class CpolConfig(C.Structure):
    _fields_ = [('radar_lat', C.c_double), ('radar_lon', C.c_double), ('radar_alt', C.c_double),
                ('range_res', C.c_double), ('earth_radius', C.c_double), ('n_gates', C.c_int),
                ('wavelength_mm', C.c_double), ('kw', C.c_double),
                ('attenuation_correction', C.c_int), ('microphysics_scheme', C.c_int),
                ('doppler_scheme', C.c_int), ('sensitivity_mode', C.c_int),
                ('sensitivity', C.c_double * 2)]


class CpolPsd(C.Structure):
    _fields_ = [('psd_kind', C.c_int)] + [(n, C.c_double) for n in
               ('a', 'b', 'alpha', 'beta', 'mu', 'nu', 'lambda_factor', 'vel_factor', 'ntot_factor',
                'n0_fixed', 'x_min', 'x_max', 'd_min', 'd_max')] + [('q_var', C.c_int), ('qn_var', C.c_int), ('t_var', C.c_int)]


class CpolLut(C.Structure):
    _fields_ = [('n_e', C.c_int), ('n_t', C.c_int), ('n_d', C.c_int),
                ('e_min', C.c_float), ('e_step', C.c_float), ('t_min', C.c_float), ('t_step', C.c_float),
                ('d_axis', c_float_p), ('table', c_float_p)]


class CpolOutputs(C.Structure):
    _fields_ = [('obs', c_float_p), ('rvel', c_double_p), ('mask', c_double_p), ('lat', c_double_p),
                ('lon', c_double_p), ('dist', c_float_p), ('height', c_float_p), ('ray_status', c_int_p),
                ('dspectrum', c_double_p), ('diag', c_double_p)]


# every symbol include/cosmopol.h declares
EXPORTS = ['cpol_create', 'cpol_destroy', 'cpol_last_error', 'cpol_stream', 'cpol_synchronize',
           'cpol_timing_enable', 'cpol_timing_read', 'cpol_timing_work', 'cpol_timing_forms', 'cpol_set_term_cutoff', 'cpol_get_term_cutoff', 'cpol_set_column_combination', 'cpol_get_column_combination', 'cpol_model_range_certified', 'cpol_set_batch_units', 'cpol_set_host_batching', 'cpol_set_config', 'cpol_set_velocity_bins', 'cpol_set_turbulence', 'cpol_set_refraction_table', 'cpol_upload_model', 'cpol_upload_lut', 'cpol_set_quadrature',
           'cpol_run_rays', 'cpol_run_rays_async', 'cpol_wait_async', 'cpol_run_rays_dev', 'cpol_run_rays_static', 'cpol_geolocate', 'cpol_interp_points',
           'cpol_observables_from_beams', 'get_all_radar_pts', 'get_refl']

_lib = None


def load():
    """Load the shared library (raises OSError if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    # A process that also uses PyTorch-ROCm must bring torch's HIP runtime up before this library initialises the
    # system's one (the other way round torch finds no GPU afterwards): do that for a caller who has imported torch.
    import sys
    torch = sys.modules.get('torch')
    if torch is not None and hasattr(torch, 'cuda') and torch.cuda.is_available():
        torch.cuda.init()
    lib = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    lib.cpol_create.argtypes = [C.c_int, C.POINTER(vp)]
    lib.cpol_destroy.argtypes = [vp]
    lib.cpol_last_error.argtypes = [vp]
    lib.cpol_last_error.restype = C.c_char_p
    lib.cpol_stream.argtypes = [vp]
    lib.cpol_stream.restype = vp
    lib.cpol_synchronize.argtypes = [vp]
    lib.cpol_timing_enable.argtypes = [vp, C.c_int]
    lib.cpol_timing_read.argtypes = [vp, c_double_p, C.POINTER(C.c_long), c_double_p, c_double_p]
    lib.cpol_timing_work.argtypes = [vp, c_double_p, c_double_p, c_double_p]
    lib.cpol_timing_forms.argtypes = [vp, c_double_p]
    lib.cpol_set_batch_units.argtypes = [vp, C.c_long]
    lib.cpol_set_host_batching.argtypes = [vp, C.c_int, C.c_double]
    lib.cpol_set_term_cutoff.argtypes = [vp, C.c_int]
    lib.cpol_get_term_cutoff.argtypes = [vp, c_float_p]
    lib.cpol_set_column_combination.argtypes = [vp, C.c_int]
    lib.cpol_get_column_combination.argtypes = [vp, C.POINTER(C.c_long)]
    lib.cpol_model_range_certified.argtypes = [vp]
    lib.cpol_set_config.argtypes = [vp, C.POINTER(CpolConfig)]
    lib.cpol_upload_model.argtypes = [vp, C.c_int, c_int_p, C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int,
                                      c_float_p, c_float_p, c_float_p, c_double_p]
    lib.cpol_upload_lut.argtypes = [vp, C.c_int, C.POINTER(CpolLut), C.POINTER(CpolPsd), C.POINTER(CpolPsd)]
    lib.cpol_set_quadrature.argtypes = [vp, C.c_int, c_double_p, c_double_p, c_double_p]
    lib.cpol_set_velocity_bins.argtypes = [vp, C.c_int, c_double_p]
    lib.cpol_set_turbulence.argtypes = [vp, C.c_int, C.c_double, C.c_double]
    lib.cpol_set_refraction_table.argtypes = [vp, C.c_int, c_double_p, c_float_p, c_float_p, c_float_p]
    lib.cpol_run_rays.argtypes = [vp, C.c_int, vp, vp, C.POINTER(CpolOutputs)]
    lib.cpol_run_rays_dev.argtypes = [vp, C.c_int, vp, vp, C.POINTER(CpolOutputs)]
    lib.cpol_run_rays_async.argtypes = [vp, C.c_int, vp, vp, C.POINTER(CpolOutputs)]
    lib.cpol_wait_async.argtypes = [vp, C.c_int]
    lib.cpol_run_rays_static.argtypes = [vp, C.c_int, vp, vp, C.POINTER(CpolOutputs)]
    lib.cpol_geolocate.argtypes = [vp, C.c_int, c_double_p, c_double_p, c_float_p, c_float_p, c_float_p,
                                   c_double_p, c_double_p, c_float_p, c_int_p]
    lib.cpol_interp_points.argtypes = [vp, C.c_int, c_float_p, c_float_p, c_float_p]
    lib.cpol_observables_from_beams.argtypes = [vp, C.c_int, C.c_int, c_float_p, c_float_p, c_double_p,
                                                c_double_p, c_double_p, c_float_p, c_double_p, c_double_p, c_double_p]
    fp, i = c_float_p, C.c_int
    lib.get_all_radar_pts.argtypes = [fp, i, fp, i, i, fp, i, fp, i, i, i, fp, i, i, i, fp, i, fp, i]
    lib.get_all_radar_pts.restype = fp
    lib.get_refl.argtypes = [fp, i, fp, i, i, fp, i, i, fp, i, i, fp, i, i, fp, i, fp, i, fp, i, fp, i, fp, i, fp, i]
    lib.get_refl.restype = fp
    _lib = lib
    return lib


def fptr(a):
    return a.ctypes.data_as(c_float_p)


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    return a.ctypes.data_as(c_int_p)


class CosmoPolError(RuntimeError):
    pass


class Context(object):
    """One library context = one GPU.  Thin, typed wrapper over the C ABI."""

    def __init__(self, device=0):
        self.lib = load()
        self.h = C.c_void_p()
        rc = self.lib.cpol_create(int(device), C.byref(self.h))
        if rc != OK:
            raise CosmoPolError("cpol_create(device=%d) failed with status %d (is a GPU visible?)" % (device, rc))
        self.n_gates = 0
        self.n_sub = 0
        self.var_names = []
        self.doppler = 0
        self.n_vel = 0

    def close(self):
        if self.h:
            self.lib.cpol_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc == OK:
            return
        msg = self.lib.cpol_last_error(self.h)
        msg = msg.decode() if msg else ''
        if rc == ERR_DOMAIN:
            raise IndexError('ERROR: RADAR DOMAIN IS NOT ENTIRELY CONTAINED IN COSMO SIMULATION DOMAIN: ABORTING')
        raise CosmoPolError("%s failed (status %d): %s" % (what, rc, msg))

    @property
    def stream(self):
        return self.lib.cpol_stream(self.h)

    def synchronize(self):
        self._check(self.lib.cpol_synchronize(self.h), 'cpol_synchronize')

    STAGES = ['geom', 'interp', 'prep_sort', 'integrate', 'observables', 'integrate_dop', 'spectrum']

    def timing_enable(self, on=True):
        self._check(self.lib.cpol_timing_enable(self.h, int(on)), 'cpol_timing_enable')

    def timing_read(self):
        ms = (C.c_double * len(self.STAGES))()
        nl = (C.c_long * len(self.STAGES))()
        vu, du = C.c_double(), C.c_double()
        self._check(self.lib.cpol_timing_read(self.h, ms, nl, C.byref(vu), C.byref(du)), 'cpol_timing_read')
        wk, wd, wo = C.c_double(), C.c_double(), C.c_double()
        self._check(self.lib.cpol_timing_work(self.h, C.byref(wk), C.byref(wd), C.byref(wo)), 'cpol_timing_work')
        cu = C.c_double()
        self._check(self.lib.cpol_timing_forms(self.h, C.byref(cu)), 'cpol_timing_forms')
        return {'ms': dict(zip(self.STAGES, list(ms))), 'launches': dict(zip(self.STAGES, list(nl))),
                'valid_units': vu.value, 'doppler_units': du.value, 'evaluated_unit_diameters': wk.value,
                'doppler_unit_diameters': wd.value, 'own_unit_diameters': wo.value, 'combined_units': cu.value}

    def set_term_cutoff(self, certified=True):
        self._check(self.lib.cpol_set_term_cutoff(self.h, int(bool(certified))), 'cpol_set_term_cutoff')

    def get_term_cutoff(self):
        """{class letter: exponent E}: terms with N(D) < 2^-E N(D_0) are left out of the PSD integration (150 = only
        exact float32 zeros)."""
        e = (C.c_float * 4)()
        self._check(self.lib.cpol_get_term_cutoff(self.h, e), 'cpol_get_term_cutoff')
        return {k: float(e[i]) for k, i in CLASS_INDEX.items() if e[i] > 0}

    def set_column_combination(self, on=True):
        """Integrate the units over the seven combinations of columns the observables are made of where that cannot
        change what the reference's rules about zero and underflowing components yield (see cosmopol.h); on=False: all
        units over all twelve columns."""
        self._check(self.lib.cpol_set_column_combination(self.h, int(bool(on))), 'cpol_set_column_combination')

    def get_column_combination(self):
        """{class letter: number of table nodes whose units may take the combined columns}; empty when none does."""
        f = (C.c_long * 8)()
        self._check(self.lib.cpol_get_column_combination(self.h, f), 'cpol_get_column_combination')
        return {k: int(f[i]) for k, i in CLASS_INDEX.items() if f[i] > 0}

    def model_range_certified(self):
        """True when the uploaded model's values let the vertical interpolation share one reciprocal per level pair
        (bit-identical to the division inside the certified range), False when every quotient is a full division."""
        rc = self.lib.cpol_model_range_certified(self.h)
        if rc < 0:
            self._check(rc, 'cpol_model_range_certified')
        return bool(rc)

    def set_batch_units(self, n):
        self._check(self.lib.cpol_set_batch_units(self.h, int(n)), 'cpol_set_batch_units')

    def set_host_batching(self, n_batches, last_fraction):
        """Batches per run_rays call and the share of the rays that the last, small one takes."""
        self._check(self.lib.cpol_set_host_batching(self.h, int(n_batches), float(last_fraction)), 'cpol_set_host_batching')

    # -- state ---------------------------------------------------------------------
    def set_config(self, **kw):
        c = CpolConfig()
        for k, v in kw.items():
            if k == 'sensitivity':
                c.sensitivity[0], c.sensitivity[1] = float(v[0]), float(v[1])
            else:
                setattr(c, k, v)
        self._check(self.lib.cpol_set_config(self.h, C.byref(c)), 'cpol_set_config')
        self.n_gates = int(c.n_gates)
        self.doppler = int(c.doppler_scheme)

    def upload_model(self, fields, zlev, llc, urc, res, pole):
        """fields: ordered {name: array-like [nz,ny,nx] f32 | int device pointer}."""
        names = list(fields.keys())
        ids = np.array([VAR_SLOTS[n] for n in names], dtype=np.int32)
        keep = []
        ptrs = (C.c_void_p * len(names))()
        shape = None
        for i, n in enumerate(names):
            a = fields[n]
            if isinstance(a, (tuple, list)):        # (device_pointer, shape)
                ptrs[i] = int(a[0])
                shape = tuple(a[1])
            else:
                a = np.ascontiguousarray(a, dtype=np.float32)
                keep.append(a)
                ptrs[i] = a.ctypes.data
                shape = a.shape
        if isinstance(zlev, (tuple, list)):
            zp = C.c_void_p(int(zlev[0]))
        else:
            zlev = np.ascontiguousarray(zlev, dtype=np.float32)
            zp = C.c_void_p(zlev.ctypes.data)
        nz, ny, nx = shape
        llc = np.asarray(llc, dtype=np.float32)
        urc = np.asarray(urc, dtype=np.float32)
        res = np.asarray(res, dtype=np.float32)
        pole = np.asarray(pole, dtype=np.float64)
        rc = self.lib.cpol_upload_model(self.h, len(names), iptr(ids), ptrs, zp, nz, ny, nx,
                                        fptr(llc), fptr(urc), fptr(res), dptr(pole))
        self._check(rc, 'cpol_upload_model')
        self.var_names = names

    def upload_lut(self, cls, e_min, e_step, t_min, t_step, d_axis, table, psd, psd_doppler=None):
        table = np.ascontiguousarray(table, dtype=np.float32)
        d_axis = np.ascontiguousarray(d_axis, dtype=np.float32)
        l = CpolLut(table.shape[0], table.shape[1], table.shape[2], float(e_min), float(e_step),
                    float(t_min), float(t_step), fptr(d_axis), fptr(table))
        p = CpolPsd(**psd)
        pd = CpolPsd(**(psd_doppler or psd))
        self._check(self.lib.cpol_upload_lut(self.h, CLASS_INDEX[cls], C.byref(l), C.byref(p), C.byref(pd)),
                    'cpol_upload_lut')

    def set_velocity_bins(self, varray):
        v = np.ascontiguousarray(varray, dtype=np.float64)
        self._check(self.lib.cpol_set_velocity_bins(self.h, len(v), dptr(v)), 'cpol_set_velocity_bins')
        self.n_vel = len(v)

    def set_turbulence(self, enable, beamwidth_deg=1.0, kolmogorov_a=1.6):
        self._check(self.lib.cpol_set_turbulence(self.h, int(bool(enable)), float(beamwidth_deg), float(kolmogorov_a)),
                    'cpol_set_turbulence')

    def set_refraction_table(self, keys=None, S=None, H=None, E=None):
        """keys: ascending total elevations; S, H, E: [len(keys), n_gates].  No arguments: back to the 4/3 model."""
        if keys is None or len(keys) == 0:
            self._check(self.lib.cpol_set_refraction_table(self.h, 0, None, None, None, None), 'cpol_set_refraction_table')
            return
        k = np.ascontiguousarray(keys, dtype=np.float64)
        S, H, E = (np.ascontiguousarray(a, dtype=np.float32) for a in (S, H, E))
        if not (S.shape == H.shape == E.shape == (len(k), self.n_gates)):
            raise ValueError('refraction profiles must be [%d, %d]' % (len(k), self.n_gates))
        self._check(self.lib.cpol_set_refraction_table(self.h, len(k), dptr(k), fptr(S), fptr(H), fptr(E)),
                    'cpol_set_refraction_table')

    def set_quadrature(self, az_off, el_off, weight):
        a = np.ascontiguousarray(az_off, dtype=np.float64)
        e = np.ascontiguousarray(el_off, dtype=np.float64)
        w = np.ascontiguousarray(weight, dtype=np.float64)
        self._check(self.lib.cpol_set_quadrature(self.h, len(a), dptr(a), dptr(e), dptr(w)), 'cpol_set_quadrature')
        self.n_sub = len(a)

    # -- hot path ------------------------------------------------------------------
    def run_rays(self, azimuth, elevation, raise_on_domain=True, check_status=True, diag=False):
        """check_status=False leaves every per-ray status (domain, refraction table) to the caller.  diag=True adds
        out['diag'] [n_vars, n_rays, n_gates]: the model variables (upload order, self.var_names) summed over the
        sub-beams as the reference's diagnostic mode does."""
        az = np.ascontiguousarray(azimuth, dtype=np.float64)
        el = np.ascontiguousarray(elevation, dtype=np.float64)
        R, G = len(az), self.n_gates
        out = {'obs': np.empty((N_OBS, R, G), dtype=np.float32), 'mask': np.empty((R, G)),
               'lat': np.empty((R, G)), 'lon': np.empty((R, G)),
               'dist': np.empty((R, G), dtype=np.float32), 'height': np.empty((R, G), dtype=np.float32),
               'ray_status': np.zeros(R, dtype=np.int32)}
        if self.doppler:
            out['rvel'] = np.empty((R, G))
        if self.doppler == 3:
            out['dspectrum'] = np.empty((R, G, self.n_vel))
        if diag:
            out['diag'] = np.empty((len(self.var_names), R, G))
        o = CpolOutputs(fptr(out['obs']), dptr(out['rvel']) if self.doppler else None, dptr(out['mask']),
                        dptr(out['lat']), dptr(out['lon']), fptr(out['dist']), fptr(out['height']),
                        iptr(out['ray_status']), dptr(out['dspectrum']) if self.doppler == 3 else None,
                        dptr(out['diag']) if diag else None)
        rc = self.lib.cpol_run_rays(self.h, R, az.ctypes.data, el.ctypes.data, C.byref(o))
        self._check(rc, 'cpol_run_rays')
        if not check_status:
            return out
        if raise_on_domain and (out['ray_status'] == ERR_DOMAIN).any():
            self._check(ERR_DOMAIN, 'cpol_run_rays')
        if (out['ray_status'] == ERR_REFRACTION).any():
            raise CosmoPolError('cpol_run_rays: the refraction table has no profile for the elevation of some rays')
        return out

    def run_rays_dev(self, az_ptr, el_ptr, n_rays, outputs=None):
        """Device-resident variant: az/el may be host or device addresses; returns the struct of
        device pointers (valid until the next call on this context)."""
        o = outputs if outputs is not None else CpolOutputs()
        rc = self.lib.cpol_run_rays_dev(self.h, int(n_rays), int(az_ptr), int(el_ptr), C.byref(o))
        self._check(rc, 'cpol_run_rays_dev')
        return o

    def run_rays_async(self, az_ptr, el_ptr, n_rays, host_outputs):
        """cpol_run_rays_async: queue one scan with its outputs delivered to the (page-locked) host buffers of
        host_outputs; synchronize() waits for the kernels and deliveries of every call so far."""
        rc = self.lib.cpol_run_rays_async(self.h, int(n_rays), int(az_ptr), int(el_ptr), C.byref(host_outputs))
        self._check(rc, 'cpol_run_rays_async')

    def wait_async(self, calls_back=0):
        """Wait for the delivery of the most recent run_rays_async call (0) or of the one before it (1)."""
        self._check(self.lib.cpol_wait_async(self.h, int(calls_back)), 'cpol_wait_async')

    def run_rays_static(self, az_dev_ptr, el_dev_ptr, n_rays, outputs=None):
        """cpol_run_rays_static: the allocation-free variant of run_rays_dev (device arrays in, every workspace from
        the library's static per-GPU array; at most STATIC_MAX_POSITIONS ray x sub-beam x gate positions)."""
        o = outputs if outputs is not None else CpolOutputs()
        rc = self.lib.cpol_run_rays_static(self.h, int(n_rays), int(az_dev_ptr), int(el_dev_ptr), C.byref(o))
        self._check(rc, 'cpol_run_rays_static')
        return o

    def geolocate(self, azimuth, elevation):
        az = np.ascontiguousarray(azimuth, dtype=np.float64)
        el = np.ascontiguousarray(elevation, dtype=np.float64)
        shp = (len(az), self.n_sub, self.n_gates)
        out = {'dist': np.empty(shp, dtype=np.float32), 'height': np.empty(shp, dtype=np.float32),
               'elev': np.empty(shp, dtype=np.float32), 'lat': np.empty(shp), 'lon': np.empty(shp),
               'coords': np.empty(shp + (2,), dtype=np.float32), 'ray_status': np.zeros(len(az), dtype=np.int32)}
        rc = self.lib.cpol_geolocate(self.h, len(az), dptr(az), dptr(el), fptr(out['dist']), fptr(out['height']),
                                     fptr(out['elev']), dptr(out['lat']), dptr(out['lon']), fptr(out['coords']),
                                     iptr(out['ray_status']))
        self._check(rc, 'cpol_geolocate')
        return out

    def interp_points(self, coords, heights):
        coords = np.ascontiguousarray(coords, dtype=np.float32)
        heights = np.ascontiguousarray(heights, dtype=np.float32)
        out = np.empty((len(self.var_names), len(heights)), dtype=np.float32)
        self._check(self.lib.cpol_interp_points(self.h, len(heights), fptr(coords), fptr(heights), fptr(out)),
                    'cpol_interp_points')
        return out

    def observables_from_beams(self, values, elev, mask, weight, az_sub=None):
        """values: {name: [B,G] f32}; returns (obs[9,G] f32, mask[G] f64, rvel[G] f64 | None)."""
        elev = np.ascontiguousarray(elev, dtype=np.float32)
        B, G = elev.shape
        vals = np.full((V_COUNT, B, G), np.nan, dtype=np.float32)
        for n, a in values.items():
            if n in VAR_SLOTS:
                vals[VAR_SLOTS[n]] = a
        mask = np.ascontiguousarray(mask, dtype=np.float64)
        weight = np.ascontiguousarray(weight, dtype=np.float64)
        az = np.ascontiguousarray(az_sub, dtype=np.float64) if az_sub is not None else None
        obs = np.empty((N_OBS, G), dtype=np.float32)
        mout = np.empty(G)
        rvel = np.empty(G) if self.doppler else None
        spec = np.empty((G, self.n_vel)) if self.doppler == 3 else None
        rc = self.lib.cpol_observables_from_beams(self.h, B, G, fptr(vals), fptr(elev), dptr(mask), dptr(weight),
                                                  dptr(az) if az is not None else None, fptr(obs), dptr(mout),
                                                  dptr(rvel) if rvel is not None else None,
                                                  dptr(spec) if spec is not None else None)
        self._check(rc, 'cpol_observables_from_beams')
        self.last_spectrum = spec
        return obs, mout, rvel
