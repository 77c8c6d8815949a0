"""get_doppler_velocity with the reference signature (doppler/doppler_sz.py:26).  Scheme 1 (PSD-weighted fall
speed, closed form), scheme 2 (RCS- and PSD-weighted, two extra columns of the PSD integration) and scheme 3
(full Doppler spectrum, one wavefront per gate; 1-moment input only, as the reference's get_refl handles three
classes) all run in the HIP kernels."""
import numpy as np

from . import cfg, runtime
from .beam import Beam
from .scattering_sz import _pack


def get_doppler_velocity(list_beams, lut_sz=0, device=0):
    scheme = cfg.CONFIG['doppler']['scheme']
    if scheme not in (1, 2, 3, 4):
        raise ValueError("doppler.scheme must be 1, 2, 3 or 4, got %r" % (scheme,))
    st = runtime.get_engine(device, doppler=scheme)
    if lut_sz:
        runtime.ensure_luts(st, lut_sz)
    len_beams = max(len(b.dist_profile) for b in list_beams)
    vals, elev, mask, w = _pack(list_beams, len_beams)
    az = np.array([b.GH_pt[0] for b in list_beams], dtype=np.float64)
    _, mask_out, rvel = st.eng.ctx.observables_from_beams(vals, elev, mask, w, az)
    if any(len(b.mask) != len_beams for b in list_beams):
        # sub-beams of unequal length (GPM): this function pads the masks with 0 where get_radar_observables pads
        # with 1 (doppler_sz.py:110 against scattering_sz.py:146)
        mask_out = np.zeros(len_beams)
        for b in list_beams:
            mask_out[:len(b.mask)] += b.mask
        mask_out /= len(list_beams)
        mask_out[np.logical_and(mask_out >= 0, mask_out < 1)] = 0
    c = list_beams[int(len(list_beams) / 2)]
    vals_out = {'RVEL': rvel}
    if scheme in (3, 4):
        vals_out['DSPECTRUM'] = st.eng.ctx.last_spectrum                 # doppler_sz.py:123
    return Beam(vals_out, mask_out, c.lats_profile, c.lons_profile, c.dist_profile, c.heights_profile)
