"""get_doppler_velocity with the reference signature (doppler/doppler_sz.py:26).  Scheme 1 (PSD-weighted fall
speed, closed form) and scheme 2 (RCS- and PSD-weighted, two extra columns of the PSD integration) run in the
HIP kernels; scheme 3 (full spectrum) is not part of the batched path yet and raises."""
import numpy as np

from . import cfg, runtime
from .beam import Beam
from .scattering_sz import _pack


def get_doppler_velocity(list_beams, lut_sz=0, device=0):
    scheme = cfg.CONFIG['doppler']['scheme']
    if scheme not in (1, 2):
        raise NotImplementedError("doppler.scheme %r: schemes 1 and 2 run on the GPU path; the reference's "
                                  "get_refl kernel is available as cosmo_pol_amd.doppler_c.get_refl" % (scheme,))
    st = runtime.get_engine(device, doppler=scheme)
    if lut_sz:
        runtime.ensure_luts(st, lut_sz)
    len_beams = max(len(b.dist_profile) for b in list_beams)
    vals, elev, mask, w = _pack(list_beams, len_beams)
    az = np.array([b.GH_pt[0] for b in list_beams], dtype=np.float64)
    _, mask_out, rvel = st.eng.ctx.observables_from_beams(vals, elev, mask, w, az)
    c = list_beams[int(len(list_beams) / 2)]
    return Beam({'RVEL': rvel}, mask_out, c.lats_profile, c.lons_profile, c.dist_profile, c.heights_profile)
