"""Drop-in for the reference's SWIG module interpolation_c (interpolation/interpolation_c.py,
interpolation_c_wrap.c:3629-3843): same call, same (pointer, array) return pair, served by the GPU."""
import numpy as np

from . import _lib


def get_all_radar_pts(n, coords_rad_pts, radar_heights, model_data, model_heights, llc_cosmo, res_cosmo):
    lib = _lib.load()
    f = lambda a: np.ascontiguousarray(a, dtype=np.float32)
    coords, H, data, zlev, llc, res = map(f, (coords_rad_pts, radar_heights, model_data, model_heights,
                                              llc_cosmo, res_cosmo))
    out = np.zeros(int(n), dtype=np.float32)
    p = _lib.fptr
    r = lib.get_all_radar_pts(p(out), int(n), p(coords), coords.shape[0], coords.shape[1], p(H), len(H),
                              p(data), data.shape[0], data.shape[1], data.shape[2],
                              p(zlev), zlev.shape[0], zlev.shape[1], zlev.shape[2], p(llc), len(llc), p(res), len(res))
    if not r:
        raise TypeError("get_all_radar_pts: bad arguments or no GPU")
    return (None, out)
