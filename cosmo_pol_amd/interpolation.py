"""get_profiles_GH with the reference signature (interpolation/interpolation.py:53), served by the
GPU: refraction + geodesic + rotated pole (cpol_geolocate) and trilinear interpolation
(cpol_interp_points) of every model variable along every sub-beam of one ray."""
import numpy as np

from . import runtime
from .beam import Beam


def get_profiles_GH(dic_variables, azimuth, elevation, radar_range=0, N=0, list_refraction=0, device=0):
    st = runtime.get_engine(device)
    runtime.ensure_model(st, dic_variables)
    az_off, el_off, w = runtime.ensure_quadrature(st)
    ctx = st.eng.ctx
    from . import cfg
    if cfg.CONFIG['refraction']['scheme'] == 2 and N != 0:        # atm_refraction.py:32-33
        st.eng.set_refraction(N, [float(elevation)])
    else:
        st.eng.set_refraction(None)
    geo = ctx.geolocate([float(azimuth)], [float(elevation)])        # raises IndexError outside the domain
    if geo['ray_status'][0] != 0:
        raise IndexError('ERROR: RADAR DOMAIN IS NOT ENTIRELY CONTAINED IN COSMO SIMULATION DOMAIN: ABORTING')
    B, G = len(w), ctx.n_gates
    raw = ctx.interp_points(geo['coords'][0].reshape(B * G, 2), geo['height'][0].reshape(B * G))
    raw = raw.reshape(len(ctx.var_names), B, G)
    beams = []
    for b in range(B):
        first = raw[0, b]
        mask = np.zeros(G)
        mask[first == -9999] = -1          # above the model domain (interpolation.py:275)
        mask[np.isnan(first)] = 1          # below the terrain (:276)
        values = {}
        for k, name in enumerate(ctx.var_names):
            v = raw[k, b].copy()
            v[mask != 0] = np.nan
            values[name] = v
        beams.append(Beam(values, mask, geo['lat'][0, b].copy(), geo['lon'][0, b].copy(), geo['dist'][0, b].copy(),
                          geo['height'][0, b].copy(), geo['elev'][0, b].copy(),
                          [az_off[b] + azimuth, el_off[b] + elevation], w[b]))
    return beams


def integrate_GH_pts(list_GH_pts):
    """Weighted mean of the model variables over the sub-beams (interpolation.py:21-51), host side:
    diagnostic output only."""
    nb = len(list_GH_pts)
    sw = sum(b.GH_weight for b in list_GH_pts)
    out = {}
    for k in list_GH_pts[0].values:
        stack = np.stack([b.values[k] * b.GH_weight / sw for b in list_GH_pts])
        out[k] = np.nansum(stack, axis=0)
    mask = np.sum([b.mask for b in list_GH_pts], axis=0) / float(nb)
    mask[np.logical_and(mask >= 0, mask < 1)] = 0
    c = list_GH_pts[int(nb / 2)]
    return Beam(out, mask, c.lats_profile, c.lons_profile, c.dist_profile, c.heights_profile)
