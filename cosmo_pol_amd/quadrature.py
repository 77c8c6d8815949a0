"""Antenna-pattern quadrature: which sub-beams (azimuth / elevation offsets from the beam axis) a gate is
integrated over and with which weights, for the integration schemes of interpolation/interpolation.py:66-236.

Everything here runs once per configuration on the host; the result is the generic (az_off, el_off, weight)
list the library takes through cpol_set_quadrature, so every scheme runs on the same kernels.

    scheme 1   Gauss-Hermite on a single Gaussian main lobe                (interpolation.py:67-85)
    scheme 2   Gauss-Hermite (radial) x Gauss-Legendre (angular) on a sum of Gaussians  (:87-115)
    scheme 3   Gauss-Legendre weighted with a measured antenna diagram     (:117-139)
    scheme 5   discrete Gautschi quadrature built on the antenna diagram   (:178-198, quadrature.py)

Schemes 4 (reads a file outside the repository) and 6 (SpectralToolbox sparse grids) are not available.
"""
import numpy as np


# ----------------------------------------------------------------------------------------------
# Gaussian quadrature for an arbitrary weight function (interpolation/quadrature.py:33-95)
# ----------------------------------------------------------------------------------------------
def discrete_gautschi(z, w, n_iter):
    """Recurrence coefficients (a_0..a_{n-1}, b_1..b_{n-1}) of the polynomials orthogonal with respect to the
    discrete measure {(z_i, w_i)} (Stieltjes procedure)."""
    p = np.full(z.shape, 1. / np.sqrt(len(z)))
    p_prev = np.zeros(z.shape)
    a = np.empty(n_iter)
    b = np.empty(n_iter)
    wz = z * w
    norm_prev = 1.
    for j in range(n_iter):
        norm = np.dot(w * p, p)
        a[j] = np.dot(wz * p, p) / norm
        b[j] = 0. if j == 0 else norm / norm_prev
        p, p_prev = (z - a[j]) * p - b[j] * p_prev, p
        norm_prev = norm
    return a, b[1:]


def get_points_and_weights(w_func=lambda x: np.ones(x.shape), left=-1.0, right=1.0, num_points=5, n=8192):
    """Nodes x_i and weights w_i with  int_left^right f(x) w(x) dx ~ sum_i w_i f(x_i)  (Golub-Welsch on the
    Jacobi matrix of the recurrence coefficients; w_func sampled at n mid-points)."""
    import scipy.linalg
    dx = (float(right) - left) / n
    z = np.linspace(left + 0.5 * dx, right - 0.5 * dx, n)
    w = dx * w_func(z)
    a, b = discrete_gautschi(z, w, num_points)
    off = np.sqrt(b)
    J = np.diag(a) + np.diag(off, k=-1) + np.diag(off, k=1)
    J[~np.isfinite(J)] = 0.
    points, v = scipy.linalg.eigh(J)
    order = points.argsort()
    return points[order], (v[0, :] ** 2 * w.sum())[order]


# ----------------------------------------------------------------------------------------------
# antenna diagram helpers (utilities/utilities.py:144-174, interpolation/antenna_fit.py)
# ----------------------------------------------------------------------------------------------
def polar2cartesian(r, t, grid, x, y, order=3):
    """Resample grid[theta, r] at the Cartesian points (x_i, y_j); returns [len(y), len(x)]."""
    from scipy.interpolate import interp1d
    from scipy.ndimage import map_coordinates
    X, Y = np.meshgrid(x, y)
    new_r = np.sqrt(X * X + Y * Y)
    new_t = np.arctan2(X, Y) + np.pi
    ir = interp1d(r, np.arange(len(r)), bounds_error=False)(new_r.ravel())
    it = interp1d(t, np.arange(len(t)))(new_t.ravel())
    ir[new_r.ravel() > np.max(r)] = len(r) - 1
    ir[new_r.ravel() < np.min(r)] = 0
    return map_coordinates(grid, np.array([it, ir]), order=order).reshape(new_r.shape)


def vector_1d_to_polar(angles, values, x, y):
    """A symmetric 1-D cut through the pattern (angles from -max to +max) turned into a 2-D pattern that
    interpolates between its positive and its negative half."""
    mid = int(np.floor(len(angles) / 2))
    r = angles[mid:]
    thet = [0, np.pi, 2 * np.pi]
    pol = np.zeros((3, len(r)))
    pol[0, :] = values[mid:]
    pol[1, :] = values[0:mid + 1][::-1]
    pol[2, :] = pol[0, :]
    return polar2cartesian(r, thet, pol, x, y)


def gaussian_sum(x, params):
    """dB pattern of a sum of Gaussians, params rows = (amplitude dB, centre, sigma) (antenna_fit.py:14-16)."""
    return 10 * np.log10(np.sum([10 ** (0.1 * p[0]) * np.exp(-(x - p[1]) ** 2 / (2 * p[2] ** 2)) for p in params], axis=0))


def optimize_gaussians(x, y, n_gaussians):
    """Fit n Gaussians to an antenna diagram y(x) in dB (antenna_fit.py:36-67): lobes = local maxima, the n
    strongest as start values with sigma 0.5, SLSQP on the rms dB error, main lobe pinned at 0 dB / 0 deg,
    sigma in [0.1, 2]."""
    from scipy.optimize import minimize
    from scipy.signal import argrelextrema
    peaks = argrelextrema(y, np.greater)
    a_lobes, mu_lobes = y[peaks], x[peaks]
    if 0 not in mu_lobes:
        mu_lobes = np.append(mu_lobes, 0)
        a_lobes = np.append(a_lobes, 0)
    params = np.column_stack((a_lobes, mu_lobes))
    params = np.flipud(params[params[:, 0].argsort()])
    sel = params[0:n_gaussians, :]
    n_gaussians = len(sel)
    p0 = np.column_stack((sel[:, 0], sel[:, 1], np.array([0.5] * n_gaussians)))
    bounds = []
    for i in range(n_gaussians):
        bounds += [[None, None], [None, None], [0.1, 2]]
    bounds[0] = [0, 0]
    bounds[1] = [0, 0]

    def obj(p):
        est = gaussian_sum(x, np.reshape(p, (len(p) // 3, 3)))
        return np.sqrt(np.sum((est - y) ** 2))

    res = minimize(obj, p0.ravel(), bounds=bounds, method='SLSQP')
    return np.reshape(res['x'], (n_gaussians, 3))


def read_antenna_diagram(path):
    """Comma-separated (angle in degrees, one-way gain in dB) -> angles, squared linear gain (:121-124)."""
    antenna = np.genfromtxt(path, delimiter=',')
    return antenna[:, 0], (10 ** (0.1 * antenna[:, 1])) ** 2


# ----------------------------------------------------------------------------------------------
# the sub-beam set of a configuration
# ----------------------------------------------------------------------------------------------
def _threshold(weights, sum_weights, weight_threshold):
    """interpolation.py:226-234: smallest weight kept and the sum of the kept weights."""
    ws = np.sort(np.array(weights).ravel())[::-1]
    wc = np.cumsum(ws / sum_weights)
    wc[-1] = 1.
    thr = ws[np.where(wc >= weight_threshold)[0][0]]
    return thr, np.sum(ws[ws >= thr])


def subbeam_set(conf):
    """(az_off, el_off, weight) arrays of the sub-beams of conf['integration'], in the order the reference
    builds its list of beams (the central one, index int(n / 2), carries the output geometry)."""
    integ = conf['integration']
    scheme = integ['scheme']
    bw = conf['radar']['3dB_beamwidth']
    if scheme == 1:
        nh, nv = int(integ['nh_GH']), int(integ['nv_GH'])
        sigma = bw / (2 * np.sqrt(2 * np.log(2)))
        ph, wh = np.polynomial.hermite.hermgauss(nh)
        pv, wv = np.polynomial.hermite.hermgauss(nv)
        ph, pv = ph * sigma, pv * sigma                      # no sqrt(2): as the reference
        w = np.outer(wh * sigma, wv * sigma)
        w *= np.abs(np.cos(pv))                              # degrees fed to cos: as the reference
    elif scheme == 2:
        nr, na = int(integ['nr_GH']), int(integ['na_GL'])
        ap = np.asarray(integ['antenna_params'], dtype=np.float64)
        a_db, mu, sigma = ap[:, 0], ap[:, 1], ap[:, 2]
        pts_ang, w_ang = np.polynomial.legendre.leggauss(na)
        pts_rad, w_rad = np.polynomial.hermite.hermgauss(nr)
        az, el, ww = [], [], []
        total = 0
        for i in range(nr):
            for j in range(len(sigma)):
                for k in range(na):
                    r = mu[j] + np.sqrt(2) * sigma[j] * pts_rad[i]
                    theta = np.pi * pts_ang[k] + np.pi
                    wt = np.pi * w_ang[k] * w_rad[i] * 10 ** (0.1 * a_db[j]) * np.sqrt(2) * sigma[j] * abs(r)
                    wt *= np.cos(r * np.sin(theta))
                    ww.append(wt)
                    total += wt
                    az.append(r * np.cos(theta))
                    el.append(r * np.sin(theta))
        thr, total = _threshold(ww, total, integ['weight_threshold'])
        keep = [i for i in range(len(ww)) if ww[i] >= thr]
        return (np.array([az[i] for i in keep]), np.array([el[i] for i in keep]),
                np.array([ww[i] / total for i in keep]))
    elif scheme == 3:
        nh, nv = int(integ['nh_GH']), int(integ['nv_GH'])
        angles, power_sq = read_antenna_diagram(integ['antenna_diagram'])
        bounds = np.max(angles)
        ph, wh = np.polynomial.legendre.leggauss(nh)
        pv, wv = np.polynomial.legendre.leggauss(nv)
        ph, pv = ph * bounds, pv * bounds
        w = vector_1d_to_polar(angles, power_sq, ph, pv).T * np.outer(wh, wv)
        w *= np.abs(np.cos(pv))
        w *= 2 * bounds
    elif scheme == 5:
        from scipy.interpolate import interp1d
        nh, nv = int(integ['nh_GH']), int(integ['nv_GH'])
        angles, power_sq = read_antenna_diagram(integ['antenna_diagram'])
        bounds = np.max(angles)
        fit = interp1d(angles, power_sq, fill_value=0)
        fit_w = interp1d(angles, power_sq * np.cos(np.abs(np.deg2rad(angles))), fill_value=0)
        ph, wh = get_points_and_weights(fit, -bounds, bounds, nh)
        pv, wv = get_points_and_weights(fit_w, -bounds, bounds, nv)
        w = np.outer(wh, wv)
    else:
        raise NotImplementedError("integration scheme %r: schemes 1, 2, 3 and 5 are available (4 reads a file that is "
                                  "not part of the repository, 6 needs SpectralToolbox)" % (scheme,))
    thr, total = _threshold(w, np.sum(w.ravel()), integ['weight_threshold'])
    broad = nh > 1 or nv > 1
    az, el, ww = [], [], []
    for i in range(len(ph)):
        for j in range(len(pv)):
            if w[i, j] >= thr or not broad:
                az.append(ph[i]); el.append(pv[j]); ww.append(w[i, j] / total)
    return np.array(az), np.array(el), np.array(ww)
