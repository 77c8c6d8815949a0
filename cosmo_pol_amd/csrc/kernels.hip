// HIP kernels of the cosmo_pol hot path for gfx950 (MI355X, wave64).
//
// Built with -ffp-contract=off: the geometry and interpolation kernels must reproduce
// the reference's float32 / float64 operation order (no implicit FMA); fused operations
// are written explicitly (fmaf / MFMA) where they are wanted.
#include "common.h"
#include <math.h>

#define DEG (M_PI / 180.0)
#define RAD2DEG (180.0 / M_PI)

// =============================================================================
// model upload: [nz][ncell] -> [ncell][nz] with an output stride (variable interleave)
// =============================================================================
__global__ void k_transpose(const float *__restrict__ src, float *__restrict__ dst, int nz, long ncell,
                            int stride, int offset)
{
    __shared__ float tile[32][33];
    long c0 = (long)blockIdx.x * 32;
    int k0 = blockIdx.y * 32;
    int tx = threadIdx.x, ty = threadIdx.y;
    for (int j = ty; j < 32; j += 8) {
        int k = k0 + j;
        long c = c0 + tx;
        if (k < nz && c < ncell) tile[j][tx] = src[(long)k * ncell + c];
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        long c = c0 + j;
        int k = k0 + tx;
        if (k < nz && c < ncell) dst[(c * nz + k) * stride + offset] = tile[tx][j];
    }
}

void launch_transpose_levels(hipStream_t s, const float *src, float *dst, int nz, long ncell, int stride, int offset)
{
    dim3 grid((unsigned)((ncell + 31) / 32), (unsigned)((nz + 31) / 32));
    hipLaunchKernelGGL(k_transpose, grid, dim3(32, 8), 0, s, src, dst, nz, ncell, stride, offset);
}

// =============================================================================
// geometry: refraction (atm_refraction.py:110-123), WGS84 direct geodesic (what
// pyproj Geod.fwd does at interpolation.py:322-325; Karney 2013 series, order 6),
// rotated-pole transform (interpolation.py:355).
// =============================================================================
#define LINE_N CPOL_LINE_N

__device__ inline double polyv(int n, const double *p, double x)
{
    double y = *p++;
    while (--n >= 0) y = y * x + *p++;
    return y;
}

__device__ inline double sin_cos_series(double sinx, double cosx, const double *c, int n)
{   // Clenshaw sum of c[1..n] sin(2 k x); c[0] unused
    c += (n + 1);
    double ar = 2 * (cosx - sinx) * (cosx + sinx);
    double y0 = (n & 1) ? *--c : 0, y1 = 0;
    n /= 2;
    while (n--) {
        y1 = ar * y0 - y1 + *--c;
        y0 = ar * y1 - y0 + *--c;
    }
    return 2 * sinx * cosx * y0;
}

__device__ inline double ang_round(double x)
{
    const double z = 1 / 16.0;
    if (x == 0) return 0;
    double y = fabs(x);
    y = y < z ? z - (z - y) : y;
    return x < 0 ? -y : y;
}

__device__ inline double ang_normalize(double x)
{
    if (fabs(x) < 180.) return x;          // remainder(x, 360) is x itself there
    x = remainder(x, 360.0);
    return x != -180 ? x : 180;
}

// sin, cos, asin and atan of a small argument by their Taylor series: below SMALL_ANGLE the first term left out is
// under 2^-55 of the result, so the value is the correctly rounded one up to the last bit, as with the
// library routine, without its range reduction.  Every gate of a ground radar (ranges of a few hundred km against
// the earth's radius) takes this path; a larger argument goes to the library.
#define SMALL_ANGLE 0.1
__device__ inline void sincos_small(double x, double *s, double *c)
{
    if (fabs(x) < SMALL_ANGLE) {
        const double x2 = x * x;
        *s = x * (1. + x2 * (-1. / 6. + x2 * (1. / 120. + x2 * (-1. / 5040. + x2 * (1. / 362880. - x2 * (1. / 39916800.))))));
        *c = 1. + x2 * (-0.5 + x2 * (1. / 24. + x2 * (-1. / 720. + x2 * (1. / 40320. - x2 * (1. / 3628800.)))));
    } else sincos(x, s, c);
}

__device__ inline double asin_series(double x)
{
    const double x2 = x * x;
    return x * (1. + x2 * (1. / 6. + x2 * (3. / 40. + x2 * (15. / 336. + x2 * (105. / 3456. + x2 * (945. / 42240. +
               x2 * (10395. / 599040. + x2 * (135135. / 9676800.))))))));
}

__device__ inline double asin_small(double x) { return fabs(x) < SMALL_ANGLE ? asin_series(x) : asin(x); }

__device__ inline double atan_series(double x)
{
    const double x2 = x * x;
    return x * (1. + x2 * (-1. / 3. + x2 * (1. / 5. + x2 * (-1. / 7. + x2 * (1. / 9. + x2 * (-1. / 11. + x2 * (1. / 13. -
               x2 * (1. / 15.))))))));
}

__device__ inline double atan_small(double x) { return fabs(x) < SMALL_ANGLE ? atan_series(x) : atan(x); }

// 1 / d and 1 / sqrt(d) for a positive, normal, finite d: the hardware's approximation and two Newton steps, good
// to an ulp or two of float64.  The consumers round their results to float32 or tolerate 1e-12 degrees.
__device__ inline double rcp_fast(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.), r, r);
    return __builtin_fma(__builtin_fma(-d, r, 1.), r, r);
}

__device__ inline double rsqrt_fast(double d)
{
    double y = __builtin_amdgcn_rsq(d);
    y = y * __builtin_fma(-0.5 * d * y, y, 1.5);
    return y * __builtin_fma(-0.5 * d * y, y, 1.5);
}

// atan2(y, x) for the small angles of a ground radar's neighbourhood (x > 0, |y| < SMALL_ANGLE x): the series of
// the quotient; the library routine otherwise.
__device__ inline double atan2_small(double y, double x)
{
    if (x > 0. && fabs(y) < SMALL_ANGLE * x) return atan_series(y * rcp_fast(x));
    return atan2(y, x);
}

__device__ inline void sincosd(double x, double *sx, double *cx)
{
    int q;
    double r = remquo(x, 90.0, &q);
    r *= DEG;
    double s = sin(r), c = cos(r);
    switch ((unsigned)q & 3U) {
    case 0U: *sx = s;  *cx = c;  break;
    case 1U: *sx = c;  *cx = -s; break;
    case 2U: *sx = -s; *cx = -c; break;
    default: *sx = -c; *cx = s;  break;
    }
    if (x != 0) { *sx += 0.0; *cx += 0.0; }
}

__device__ inline double atan2d(double y, double x)
{
    int q = 0;
    if (fabs(y) > fabs(x)) { double t = x; x = y; y = t; q = 2; }
    if (x < 0) { x = -x; ++q; }
    double ang = atan2(y, x) * RAD2DEG;
    switch (q) {
    case 1: ang = (y >= 0 ? 180 : -180) - ang; break;
    case 2: ang = 90 - ang; break;
    case 3: ang = -90 + ang; break;
    }
    return ang;
}

__constant__ double c_C1[18] = {-1, 6, -16, 32, -9, 64, -128, 2048, 9, -16, 768, 3, -5, 512, -7, 1280, -7, 2048};
__constant__ double c_C1p[18] = {205, -432, 768, 1536, 4005, -4736, 3840, 12288, -225, 116, 384,
                                 -7173, 2695, 7680, 3467, 7680, 38081, 61440};
__constant__ double c_A3[18] = {-3, 128, -2, -3, 64, -1, -3, -1, 16, 3, -1, -2, 8, 1, -1, 2, 1, 1};
__constant__ double c_C3[45] = {3, 128, 2, 5, 128, -1, 3, 3, 64, -1, 0, 1, 8, -1, 1, 4, 5, 256, 1, 3, 128,
                                -3, -2, 3, 64, 1, -3, 2, 32, 7, 512, -10, 9, 384, 5, -9, 5, 192, 7, 512,
                                -14, 7, 512, 21, 2560};

// One thread per (ray, sub-beam): constants of the geodesic line through the radar.
__global__ void k_geom_line(GeomParams gp, int n_rays, const double *__restrict__ az, const double *__restrict__ el,
                            const double *__restrict__ quad, double *__restrict__ line, double *__restrict__ phi,
                            int *__restrict__ status)
{
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rays * gp.n_sub) return;
    int r = t / gp.n_sub, b = t % gp.n_sub;
    double azi1 = quad[b] + az[r];                       // pts_hor[i] + azimuth (interpolation.py:265)
    double el_b = quad[gp.n_sub + b] + el[r];            // pts_ver[j] + elevation
    double *L = line + (long)t * LINE_N;
    {   // azimuth of the sub-beam for the wind projection (doppler_sz.py:76)
        double pr = azi1 * (M_PI / 180.0);
        phi[2 * t] = sin(pr); phi[2 * t + 1] = cos(pr);
    }

    const double f = 1.0 / 298.257223563, f1 = 1 - f, e2 = f * (2 - f), ep2 = e2 / (f1 * f1);
    const double n = f / (2 - f);
    const double tiny = 1.4916681462400413e-154;         // sqrt(DBL_MIN)
    double A3x[6], C3x[15], C1a[7], C1pa[7], C3a[6];
    {
        int o = 0, k = 0;
        for (int j = 5; j >= 0; --j) {
            int m = 5 - j < j ? 5 - j : j;
            A3x[k++] = polyv(m, c_A3 + o, n) / c_A3[o + m + 1];
            o += m + 2;
        }
        o = 0; k = 0;
        for (int l = 1; l < 6; ++l)
            for (int j = 5; j >= l; --j) {
                int m = 5 - j < j ? 5 - j : j;
                C3x[k++] = polyv(m, c_C3 + o, n) / c_C3[o + m + 1];
                o += m + 2;
            }
    }
    double salp1, calp1, sbet1, cbet1;
    azi1 = ang_normalize(azi1);
    sincosd(ang_round(azi1), &salp1, &calp1);
    sincosd(ang_round(gp.lat0), &sbet1, &cbet1);
    sbet1 *= f1;
    { double h = hypot(sbet1, cbet1); sbet1 /= h; cbet1 /= h; }
    cbet1 = cbet1 > tiny ? cbet1 : tiny;
    double salp0 = salp1 * cbet1;
    double calp0 = hypot(calp1, salp1 * sbet1);
    double ssig1 = sbet1, somg1 = salp0 * sbet1;
    double csig1 = (sbet1 != 0 || calp1 != 0) ? cbet1 * calp1 : 1;
    double comg1 = csig1;
    { double h = hypot(ssig1, csig1); ssig1 /= h; csig1 /= h; }
    double k2 = calp0 * calp0 * ep2;
    double eps = k2 / (2 * (1 + sqrt(1 + k2)) + k2);
    double A1m1;
    {
        const double co[5] = {1, 4, 64, 0, 256};
        double tt = polyv(3, co, eps * eps) / co[4];
        A1m1 = (tt + eps) / (1 - eps);
    }
    {
        double eps2 = eps * eps, d = eps;
        int o = 0;
        for (int l = 1; l <= 6; ++l) {
            int m = (6 - l) / 2;
            C1a[l] = d * polyv(m, c_C1 + o, eps2) / c_C1[o + m + 1];
            C1pa[l] = d * polyv(m, c_C1p + o, eps2) / c_C1p[o + m + 1];
            o += m + 2;
            d *= eps;
        }
    }
    double B11 = sin_cos_series(ssig1, csig1, C1a, 6);
    double s = sin(B11), c = cos(B11);
    double stau1 = ssig1 * c + csig1 * s;
    double ctau1 = csig1 * c - ssig1 * s;
    double A3c = -f * salp0 * polyv(5, A3x, eps);
    {
        double mult = 1;
        int o = 0;
        for (int l = 1; l < 6; ++l) {
            int m = 6 - l - 1;
            mult *= eps;
            C3a[l] = mult * polyv(m, C3x + o, eps);
            o += m + 1;
        }
    }
    double B31 = sin_cos_series(ssig1, csig1, C3a, 5);
    L[0] = salp0; L[1] = calp0; L[2] = ssig1; L[3] = csig1; L[4] = somg1; L[5] = comg1;
    L[6] = k2; L[7] = A1m1; L[8] = B11; L[9] = stau1; L[10] = ctau1; L[11] = A3c; L[12] = B31;
    for (int l = 1; l <= 6; ++l) L[12 + l] = C1pa[l];
    for (int l = 1; l <= 5; ++l) L[18 + l] = C3a[l];
    L[24] = el_b;
    L[30] = 1. / (6378137.0 * f1 * (1 + A1m1));
    L[31] = ang_normalize(gp.lon0);
    {   // per-line trigonometry of k_geom_pts: elevation of the 4/3 model, rotated-pole latitude
        const double ela = el_b * M_PI / 180.;
        L[26] = sin(ela); L[27] = cos(ela);
        const double pollat = -gp.pole_lat_s;
        L[28] = sin(pollat * DEG); L[29] = cos(pollat * DEG);
        // reference point of the per-gate small-angle forms: the radar itself.  Longitude difference to the rotated
        // pole (its sine and cosine), geographic latitude as the geodesic solution returns it at distance zero, and
        // the pair (sin, f1 cos) of the reduced latitude it is the arc tangent of
        double pollon = gp.pole_lon_s - 180.0;
        if (pollon < -180.0) pollon += 360.0;
        const double dl0 = (L[31] - pollon) * DEG;
        L[32] = sin(dl0); L[33] = cos(dl0); L[37] = dl0;
        const double sb0 = calp0 * ssig1, cb0 = hypot(salp0, calp0 * csig1);
        L[34] = atan2d(sb0, f1 * cb0);
        const double hn = hypot(sb0, f1 * cb0);
        L[35] = sb0 / hn; L[36] = f1 * cb0 / hn;
    }
    if (gp.n_refr > 0) {
        // profile of this total elevation in the refraction table (exact key)
        int lo = 0, hi = gp.n_refr - 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (gp.refr_keys[mid] < el_b) lo = mid + 1; else hi = mid;
        }
        if (gp.refr_keys[lo] == el_b) L[25] = (double)lo;
        else { L[25] = -1.; status[r] = CPOL_ERR_REFRACTION; }
    }
}

// Position of gate g on the geodesic line L of its (ray, sub-beam): ground distance, height and elevation of the
// refraction model, geographic and rotated-pole coordinates.
struct GeoPoint {
    float  S, H, E;
    double lat, lon;
    float  rlat, rlon;
    bool   outside;        // outside the model domain (interpolation.py:359-362)
};

__device__ inline void geom_point(const GeomParams &gp, const ModelView &mv, const double *__restrict__ L, int g, GeoPoint &out)
{
    float S, H, E;
    if (gp.n_refr > 0) {
        // ---- refraction table solved on the host (scheme 2, atm_refraction.py:45-84) ----
        long p = (long)L[25];
        if (p < 0) p = 0;                                  // ray already flagged by k_geom_line
        S = gp.refr_S[p * gp.n_gates + g]; H = gp.refr_H[p * gp.n_gates + g]; E = gp.refr_E[p * gp.n_gates + g];
    } else {
        // ---- refraction, 4/3 earth model (atm_refraction.py:110-123) ----
        double rr = gp.res / 2 + g * gp.res;                 // np.arange(res/2, range, res)
        double ela = L[24] * M_PI / 180.;
        double sel = L[26], cel = L[27];
        const double ke = 4. / 3.;
        double keR = ke * gp.earth_radius;
        double Hd = sqrt(rr * rr + keR * keR + 2 * rr * ke * gp.earth_radius * sel) - keR + gp.alt;
        double Sd = keR * asin_small((rr * cel) * rcp_fast(keR + Hd));
        double Ed = ela + atan_small(rr * cel * rcp_fast(rr * sel + keR + gp.alt));
        S = (float)Sd; H = (float)Hd;
        E = (float)Ed * (float)(180.0 / M_PI);
    }
    out.S = S; out.H = H; out.E = E;

    // ---- geodesic position at distance S along the line ----
    const double f = 1.0 / 298.257223563, f1 = 1 - f;
    const double tiny = 1.4916681462400413e-154;
    double salp0 = L[0], calp0 = L[1], ssig1 = L[2], csig1 = L[3], somg1 = L[4], comg1 = L[5];
    double B11 = L[8], stau1 = L[9], ctau1 = L[10], A3c = L[11], B31 = L[12];
    double C1pa[7], C3a[6];
    for (int l = 1; l <= 6; ++l) C1pa[l] = L[12 + l];
    for (int l = 1; l <= 5; ++l) C3a[l] = L[18 + l];
    double tau12 = (double)S * L[30];                    // S / (b (1 + A1 - 1))
    double s, c;
    sincos_small(tau12, &s, &c);
    double B12 = -sin_cos_series(stau1 * c + ctau1 * s, ctau1 * c - stau1 * s, C1pa, 6);
    double sig12 = tau12 - (B12 - B11);
    // sin / cos of sig12 = tau12 + dlt from those of tau12: |dlt| < 3e-3 (series terms of order f), so five
    // Taylor terms are exact to 1e-19
    const double dlt = -(B12 - B11), dl2 = dlt * dlt;
    const double sdlt = dlt * (1. + dl2 * (-1. / 6. + dl2 * (1. / 120.)));
    const double cdlt = 1. + dl2 * (-0.5 + dl2 * (1. / 24. - dl2 * (1. / 720.)));
    double ssig12 = s * cdlt + c * sdlt, csig12 = c * cdlt - s * sdlt;
    double ssig2 = ssig1 * csig12 + csig1 * ssig12;
    double csig2 = csig1 * csig12 - ssig1 * ssig12;
    double sbet2 = calp0 * ssig2;
    const double cc2 = calp0 * csig2;
    double cbet2 = sqrt(salp0 * salp0 + cc2 * cc2);      // operands of order one: no scaling needed
    if (cbet2 == 0) cbet2 = csig2 = tiny;
    double somg2 = salp0 * ssig2, comg2 = csig2;
    double omg12 = atan2_small(somg2 * comg1 - comg2 * somg1, comg2 * comg1 + somg2 * somg1);
    double lam12 = omg12 + A3c * (sig12 + (sin_cos_series(ssig2, csig2, C3a, 5) - B31));
    double lon12 = lam12 * RAD2DEG;
    double lon = ang_normalize(L[31] + ang_normalize(lon12));      // L[31]: the radar's longitude, normalised
    // lat = atan2(sbet2, f1 cbet2) as the radar's own latitude L[34] = atan2(L[35], L[36]) plus the small angle between
    // the two directions (difference formula of the arc tangent), instead of a full-range atan2 per gate
    const double fc2 = f1 * cbet2;
    double lat = L[34] + atan2_small(sbet2 * L[36] - fc2 * L[35], fc2 * L[36] + sbet2 * L[35]) * RAD2DEG;
    out.lat = lat; out.lon = lon;

    // ---- rotated pole (north pole from the southern-pole metadata) ----
    double sp = L[28], cp = L[29];
    // lat = atan2(sbet2, f1 cbet2): its sine and cosine are that pair normalised
    const double rh = rsqrt_fast(sbet2 * sbet2 + fc2 * fc2);
    double sphi = sbet2 * rh, cphi = fc2 * rh, sdl, cdl;
    // longitude difference to the rotated pole, formed as the reference forms it; its sine and cosine from those of
    // the radar's own difference L[37] (L[32], L[33]) and the small angle in between, which the subtraction of the two
    // nearby float64 values gives exactly
    double pollon = gp.pole_lon_s - 180.0;
    if (pollon < -180.0) pollon += 360.0;
    const double dl = (lon - pollon) * DEG, ddl = dl - L[37];
    if (fabs(ddl) < SMALL_ANGLE) {
        double sl, cl;
        sincos_small(ddl, &sl, &cl);
        sdl = L[32] * cl + L[33] * sl; cdl = L[33] * cl - L[32] * sl;
    } else sincos(dl, &sdl, &cdl);
    double a0 = cp * cphi * cdl + sp * sphi;
    double a1 = -sdl * cphi;
    double a2 = -sp * cphi * cdl + cp * sphi;
    double rlat = asin_small(a0) * RAD2DEG, rlon = atan2_small(a1, a2) * RAD2DEG;
    out.rlat = (float)rlat;
    out.rlon = (float)rlon;
    // domain check of interpolation.py:359-362 (float64 coordinates against float32 corners)
    out.outside = rlon < (double)mv.llc0 || rlat < (double)mv.llc1 || rlon > (double)mv.urc0 || rlat > (double)mv.urc1;
}

// One thread per (ray, sub-beam, gate).
__global__ void k_geom_pts(GeomParams gp, long n_total, const double *__restrict__ line,
                           float *__restrict__ S_out, float *__restrict__ H_out, float *__restrict__ E_out,
                           double *__restrict__ lat_out, double *__restrict__ lon_out, float *__restrict__ coords,
                           ModelView mv, int *__restrict__ status)
{
    long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_total) return;
    int g = (int)(u % gp.n_gates);
    long rb = u / gp.n_gates;
    GeoPoint p;
    geom_point(gp, mv, line + rb * LINE_N, g, p);
    S_out[u] = p.S; H_out[u] = p.H; E_out[u] = p.E;
    lat_out[u] = p.lat; lon_out[u] = p.lon;
    coords[2 * u] = p.rlat;
    coords[2 * u + 1] = p.rlon;
    if (p.outside) status[(int)(rb / gp.n_sub)] = CPOL_ERR_DOMAIN;
}

void launch_geom(hipStream_t s, const GeomParams &gp, int n_rays, const double *az, const double *el,
                 const double *quad, double *line, float *S, float *H, float *E, double *lat, double *lon,
                 float *coords, double *phi, const ModelView &mv, int *status)
{
    int nl = n_rays * gp.n_sub;
    hipLaunchKernelGGL(k_geom_line, dim3((nl + 63) / 64), dim3(64), 0, s, gp, n_rays, az, el, quad, line, phi, status);
    long n_total = (long)nl * gp.n_gates;
    hipLaunchKernelGGL(k_geom_pts, dim3((unsigned)((n_total + 255) / 256)), dim3(256), 0, s, gp, n_total, line,
                       S, H, E, lat, lon, coords, mv, status);
}

// Eight heights of every column, at levels (k (nz - 1)) / 7, k = 0..7: the model top, the terrain and six levels
// in between, 32 bytes that start the level search of k_interp with one access instead of three bisection steps.
// Order inside the record: top (k = 0), terrain (k = 7), then k = 1..6 - a neighbour column's top / terrain test
// reads the first eight bytes.
__global__ void k_coarse_levels(const float *__restrict__ zcol, float *__restrict__ zc8, int nz, long ncell)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ncell * 8) return;
    const long c = t >> 3;
    const int i = (int)(t & 7);
    const int k = i == 0 ? 0 : i == 1 ? 7 : i - 1;
    zc8[t] = zcol[c * nz + (k * (nz - 1)) / 7];
}

void launch_coarse_levels(hipStream_t s, const float *zcol, float *zc8, int nz, long ncell)
{
    hipLaunchKernelGGL(k_coarse_levels, dim3((unsigned)((ncell * 8 + 255) / 256)), dim3(256), 0, s, zcol, zc8, nz, ncell);
}

// =============================================================================
// trilinear interpolation (interpolation_c.c:12-164), one thread per point, every
// variable in one pass.  The level search is the reference's loop, so the chosen
// levels are identical; the arithmetic is float32 in the reference's order.
// =============================================================================
__device__ inline int level_search(const float *__restrict__ col, int dim, float key)
{   // interpolation_c.c:108-135 on a contiguous, descending column
    int high = 0, low = dim - 1;
    if (key > col[0]) return -1;
    else if (key < col[dim - 1]) return -2;
    while (high < low) {
        int middle = (high + low) / 2;
        float v = col[middle];
        if (v == key) return middle;
        else if (low == middle || high == middle) return high;
        else if (v < key) low = middle;
        else if (v > key) high = middle;
        else return high;          // NaN height: the reference would not terminate
    }
    return high;
}

// The correctly rounded float32 quotient n / d from r, the refined reciprocal of d: the instruction sequence the
// compiler emits for a float32 division, without its scaling of the operands and its handling of special values
// - it returns the same bits wherever neither acts: d and n / d normal numbers, n zero or at least 2^-102, the
// exponents of n and d less than 96 apart, everything finite.  k_model_range certifies an uploaded model for that.
__device__ inline float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = fmaf(-d, r0, 1.f);
    return fmaf(e, r0, r0);
}
__device__ inline float div_refined(float n, float d, float r)
{
    float q = n * r;
    float e = fmaf(-d, q, n);
    q = fmaf(e, r, q);
    e = fmaf(-d, q, n);
    return fmaf(e, r, q);
}

// Sets *flag when the model leaves the range in which div_refined is the division: a variable that is not zero,
// finite and within [2^-78, 2^85] in magnitude (the difference of two such values is zero or at least 2^-101), or
// a level thickness outside [2^-10, 2^14] m.
__global__ void k_model_range(const float *__restrict__ dcol, size_t n_val, const float *__restrict__ zcol, int nz,
                              size_t n_z, unsigned *__restrict__ flag)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_val; i += stride) {
        const float a = fabsf(dcol[i]);
        bad = bad || !(a == 0.f || (a >= 0x1p-78f && a <= 0x1p85f));
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_z; i += stride) {
        if ((int)(i % nz) == nz - 1) continue;
        const float d = fabsf(zcol[i] - zcol[i + 1]);
        bad = bad || !(d >= 0x1p-10f && d <= 0x1p14f);
    }
    if (bad) *flag = 1u;
}

void launch_model_range(hipStream_t s, const float *dcol, size_t n_val, const float *zcol, int nz, size_t n_z, unsigned *flag)
{
    hipLaunchKernelGGL(k_model_range, dim3(2048), dim3(256), 0, s, dcol, n_val, zcol, nz, n_z, flag);
}

#ifndef FRONT_OCC
#define FRONT_OCC 6
#endif
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // 16- and 8-byte loads at 4-byte alignment
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
#define INTERP_XCD_RUN 72
#define INTERP_REC 17
#define INTERP_BATCH 4
// Interpolation of one point per thread by a workgroup of 256 threads; every thread of the workgroup calls it (it
// synchronises).  Returns the point's code (0 inside the model, 1 below the terrain or off the grid, -1 above the
// model top) and leaves the interpolated variable of slot v in s_rec[wave][lane][v].
__device__ inline int interp_point(const ModelView &mv, float (*__restrict__ s_rec)[64][INTERP_REC], bool live, float H,
                                   float rlat, float rlon)
{
    // Phase A (one thread per point): grid position, terrain test, level search.
    // Phase B (wave-cooperative): the 2 x nvp values of a neighbour column are contiguous in memory, so
    // sixteen lanes fetch a point's four columns with 16-byte loads instead of 8 nv scattered ones.
    // Phase C: results go through LDS so that every variable plane is written 64 points at a time.
    // Exchange between the phases, one record of 17 words per point (odd stride: no bank conflicts):
    //   [0..3] offsets of the four level pairs, [4..7] level thicknesses, [8..11] heights above the lower
    //   level, [12] x, [13] y, [14] code.
    // Phase B overwrites words 0..nvp-1 of a point's record with its results once it has read the record (the
    // lanes of a wave run in lockstep and a record is read only in the iteration that owns it, the offsets at
    // the top of its group of four), so the results need no area of their own: 17 KB per workgroup.
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float p0 = live ? (rlat - mv.llc1) / mv.res1 : -1.f;       // lat-like -> row
    float p1 = live ? (rlon - mv.llc0) / mv.res0 : -1.f;       // lon-like -> column
    int iy = (int)floorf(p0), ix = (int)floorf(p1);
    float x = fmodf(p0, 1.0f), y = fmodf(p1, 1.0f);
    float dx = 1.0f - x, dy = 1.0f - y;
    int code = 0;          // 0 ok, 1 below terrain (NaN), -1 above the domain (-9999)
    int c1[4];
    const int nz = mv.nz, nv = mv.nv;
    long cell[4];
    if (iy < 0 || ix < 0 || iy + 1 >= mv.ny || ix + 1 >= mv.nx || !(p0 == p0) || !(p1 == p1)) {
        code = 1;          // outside the grid: the reference raises before reaching the C code
    } else {
        cell[0] = (long)iy * mv.nx + ix; cell[1] = cell[0] + 1;
        cell[2] = cell[0] + mv.nx;       cell[3] = cell[2] + 1;
        // top, terrain and six levels in between of the four columns
        const float4 *q = (const float4 *)(mv.zc8 + cell[0] * 8);
        const float4 q0 = q[0], q1 = q[1];
        const float ck[8] = {q0.x, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q0.y};      // record order: top, terrain, k = 1..6
        float ztop[4], zbot[4];
        ztop[0] = ck[0]; zbot[0] = ck[7];
#pragma unroll
        for (int j = 1; j < 4; j++) {
            const float2 tb = *(const float2 *)(mv.zc8 + cell[j] * 8);
            ztop[j] = tb.x; zbot[j] = tb.y;
        }
        const float t0 = zbot[0], t1 = zbot[1], t2 = zbot[2], t3 = zbot[3];
        float topo = dx * dy * t0 + x * t2 * dy + dx * t1 * y + x * y * t3;
        if (!(topo < H)) code = 1;
        else {
            // The level search of interpolation_c.c:108-135 on a strictly descending column returns the last
            // level that is not below H (an exact hit on a level returns that level, a hit on the lowest one
            // the level above it), i.e. (number of levels >= H) - 1 capped at nz - 2; "above the top" and "below
            // the lowest level" are decided first, as there.
            const float *col[4];
            int res[4];
            bool above = false;
            bool below[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                col[j] = mv.zcol + cell[j] * nz;
                above = above || (H > ztop[j]);
                below[j] = H < zbot[j];
                res[j] = 0;
            }
            if (above) code = -1;
            else {
                int cnt[4] = {0, 0, 0, 0};
                // Column 0: the eight coarse heights bracket the count between two of their levels, and the levels
                // in between are searched four ways at a time (three independent probes per round: two rounds for
                // up to 15 levels) - three dependent memory accesses where a bisection takes log2(nz).
                // The levels of the three neighbours lie close to it (terrain-following coordinates), so their
                // counts are read off a window of four heights around column 0's count, and only a column whose
                // window does not bracket H is searched on its own.
                {
                    int kk = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) kk += ck[k] >= H ? 1 : 0;          // >= 1: H is not above the top
                    const int c_hi = kk < 8 ? (kk * (nz - 1)) / 7 : nz;             // first coarse level below H
                    int lo = ((kk - 1) * (nz - 1)) / 7 + 1;                         // levels before lo are >= H
                    int nn = kk < 8 ? c_hi - lo : 0;                                // undecided levels lo .. lo + nn - 1
                    while (__any(nn > 0)) {
                        const int m = (nn + 3) >> 2;
                        int c = 0;
#pragma unroll
                        for (int i = 1; i <= 3; i++) {
                            const int off = i * m - 1;
                            const bool in = nn > 0 && off < nn;
                            const float v = col[0][in ? lo + off : 0];
                            c += (in && v >= H) ? 1 : 0;
                        }
                        if (nn > 0) {
                            lo += c * m;
                            nn = c == 3 ? nn - 3 * m : min(m - 1, nn - c * m);     // no fourth probe bounds the last run
                        }
                    }
                    cnt[0] = kk < 8 ? lo : nz;
                }
                bool redo[4] = {false, false, false, false};
                if (nz >= 4) {
                    // four consecutive levels of each neighbour around column 0's count, one 16-byte load per column
                    // (4-byte aligned); on a descending column the count is exact when the level above the window is
                    // not below H and the level below it is
                    const int base = min(max(cnt[0] - 2, 0), nz - 4);
#pragma unroll
                    for (int j = 1; j < 4; j++) {
                        const f32x4u wv = *(const f32x4u *)(col[j] + base);
                        cnt[j] = base + (wv.x >= H ? 1 : 0) + (wv.y >= H ? 1 : 0) + (wv.z >= H ? 1 : 0) + (wv.w >= H ? 1 : 0);
                        redo[j] = !((base == 0 || wv.x >= H) && (base + 3 == nz - 1 || wv.w < H));
                    }
                } else redo[1] = redo[2] = redo[3] = true;
                if (redo[1] || redo[2] || redo[3]) {
                    int c2[4] = {0, 0, 0, 0};
                    for (int step = 1 << (31 - __clz(nz)); step > 0; step >>= 1) {
#pragma unroll
                        for (int j = 1; j < 4; j++) {
                            const bool in = c2[j] + step <= nz;
                            const float v = col[j][in ? c2[j] + step - 1 : 0];
                            if (in && v >= H) c2[j] += step;
                        }
                    }
#pragma unroll
                    for (int j = 1; j < 4; j++) if (redo[j]) cnt[j] = c2[j];
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int r = cnt[j] - 1;
                    if (r > nz - 2) r = nz - 2;
                    res[j] = below[j] ? -2 : r;
                }
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    int idx = res[j];
                    if (idx == -2) idx = nz - 3;
                    else if (idx == nz - 2) idx--;
                    c1[j] = idx;
                }
            }
        }
    }
    float *rec = s_rec[wv][lane];
    rec[12] = x; rec[13] = y; rec[14] = __int_as_float(code);
#pragma unroll
    for (int j = 0; j < 4; j++) {
        float fzj = 1.f, hzj = 0.f;
        int basej = 0;
        if (code == 0) {
            const f32x2u zp = *(const f32x2u *)(mv.zcol + cell[j] * nz + c1[j]);      // the pair in one 8-byte load
            const float za = zp.x, zb = zp.y;
            fzj = za - zb;
            hzj = H - zb;
            basej = (int)((cell[j] * nz + c1[j]) * (mv.nvp >> 2));    // offset of the level pair in 16-byte words (< 2^31)
        }
        rec[j] = __int_as_float(basej); rec[4 + j] = fzj; rec[8 + j] = hzj;
    }
    __syncthreads();
    {
        // 16 lanes per point: lane (j, k) of a point's group takes neighbour column j and the four variables
        // 4k .. 4k+3.  A level of a column is nvp (a multiple of four) consecutive floats, so a lane fetches its
        // variables of the two levels with two 16-byte loads and a column's level pair is read as two runs of
        // 4 nvp bytes; one wave instruction serves four points.
        const int grp = lane >> 4, j = (lane >> 2) & 3, k = lane & 3;
        const int nq = mv.nvp >> 2;
        const bool kact = k < nq;
        const float4 *dlane = (const float4 *)mv.dcol + (kact ? k : 0);
        for (int p4 = 0; p4 < 16; p4 += INTERP_BATCH) {
            // issue the loads of INTERP_BATCH groups of four points before using any of them
            float4 va[INTERP_BATCH], vb[INTERP_BATCH];
#pragma unroll
            for (int i = 0; i < INTERP_BATCH; i++) {
                const float *pr = s_rec[wv][4 * (p4 + i) + grp];
                // a point outside the model reads the first column instead: its results are never used
                const int b = __float_as_int(pr[14]) == 0 ? __float_as_int(pr[j]) : 0;
                va[i] = dlane[b];
                vb[i] = dlane[b + nq];
            }
#pragma unroll
            for (int i = 0; i < INTERP_BATCH; i++) {
                // nothing to do for four points that all lie outside the model (the tail of a ray above the model
                // top, the gates below the terrain): their outputs are sentinels
                float *pr = s_rec[wv][4 * (p4 + i) + grp];
                if (!__any(__float_as_int(pr[14]) == 0)) continue;
                const float d = pr[4 + j], hz = pr[8 + j];
                const float px = pr[12], py = pr[13], pdx = 1.0f - px, pdy = 1.0f - py;
                const float A[4] = {va[i].x, va[i].y, va[i].z, va[i].w}, B[4] = {vb[i].x, vb[i].y, vb[i].z, vb[i].w};
                // the column's product of  dx*dy*v0 + x*v2*dy + dx*v1*y + x*y*v3  (interpolation_c.c:162) in the
                // reference's left-to-right order, written (f w) g with g = 1 where the weights come first
                const float f = j == 0 ? pdx * pdy : j == 1 ? pdx : j == 2 ? px : px * py;
                const float g = j == 1 ? py : j == 2 ? pdy : 1.0f;
                float o[4];
                const float rd = mv.fast_div ? refined_rcp(d) : 0.f;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float nn = B[c] - A[c];
                    const float w = B[c] - (mv.fast_div ? div_refined(nn, d, rd) : nn / d) * hz;     // vertical interpolation (:151)
                    const float t = (f * w) * g;
                    // lanes of column 0 collect ((t0 + t2) + t1) + t3: columns 2, 1, 3 sit 8, 4, 12 lanes up
                    const int s2 = __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, t), 0x201f);
                    const int s1 = __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, t), 0x101f);
                    const int s3 = __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, t), 0x301f);
                    o[c] = ((t + __builtin_bit_cast(float, s2)) + __builtin_bit_cast(float, s1)) + __builtin_bit_cast(float, s3);
                }
                if (j == 0 && kact) {          // the record has been read: its first words take the results
#pragma unroll
                    for (int c = 0; c < 4; c++) pr[4 * k + c] = o[c];
                }
            }
        }
    }
    __syncthreads();
    return code;
}

// Workgroups are dealt to the eight XCDs in turn, each XCD with an L2 of its own.  Neighbouring sub-beams and
// rays read the same model columns, so an XCD takes runs of INTERP_XCD_RUN consecutive blocks (about four
// rays of nine sub-beams); short runs keep the XCDs evenly loaded although high elevations leave the model early.
__device__ inline long interp_block_id()
{
    const long kx = blockIdx.x >> 3;
    return ((kx / INTERP_XCD_RUN) * 8 + (blockIdx.x & 7)) * INTERP_XCD_RUN + kx % INTERP_XCD_RUN;
}

__global__ void k_interp(ModelView mv, long n, const float *__restrict__ coords, const float *__restrict__ Hh,
                         float *__restrict__ vals, long plane, signed char *__restrict__ mask,
                         float *__restrict__ raw_out)
{
    __shared__ float s_rec[4][64][INTERP_REC];
    const long u = interp_block_id() * blockDim.x + threadIdx.x;
    const bool live = u < n;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float qnan = __builtin_nanf("");
    const int code = interp_point(mv, s_rec, live, live ? Hh[u] : 0.f, live ? coords[2 * u] : 0.f, live ? coords[2 * u + 1] : 0.f);
    if (mask && live) mask[u] = (signed char)code;
    if (live) {
        for (int v = 0; v < mv.nv; v++) {
            float o = s_rec[wv][lane][v];
            // raw_out keeps the reference's sentinels; vals gets NaN (interpolation.py:277)
            if (raw_out) raw_out[(long)v * n + u] = code == 0 ? o : (code < 0 ? -9999.f : qnan);
            if (vals) vals[(long)mv.var_of_slot[v] * plane + u] = code == 0 ? o : qnan;
        }
    }
}

void launch_interp(hipStream_t s, const ModelView &mv, long n, const float *coords, const float *H,
                   float *vals, long plane, signed char *mask, float *raw_out)
{
    const long nb = (n + 255) / 256;
    const long per = 8L * INTERP_XCD_RUN;
    hipLaunchKernelGGL(k_interp, dim3((unsigned)((nb + per - 1) / per * per)), dim3(256), 0, s, mv, n, coords, H, vals,
                       plane, mask, raw_out);
}

// =============================================================================
// Lookup-table image for k_integrate, built on the device whenever tables or schemes change:
//   img[((q*4*ncol + k*ncol + j) * 4 + s] = B'[d][j],  d = 16 q + 4 s + k
//   B'[d][j <  12] = sz[d][j] * D^mu                       (scattering_sz.py:95-103, PSD weight folded in)
//   B'[d][12]      = rcs[d] * D^mu_d,  B'[d][13] = rcs[d] * v(D) * D^mu_d     (doppler_sz.py:192,218-220)
// with rcs = 2 pi (sz0 - sz1 - sz2 + sz3) and v = alpha D^beta in float32 as in the reference.
// ncol = CPOL_N_COMB (8): the image of the combined columns instead,
//   B'[d][0..6] = (h, v, s4 + s7, s6 - s5, s10 - s8, s11, s9)[d] * D^mu,  B'[d][7] = 0, or with with_dop the second
//   Doppler moment (the first is 2 pi B'[d][0]).
// xq[(q*4 + k)*4 + s] = D^nu - D_0^nu in the same permuted order.
// =============================================================================
__global__ void k_build_image(const float *__restrict__ raw, const float *__restrict__ d_axis, float *__restrict__ img,
                              float *__restrict__ xq, long nslice, int ncol, double mu, double nu, int with_dop,
                              int zero_pol, double mu_d, double alpha_d, double beta_d, double x_ref)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = nslice * 1024 * ncol;
    if (t >= total) return;
    int j = (int)(t % ncol);
    int d = (int)((t / ncol) % 1024);
    long sl = t / ((long)ncol * 1024);
    float D = d_axis[d];
    const float *src = raw + (sl * 1024 + d) * CPOL_N_SZ;
    float val;
    if (ncol == CPOL_N_COMB) {
        // the combinations of columns the observables are made of (scattering_sz.py:168-194), formed in float64 from the
        // table's float32 entries and rounded once: h, v, s4 + s7, s6 - s5, s10 - s8, s11, s9, and a spare column
        const double z[CPOL_N_SZ] = {src[0], src[1], src[2], src[3], src[4], src[5], src[6], src[7], src[8], src[9], src[10], src[11]};
        const double g = j == 0 ? ((z[0] - z[1]) - z[2]) + z[3] : j == 1 ? ((z[0] + z[1]) + z[2]) + z[3] : j == 2 ? z[4] + z[7]
                       : j == 3 ? z[6] - z[5] : j == 4 ? z[10] - z[8] : j == 5 ? z[11] : j == 6 ? z[9] : 0.;
        val = (float)(g * pow((double)D, mu));
        if (j == 7 && with_dop) {
            // the second Doppler scheme-2 moment, rcs v(D) D^mu_d, as column 13 of the 14-column image holds it; the
            // first, rcs D^mu_d, is 2 pi times column 0 when the Doppler PSD is the polarimetric one (mu_d = mu)
            float rcs = (float)(2 * M_PI) * (((src[0] - src[1]) - src[2]) + src[3]);
            val = (rcs * ((float)alpha_d * powf(D, (float)beta_d))) * (float)pow((double)D, mu_d);
        }
    }
    else if (j < CPOL_N_SZ) val = zero_pol ? 0.f : src[j] * (float)pow((double)D, mu);
    else if (!with_dop) val = 0.f;
    else {
        float rcs = (float)(2 * M_PI) * (((src[0] - src[1]) - src[2]) + src[3]);
        float xm = (float)pow((double)D, mu_d);
        val = rcs * xm;
        if (j == 13) val = (rcs * ((float)alpha_d * powf(D, (float)beta_d))) * xm;
    }
    int q = d >> 4, s4 = (d >> 2) & 3, k = d & 3;
    const int jw = (ncol == CPOL_N_COMB && (k & 2)) ? j ^ 4 : j;      // word of the column inside its row of the image
    img[sl * 1024 * ncol + (((long)(q * 4 + k) * ncol + jw) << 2) + s4] = val;
    // exponent argument relative to x_ref = the smallest D^nu of the axis (the first diameter of the reference's
    // tables): k_prep folds exp(-lambda x_ref) into the unit's coefficient in float64, so the float32 sums keep
    // their leading terms however large lambda is
    if (sl == 0 && j == 0) xq[((q * 4 + k) << 2) + s4] = (float)(pow((double)D, nu) - x_ref);
}

void launch_build_image(hipStream_t s, const float *raw, const float *d_axis, float *img, float *xq, long nslice,
                        int ncol, double mu, double nu, int with_dop, int zero_pol, double mu_d, double alpha_d,
                        double beta_d, double x_ref)
{
    long total = nslice * 1024 * ncol;
    hipLaunchKernelGGL(k_build_image, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, raw, d_axis, img, xq, nslice,
                       ncol, mu, nu, with_dop, zero_pol, mu_d, alpha_d, beta_d, x_ref);
}

// The scale of a table node's first-diameter terms: floor[node] = the smallest |sz[node][first diameter][k]| D_0^mu
// over the twelve columns, or 0 when one of them is zero there or the node holds a NaN anywhere.  In the integration
// kernels the first diameter's N(D) factor is exactly 1, so a unit's coefficient times this number is the size of the
// smallest leading term among its twelve sums; a column that keeps one sign along the diameter axis has a sum at least
// that large.  One workgroup per node.
__global__ __launch_bounds__(256)
void k_node_floor(const float *__restrict__ raw, const float *__restrict__ d_axis, double mu, float *__restrict__ floor_out)
{
    __shared__ int s_bad[4];
    const long sl = blockIdx.x;
    int bad = 0;
    for (int d = threadIdx.x; d < 1024; d += 256) {
        const float *src = raw + (sl * 1024 + d) * CPOL_N_SZ;
#pragma unroll
        for (int k = 0; k < CPOL_N_SZ; k++) bad |= src[k] != src[k];
    }
    bad = __any(bad);
    if ((threadIdx.x & 63) == 0) s_bad[threadIdx.x >> 6] = bad;
    __syncthreads();
    if (threadIdx.x == 0) {
        float fl = 0.f;
        if (!(s_bad[0] | s_bad[1] | s_bad[2] | s_bad[3])) {
            const float dm = (float)pow((double)d_axis[0], mu);
            const float *src = raw + sl * 1024 * CPOL_N_SZ;
            fl = __builtin_inff();
            for (int k = 0; k < CPOL_N_SZ; k++) fl = fminf(fl, fabsf(src[k] * dm));
            if (!(fl > 0.f) || !(fl < __builtin_inff())) fl = 0.f;
        }
        floor_out[sl] = fl;
    }
}

void launch_node_floor(hipStream_t s, const float *raw, const float *d_axis, long nslice, double mu, float *floor_out)
{
    hipLaunchKernelGGL(k_node_floor, dim3((unsigned)nslice), dim3(256), 0, s, raw, d_axis, mu, floor_out);
}

// Certificate for a term cutoff below the exact-zero one.  A unit with slope l2 leaves out the quads whose first
// D^nu - x_ref exceeds cut_e / l2, so the units that stop at the quad boundary c = 16 (q + 1) have slopes in
// (cut_e / x_c, cut_e / x_{c-16}].  The observables are made of linear combinations of the twelve columns
// (scattering_sz.py:168-194), and a difference of two nearly equal columns can lose far more of itself to an
// omitted tail than either column does, so the certificate is about those combinations:
//     h = s0 - s1 - s2 + s3 (ZH, ZDR, RHOHV), v = s0 + s1 + s2 + s3 (ZV, ZDR, RHOHV), k = s10 - s8 (KDP),
//     s11 (AH), s9 (AV), the complex number (s4 + s7) + i (s6 - s5) whose modulus enters RHOHV and whose
//     argument is DELTA_HV, and the two Doppler scheme-2 moments.
// For every table node, every boundary, CERT_SAMPLES slopes spread over the boundary's interval and every such
// quantity g the kernel forms
//     sum_{d >= c} |g_d| 2^(-l2 x_d)  /  | sum_d g_d 2^(-l2 x_d) |
// - the omitted terms at their worst (all of one sign) against the signed value the unit contributes; the two parts
// of the complex number are both measured against its modulus, the scale RHOHV and DELTA_HV see them at.  The
// maximum over everything goes to *worst; the host accepts cut_e when it is below 2^-28, a sixteenth of float32's
// resolution.  A combination that changes sign along the diameter axis can have a unit value near zero at some
// slope; the ratio is then large and the class keeps the exact-zero rule, which is always safe.
// with_dop: 0 = the polarimetric quantities, 1 = those and the two Doppler moments under the same PSD, 2 = the two
// Doppler moments alone under the PSD passed as (mu, nu, x_ref) = (mu_d, ...): the separate Doppler units.
#define CERT_SAMPLES 4
#define CERT_NG 9
__global__ __launch_bounds__(256)
void k_cut_certificate(const float *__restrict__ raw, const float *__restrict__ d_axis, double mu, double nu, double x_ref,
                       int with_dop, double mu_d, double alpha_d, double beta_d, double cut_e, unsigned *__restrict__ worst)
{
    constexpr int NG = CERT_NG;
    __shared__ double s_red[4][2 * NG];
    const long sl = blockIdx.x;
    const int c = 16 * ((int)blockIdx.y + 1);
    const double xc = pow((double)d_axis[c], nu) - x_ref;
    if (!(xc > 0.)) return;
    // slopes of the boundary's interval, from the smallest (the omitted terms weigh most) towards the largest; the
    // first boundary's interval has no upper end and is sampled up to eight times its lower one
    const double x_prev = pow((double)d_axis[c - 16], nu) - x_ref;
    const double span = (c > 16 && x_prev > 0.) ? xc / x_prev : 8.;
    const double l2 = cut_e / xc * pow(span, (double)blockIdx.z / CERT_SAMPLES);
    double tot[NG], tail[NG];
#pragma unroll
    for (int j = 0; j < NG; j++) { tot[j] = 0.; tail[j] = 0.; }
    for (int d = threadIdx.x; d < 1024; d += 256) {
        const double D = (double)d_axis[d];
        const double n = exp2(-l2 * (pow(D, nu) - x_ref));
        const double dm = pow(D, mu) * n;
        const float *src = raw + (sl * 1024 + d) * CPOL_N_SZ;
        double z[CPOL_N_SZ], g[NG];
#pragma unroll
        for (int j = 0; j < CPOL_N_SZ; j++) z[j] = (double)src[j] * dm;
        g[0] = ((z[0] - z[1]) - z[2]) + z[3];
        g[1] = ((z[0] + z[1]) + z[2]) + z[3];
        g[2] = z[10] - z[8];
        g[3] = z[4] + z[7];
        g[4] = z[6] - z[5];
        g[5] = z[9];
        g[6] = z[11];
        g[7] = 0.; g[8] = 0.;
        if (with_dop) {
            const double rcs = 2 * M_PI * ((((double)src[0] - src[1]) - src[2]) + src[3]) * pow(D, mu_d) * n;
            g[7] = rcs; g[8] = rcs * alpha_d * pow(D, beta_d);
        }
        if (with_dop == 2) {                               // the Doppler units' own PSD (mu, nu are its): the two moments only
#pragma unroll
            for (int j = 0; j < 7; j++) g[j] = 0.;
        }
#pragma unroll
        for (int j = 0; j < NG; j++) { tot[j] += g[j]; if (d >= c) tail[j] += fabs(g[j]); }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NG; j++) {
        for (int off = 32; off > 0; off >>= 1) { tot[j] += __shfl_xor(tot[j], off); tail[j] += __shfl_xor(tail[j], off); }
        if (lane == 0) { s_red[wv][j] = tot[j]; s_red[wv][NG + j] = tail[j]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[NG], tl[NG];
        for (int j = 0; j < NG; j++) {
            t[j] = fabs(s_red[0][j] + s_red[1][j] + s_red[2][j] + s_red[3][j]);
            tl[j] = s_red[0][NG + j] + s_red[1][NG + j] + s_red[2][NG + j] + s_red[3][NG + j];
        }
        t[3] = t[4] = hypot(t[3], t[4]);                   // both parts against the modulus
        double r = 0.;
        for (int j = 0; j < NG; j++) {
            if (tl[j] == 0.) continue;                     // nothing omitted (an all-zero combination)
            if (t[j] > 0.) r = fmax(r, tl[j] / t[j]);
            else r = 1.;                                   // nothing kept: no certificate
            if (tl[j] != tl[j] || t[j] != t[j]) r = 1.;    // not a number: no certificate
        }
        const float rf = (float)fmin(r * 1.000001, 1.);   // rounded up; non-negative floats order like their bit patterns
        atomicMax(worst, __float_as_uint(rf));
    }
}

float cut_certificate(hipStream_t s, const float *raw, const float *d_axis, long nslice, double mu, double nu, double x_ref,
                      int with_dop, double mu_d, double alpha_d, double beta_d, double cut_e, float *worst_dev)
{
    if (hipMemsetAsync(worst_dev, 0, sizeof(float), s) != hipSuccess) return 1.f;
    hipLaunchKernelGGL(k_cut_certificate, dim3((unsigned)nslice, 63, CERT_SAMPLES), dim3(256), 0, s, raw, d_axis, mu, nu, x_ref,
                       with_dop, mu_d, alpha_d, beta_d, cut_e, (unsigned *)worst_dev);
    float w = 1.f;
    if (hipMemcpyAsync(&w, worst_dev, sizeof(float), hipMemcpyDeviceToHost, s) != hipSuccess) return 1.f;
    if (hipStreamSynchronize(s) != hipSuccess) return 1.f;
    return w;
}

// Compact image of the Doppler scheme-2 columns for units with a PSD of their own (k_integrate_dop):
//   img[sl][d] = D^nu_d - D_0^nu_d  (d < 1024),   img[sl][1024 + 2 d + c] = rcs D^mu_d (c = 0), rcs v(D) D^mu_d (c = 1)
// i.e. 12 KB per (e, t) node; the two moments of a diameter sit side by side for packed arithmetic.  The exponent is taken
// relative to the first diameter (k_prep folds exp(-lambda D_0^nu) into the unit's N0), so that the float32
// sums keep their leading terms however large lambda is; the reference integrates in float64.
__global__ void k_build_dop_image(const float *__restrict__ raw, const float *__restrict__ d_axis,
                                  float *__restrict__ img, float *__restrict__ xq, long nslice, double mu_d, double nu_d,
                                  double alpha_d, double beta_d, double x_ref)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nslice * 1024) return;
    int d = (int)(t % 1024);
    long sl = t / 1024;
    float D = d_axis[d];
    const float *src = raw + (sl * 1024 + d) * CPOL_N_SZ;
    float rcs = (float)(2 * M_PI) * (((src[0] - src[1]) - src[2]) + src[3]);
    float xm = (float)pow((double)D, mu_d);
    float x = (float)pow((double)D, nu_d);
    float x0 = (float)x_ref;                    // the smallest D^nu_d of the axis, as float32
    float *o = img + sl * CPOL_DOP_SLICE;
    o[d] = (float)((double)x - (double)x0);
    o[1024 + 2 * d] = rcs * xm;
    o[1024 + 2 * d + 1] = (rcs * ((float)alpha_d * powf(D, (float)beta_d))) * xm;
    if (sl == 0) xq[d] = x;
}

void launch_build_dop_image(hipStream_t s, const float *raw, const float *d_axis, float *img, float *xq, long nslice,
                            double mu_d, double nu_d, double alpha_d, double beta_d, double x_ref)
{
    long total = nslice * 1024;
    hipLaunchKernelGGL(k_build_dop_image, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, raw, d_axis, img, xq,
                       nslice, mu_d, nu_d, alpha_d, beta_d, x_ref);
}

// =============================================================================
// PSD set-up per (gate, sub-beam, class): valid test, nearest lookup-table node,
// PSD parameters (hydrometeors.py set_psd), bucket histogram; Doppler scheme 1
// radial velocity per sub-beam.
// =============================================================================
__device__ inline int nearest_node(float x, float lo, float step, int n)
{   // lut.py:240-241: np.round (half to even) of float32 (x - lo) / step, clamped
    float c = rintf((x - lo) / step);
    c = fminf(fmaxf(0.f, c), (float)(n - 1));
    return (int)c;
}

__device__ inline float fold_elev(float e)
{   // scattering_sz.py:66-68
    if (e > 90.f) e = 180.f - e;
    if (e < 0.f) e = -e;
    return e;
}

// x^y for x > 0 through exp(y log x): relative error ~ |y log x| * 2^-53, far below the
// float32 resolution of every consumer, at about half the cost of pow().
__device__ inline double powxy(double x, double y) { return exp(y * log(x)); }

// lambda and N0 of one class at one gate (float64, float32 where the reference's
// NumPy promotion keeps float32).  Returns false when Q is not usable.
__device__ inline bool psd_params(const PsdView &p, float q, float qn, float T, double *lam, double *n0,
                                  double *loglam = nullptr)
{
    double ll;                       // natural log of *lam: later powers of lambda reuse it
    if (p.kind == 0) {
        ll = log(p.lambda_factor / (double)q) * p.c_exp;
        *lam = exp(ll);
        *n0 = p.n0_fixed;
    } else if (p.kind == 1) {
        // exp through float64 so that subnormal float32 results are kept, as glibc's expf keeps them
        float n0f = 13.5f * (565000.0f * (float)exp((double)(-0.107f * (T - 273.15f)))) / 1000.f;
        float an0 = (float)p.a * n0f;
        ll = log((double)an0 * p.lambda_factor / (double)q) * p.c_exp;
        *lam = exp(ll);
        *n0 = (double)n0f;
    } else {
        float xm = q * 1.0f / (qn + 1.1920929e-07f);
        xm = fminf(fmaxf(xm, (float)p.x_min), (float)p.x_max);
        double l0 = log(p.lambda_factor * (double)xm) * p.c_exp;
        double n = p.c_n * (double)qn * exp(l0 * p.c_n_exp);
        *n0 = n * p.n0_unit;
        *lam = exp(l0) * p.lam_unit;
        ll = l0 + p.log_lam_unit;
    }
    if (loglam) *loglam = ll;
    return true;
}

// D^nu beyond which a unit's terms are left out: exp2(-l2 * D^nu) < 2^-(cut_e - 2).  cut_e = 152: the exponent
// passes -150 even after the roundings of the quotient and of the product, N(D) is an exact float32 zero and the
// terms past it add nothing to a float32 sum.  A smaller cut_e is used only for a class whose tables carry a
// certificate (k_cut_certificate) that the terms past it stay below 2^-28 of the sum for every lambda.  +inf (no
// cutoff) for a lambda that is not a positive finite number, so that its NaN / inf results are kept.
__device__ inline float psd_cutoff_x(float l2, float cut_e = 152.f)
{
    return (l2 > 0.f && l2 < __builtin_inff()) ? cut_e / l2 : __builtin_inff();
}

// ---- float64 log2 / exp2 for the PSD set-up -----------------------------------------------------------------
// The consumers of the PSD parameters are a float32 pair hi + lo of lambda log2(e) (exponent arguments exact to one
// float32 rounding need it to 2^-40) and coefficients that end in float32 results (1e-9 is plenty).  The library
// routines are correct to an ulp of float64 and pay for it with extended-precision arithmetic; these two stay
// below 2^-44 relative (log2: absolute, on a result of order one or more) with plain float64 operations:
//   log2_psd: x = 2^e m, m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), log2 m = (2 / ln 2) s (1 + s^2/3 + ... + s^16/17);
//             |s| <= 0.1716, the first term left out is s^18 / 19 < 9e-16 of the leading one.
//             0 -> -inf, +inf -> +inf, negative or NaN -> NaN.
//   exp2_psd: y = n + f, |f| <= 1/2, 2^f by the Taylor polynomial of degree 12 in f ln 2 (next term (0.3466)^13 / 13!
//             = 1.7e-16), scaled by v_ldexp_f64, which rounds into the subnormal range and overflows to inf like
//             the library routine.  -inf -> 0, +inf -> inf, NaN -> NaN.
__device__ inline double rcp_psd(double d)
{   // 1 / d for a normal, finite d: hardware seed and two Newton steps
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.), r, r);
    return r;
}

__device__ inline double log2_psd(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);          // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = (m - 1.) * rcp_psd(m + 1.);
    const double z = s * s;
    double p = 2.8853900817779268 / 17.;
    p = __builtin_fma(p, z, 2.8853900817779268 / 15.);
    p = __builtin_fma(p, z, 2.8853900817779268 / 13.);
    p = __builtin_fma(p, z, 2.8853900817779268 / 11.);
    p = __builtin_fma(p, z, 2.8853900817779268 / 9.);
    p = __builtin_fma(p, z, 2.8853900817779268 / 7.);
    p = __builtin_fma(p, z, 2.8853900817779268 / 5.);
    p = __builtin_fma(p, z, 2.8853900817779268 / 3.);
    p = __builtin_fma(p, z, 2.8853900817779268);
    double r = __builtin_fma(s, p, (double)e);
    r = x == 0. ? -__builtin_inf() : r;
    r = x == __builtin_inf() ? x : r;
    return x >= 0. ? r : __builtin_nan("");
}

__device__ inline double exp2_psd(double y)
{
    y = y < -1100. ? -1100. : y;                        // v_ldexp_f64 takes a 32-bit exponent: keep it small
    y = y > 1100. ? 1100. : y;                          // (a NaN passes both tests and comes out as NaN)
    const double n = __builtin_rint(y);
    const double t = (y - n) * 0.69314718055994531;
    double p = 1. / 479001600.;
    p = __builtin_fma(p, t, 1. / 39916800.);
    p = __builtin_fma(p, t, 1. / 3628800.);
    p = __builtin_fma(p, t, 1. / 362880.);
    p = __builtin_fma(p, t, 1. / 40320.);
    p = __builtin_fma(p, t, 1. / 5040.);
    p = __builtin_fma(p, t, 1. / 720.);
    p = __builtin_fma(p, t, 1. / 120.);
    p = __builtin_fma(p, t, 1. / 24.);
    p = __builtin_fma(p, t, 1. / 6.);
    p = __builtin_fma(p, t, 0.5);
    p = __builtin_fma(p, t, 1.);
    p = __builtin_fma(p, t, 1.);
    return __builtin_ldexp(p, (int)n);
}

// The PSD of one class at one point as the integration kernels take it: l2 = lambda log2(e) and
// nf = N0 2^(-l2 x0), the intercept with the factor of the reference abscissa x0 folded in (hydrometeors.py
// set_psd; the branches of psd_params, in base 2, with the two exponentials of N0 and of the factor merged into
// one).  Not usable content (a mass density that is zero, negative or NaN where the branch divides by it): NaN.
__device__ inline void psd_fast(const PsdView &p, float q, float qn, float T, double x0, double *l2_out, double *nf_out)
{
    double l2, nf;
    if (p.kind == 2) {
        float xm = q * 1.0f / (qn + 1.1920929e-07f);
        xm = fminf(fmaxf(xm, (float)p.x_min), (float)p.x_max);
        const double L = (p.log2_lf + log2_psd((double)xm)) * p.c_exp;
        l2 = exp2_psd(L + p.l2_const);
        nf = p.cn_unit * (double)qn * exp2_psd(__builtin_fma(L, p.c_n_exp, -l2 * x0));
    } else {
        double n0 = p.n0_fixed, lq = p.log2_lf;
        if (p.kind == 1) {
            // float32 intercept as the reference forms it; the exponential through float64 so that subnormal float32
            // results are kept, as glibc's expf keeps them
            const float n0f = 13.5f * (565000.0f * (float)exp2_psd((double)(-0.107f * (T - 273.15f)) * 1.4426950408889634)) / 1000.f;
            const float an0 = (float)p.a * n0f;
            n0 = (double)n0f;
            lq += log2_psd((double)an0);
        }
        const double L = (lq - log2_psd((double)q)) * p.c_exp;
        l2 = exp2_psd(L + p.l2_const);
        nf = n0 * exp2_psd(-l2 * x0);
        if (!(q > 0.f)) { l2 = __builtin_nan(""); nf = l2; }
    }
    *l2_out = l2; *nf_out = nf;
}

// Position of the run of equal keys this lane belongs to inside its wave.
__device__ inline void wave_run(int k, int lane, int *head_lane, int *len)
{
    int prev = __shfl_up(k, 1);
    bool head = (lane == 0) || (k != prev);
    unsigned long long hb = __ballot(head);
    unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1);
    int hl = 63 - __clzll(hb & upto);
    unsigned long long above = (hl == 63) ? 0ull : (hb & ~((2ull << hl) - 1));
    int end = above ? (__ffsll((long long)above) - 1) : 64;
    *head_lane = hl;
    *len = end - hl;
}

// Doppler scheme 1 for one point: closed-form PSD-weighted fall speed (doppler_sz.py:130-153) and projection on
// the beam (:75-87).  rb = index of the point's (ray, sub-beam); the density correction refers to its first gate.
__device__ inline double doppler1_radial(const ClassTable &ct, const float *__restrict__ vals, long plane, long uu, long rb,
                                         int n_gates, float e_raw, const double *__restrict__ phi_sc)
{
    double vh = 0., nn = 0.;            // nansum_arr: NaN counts as 0 (utilities.py:46-66)
    for (int h = 0; h < ct.n_cls; h++) {
        const PsdView &p = ct.dop[h];
        float q = vals[(long)p.q_var * plane + uu];
        float qn = p.qn_var >= 0 ? vals[(long)p.qn_var * plane + uu] : 0.f;
        // a negative mass density (extrapolation below the lowest level) still yields finite
        // 2-moment parameters in the reference (clamped mean mass); the other branches give NaN
        if (q > 0.f || (p.kind == 2 && q < 0.f)) {
            double l, n0, vi, ni, ll;
            float Tp = vals[(long)p.t_var * plane + uu];
            psd_params(p, q, qn, Tp, &l, &n0, &ll);
            vi = p.vel_factor * n0 * p.alpha / p.nu * exp(ll * (-(p.beta + p.mu + 1) / p.nu));
            if (p.kind == 2) ni = (double)qn;
            else ni = p.ntot_factor * n0 / p.nu * exp(ll * (-(p.mu + 1) / p.nu));
            vh += (vi != vi ? 0. : vi);
            nn += (ni != ni ? 0. : ni);
        } else if (p.kind == 2 && qn == qn) {
            nn += (double)qn;             // lambda is NaN, N_int = ntot = QN (hydrometeors.py:77-78)
        }
    }
    float rho = vals[(long)CPOL_V_RHO * plane + uu];
    float rho0 = vals[(long)CPOL_V_RHO * plane + rb * n_gates];
    float rc = sqrtf(rho / rho0);
    double vhyd = vh / nn * (double)rc;
    float th = e_raw * (float)(M_PI / 180.0);
    double sph = phi_sc[2 * rb], cph = phi_sc[2 * rb + 1];
    double U = vals[(long)CPOL_V_U * plane + uu], V = vals[(long)CPOL_V_V * plane + uu];
    double W = vals[(long)CPOL_V_W * plane + uu];
    return (U * sph + V * cph) * (double)cosf(th) + (W - vhyd) * (double)sinf(th);
}

// PSD set-up of one point for every class: table node (bucket key), lambda, float64 coefficients, and the point's
// position inside its (bucket, shard) counter.  val(v) returns the point's model variable v; the 64 lanes of a wave
// hold 64 consecutive points.
// Returns whether one of the point's units takes the combined columns.
template <class ValFn>
__device__ inline bool prep_classes(const ClassTable &ct, long nu, long u, bool live, int lane, float T, float e_raw,
                                    double w, int doppler, ValFn val, cpol_key_t *__restrict__ key,
                                    int *__restrict__ rank, float2 *__restrict__ lam, double *__restrict__ coef,
                                    double *__restrict__ n0d, int *__restrict__ hist)
{
    const float e = fold_elev(e_raw);
    bool any_comb = false;
    for (int h = 0; h < ct.n_cls; h++) {
        const PsdView &p = ct.psd[h];
        const LutView &lv = ct.lut[h];
        float q = val(p.q_var);
        float qn = p.qn_var >= 0 ? val(p.qn_var) : 0.f;
        bool valid = live && (q > 0.f);
        if (__builtin_amdgcn_ballot_w64(valid) == 0) {               // no unit of this class in the whole wavefront
            if (live) key[(long)h * nu + u] = (cpol_key_t)-1;
            continue;
        }
        int k = -1;
        if (valid) {
            int ie = nearest_node(e, lv.e_min, lv.e_step, lv.n_e);
            int it = nearest_node(T, lv.t_min, lv.t_step, lv.n_t);
            // the integration works with exp(-lambda (D^nu - D_0^nu)): the factor of the first diameter goes
            // into the float64 coefficients
            double l2, nf;
            psd_fast(p, q, qn, T, ct.pol_x0[h], &l2, &nf);
            float hi = (float)l2;
            lam[(long)h * nu + u] = make_float2(hi, (float)(l2 - (double)hi));
            const double c = nf * (double)lv.dD * w;
            coef[(long)h * nu + u] = c;
            if (doppler == 2 && ct.dop_shared) n0d[(long)h * nu + u] = nf;
            // form of the unit (ClassTable::comb): the combined columns when c times the smallest first-diameter term of
            // its node - the scale of its smallest contribution to any of the twelve components - is a normal float32
            // far from the subnormals; else (also for c that is not a number, or a node with a zero or NaN entry) all
            // twelve columns, treated as the reference treats them
            const bool combined = ct.comb && c * (double)ct.lut_comb[h].floor[ie * lv.n_t + it] >= 0x1p-100;
            any_comb = any_comb || combined;
            const int plane = combined ? h : (ct.comb ? ct.n_cls + h : h);
            k = (plane * ct.ne_max + ie) * ct.nt_max + it;
        }
        if (live) key[(long)h * nu + u] = (cpol_key_t)k;
        // position of the unit inside its (bucket, shard) counter: one atomic per run of equal keys
        int hl, len;
        wave_run(k, lane, &hl, &len);
        int base = 0;
        if (k >= 0 && hl == lane) base = atomicAdd(&hist[k * CPOL_SORT_SHARDS + (int)((u >> 8) & (CPOL_SORT_SHARDS - 1))], len);
        base = __shfl(base, hl);
        if (k >= 0) rank[(long)h * nu + u] = base + (lane - hl);
        if (ct.n_kinds == 2) {
            // Doppler scheme 2 with a PSD of its own (doppler_sz.py:57: 1-moment classes fed the
            // 2-moment variables): same table node, separate unit (kind 1)
            const PsdView &pd = ct.dop[h];
            const long dslot = (long)(ct.n_cls + h) * nu + u;
            if (valid) {
                float qd = val(pd.q_var);
                float qnd = pd.qn_var >= 0 ? val(pd.qn_var) : 0.f;
                float Td = val(pd.t_var);
                double l2, nf;
                psd_fast(pd, qd, qnd, Td, ct.dop_x0[h], &l2, &nf);      // k_integrate_dop works relative to the first diameter
                if (qd == 0.f) { l2 = __builtin_nan(""); nf = l2; }
                float hi = (float)l2;
                lam[dslot] = make_float2(hi, (float)(l2 - (double)hi));
                n0d[dslot] = nf;
            }
            // the Doppler units of a bucket are as many as its polarimetric ones and take the same ranks:
            // k_scan_shards and k_scatter reuse the polarimetric counters, no second round of atomics
        }
    }
    return any_comb;
}

__global__ __launch_bounds__(256, 6) void k_prep(ClassTable ct, long nu, int n_gates, const float *__restrict__ vals, long plane,
                       const float *__restrict__ elev, signed char *__restrict__ mask /* invalid gates carry NaN; gains the code CPOL_MASK_COMB */,
                       const double *__restrict__ wsub, int n_sub, const double *__restrict__ phi_sc, int doppler,
                       cpol_key_t *__restrict__ key, int *__restrict__ rank, float2 *__restrict__ lam,
                       double *__restrict__ coef, double *__restrict__ n0d, int *__restrict__ hist,
                       double *__restrict__ vr)
{
    long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int lane = threadIdx.x & 63;
    bool live = u < nu;
    long uu = live ? u : 0;
    long rb = uu / n_gates;
    int b = (int)(rb % n_sub);
    float T = vals[(long)CPOL_V_T * plane + uu];
    float e_raw = elev[uu];
    const bool any_comb = prep_classes(ct, nu, u, live, lane, T, e_raw, wsub[b], doppler,
                                       [&](int v) { return vals[(long)v * plane + uu]; }, key, rank, lam, coef, n0d, hist);
    if (live && any_comb && mask[u] == 0) mask[u] = CPOL_MASK_COMB;
    if (doppler == 1 && live) vr[u] = doppler1_radial(ct, vals, plane, uu, rb, n_gates, e_raw, phi_sc);
}

// Doppler scheme 1 on its own, for callers whose PSD set-up ran inside another kernel.
__global__ void k_vr1(ClassTable ct, long nu, int n_gates, const float *__restrict__ vals, long plane,
                      const float *__restrict__ elev, const double *__restrict__ phi_sc, double *__restrict__ vr)
{
    const long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nu) return;
    vr[u] = doppler1_radial(ct, vals, plane, u, u / n_gates, n_gates, elev[u], phi_sc);
}

// The per-point front of the hot path in one kernel: position of the point (geom_point), interpolation of the
// model state (interp_point) and PSD set-up (prep_classes), the values passing from one to the next in registers
// and LDS.  Only what a later kernel reads goes to memory: the elevation, the interpolation code, the variables
// named in write_mask (bit v = model variable v), and the central sub-beam's position as the ray's outputs.
__global__ __launch_bounds__(256, FRONT_OCC)
void k_front(GeomParams gp, ModelView mv, ClassTable ct, long nu, const double *__restrict__ line,
             const double *__restrict__ wsub, int doppler, FrontOut fo)
{
    __shared__ float s_rec[4][64][INTERP_REC];
    const long u = interp_block_id() * blockDim.x + threadIdx.x;
    if (u - threadIdx.x >= nu) return;          // a workgroup of the grid's padding
    const bool live = u < nu;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long uu = live ? u : 0;
    const int g = (int)(uu % gp.n_gates);
    const long rb = uu / gp.n_gates;
    const int b = (int)(rb % gp.n_sub);
    GeoPoint p;
    double w_sub;
    {
        // The 64 positions of a wavefront nearly always lie on one (ray, sub-beam) line: its record (and the sub-beam's
        // weight) is then read through a wave-uniform pointer, i.e. by scalar loads that bypass the vector memory
        // path, which this kernel keeps as busy as its vector ALUs.
        const int rb_first = __builtin_amdgcn_readfirstlane((int)rb);
        if (__builtin_amdgcn_ballot_w64(rb != (long)rb_first) == 0) {
            geom_point(gp, mv, line + (long)rb_first * LINE_N, g, p);
            w_sub = wsub[rb_first % gp.n_sub];
        } else {
            geom_point(gp, mv, line + rb * LINE_N, g, p);
            w_sub = wsub[b];
        }
    }
    if (live) {
        fo.E[u] = p.E;
        if (p.outside) fo.status[(int)(rb / gp.n_sub)] = CPOL_ERR_DOMAIN;
        if (b == gp.n_sub / 2) {       // geometry of the central sub-beam (scattering_sz.py:154-157)
            const long o = (fo.ray0 + rb / gp.n_sub) * (long)gp.n_gates + g;
            if (fo.dist) fo.dist[o] = p.S;
            if (fo.height) fo.height[o] = p.H;
            if (fo.lat) fo.lat[o] = p.lat;
            if (fo.lon) fo.lon[o] = p.lon;
        }
    }
    const int code = interp_point(mv, s_rec, live, p.H, p.rlat, p.rlon);
    const float qnan = __builtin_nanf("");
    const float *rec = s_rec[wv][lane];
    if (live) {
        for (int v = 0; v < mv.nv; v++) {
            const int var = mv.var_of_slot[v];
            if ((fo.write_mask >> var) & 1u) fo.vals[(long)var * fo.plane + u] = code == 0 ? rec[v] : qnan;
        }
    }
    auto val = [&](int v) { return code == 0 ? rec[mv.slot_of_var[v]] : qnan; };
    const bool any_comb = prep_classes(ct, nu, u, live, lane, val(CPOL_V_T), p.E, w_sub, doppler, val, fo.key, fo.rank,
                                       fo.lam, fo.coef, fo.n0d, fo.hist);
    // interpolation code of the point; a point inside the model (code 0) that holds a unit in the combined form carries
    // CPOL_MASK_COMB instead, so that the reduction of its gate knows without looking at the units
    if (live) fo.mask[u] = (signed char)((code == 0 && any_comb) ? CPOL_MASK_COMB : code);
}

void launch_front(hipStream_t s, const GeomParams &gp, const ModelView &mv, const ClassTable &ct, int n_rays,
                  const double *az, const double *el, const double *quad, double *line, double *phi,
                  const double *wsub, int doppler, const FrontOut &fo)
{
    const int nl = n_rays * gp.n_sub;
    hipLaunchKernelGGL(k_geom_line, dim3((nl + 63) / 64), dim3(64), 0, s, gp, n_rays, az, el, quad, line, phi, fo.status);
    const long nu = (long)nl * gp.n_gates;
    const long nb = (nu + 255) / 256;
    const long per = 8L * INTERP_XCD_RUN;
    hipLaunchKernelGGL(k_front, dim3((unsigned)((nb + per - 1) / per * per)), dim3(256), 0, s, gp, mv, ct, nu, line, wsub,
                       doppler, fo);
}

void launch_vr1(hipStream_t s, const ClassTable &ct, long nu, int n_gates, const float *vals, long plane,
                const float *elev, const double *phi_sc, double *vr)
{
    hipLaunchKernelGGL(k_vr1, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s, ct, nu, n_gates, vals, plane, elev,
                       phi_sc, vr);
}

__global__ void k_sincos_deg(int n, const double *__restrict__ deg, double *__restrict__ sc)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double r = deg[i] * (M_PI / 180.0);
    sc[2 * i] = sin(r); sc[2 * i + 1] = cos(r);
}

void launch_sincos_deg(hipStream_t s, int n, const double *deg, double *sc)
{
    hipLaunchKernelGGL(k_sincos_deg, dim3((n + 63) / 64), dim3(64), 0, s, n, deg, sc);
}

void launch_prep(hipStream_t s, const ClassTable &ct, long nu, int n_gates, const float *vals, long plane,
                 const float *elev, signed char *mask, const double *weight_per_sub, int n_sub,
                 const double *phi_sc, int doppler, cpol_key_t *key, int *rank, float2 *lam, double *coef, double *n0d,
                 int *hist, double *vr)
{
    hipLaunchKernelGGL(k_prep, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s, ct, nu, n_gates, vals, plane,
                       elev, mask, weight_per_sub, n_sub, phi_sc, doppler, key, rank, lam, coef, n0d, hist, vr);
}

// Offsets from the sharded bucket histogram (nb buckets x CPOL_SORT_SHARDS counters):
// k_scan_shards (one thread per bucket, workgroups of SCAN_BLOCK buckets) turns a bucket's counters into shard
// cursors relative to the bucket start, and leaves in offs / chunk_off the bucket's unit and chunk offsets inside its
// workgroup's run of buckets and in blk_tot the run's totals; k_scan_buckets (one workgroup) scans the totals and adds
// every run's base to its buckets: bucket offsets and the chunk table of the integrate kernels.
#define SCAN_BLOCK 256
__global__ __launch_bounds__(SCAN_BLOCK)
void k_scan_shards(int nb, int nb_counted, const int *__restrict__ hist, int *__restrict__ offs,
                   int *__restrict__ cursor, int *__restrict__ chunk_off, int *__restrict__ blk_tot)
{
    __shared__ int s_v[SCAN_BLOCK / 64], s_c[SCAN_BLOCK / 64];
    const int j = blockIdx.x * SCAN_BLOCK + threadIdx.x;
    int v = 0;
    if (j < nb) {
        // buckets beyond nb_counted (separate Doppler units) mirror the counters of bucket j - nb_counted
        const int4 *h4 = (const int4 *)(hist + (j < nb_counted ? j : j - nb_counted) * CPOL_SORT_SHARDS);
        int4 *c4 = (int4 *)(cursor + j * CPOL_SORT_SHARDS);
#pragma unroll
        for (int i = 0; i < CPOL_SORT_SHARDS / 4; i++) {
            int4 h = h4[i], c;
            c.x = v; v += h.x; c.y = v; v += h.y; c.z = v; v += h.z; c.w = v; v += h.w;
            c4[i] = c;
        }
    }
    const int c = (v + CPOL_CHUNK_UNITS - 1) / CPOL_CHUNK_UNITS;
    // exclusive scan of (v, c) over the workgroup's buckets
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int iv = v, ic = c;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int ov = __shfl_up(iv, m), oc = __shfl_up(ic, m);
        if (lane >= m) { iv += ov; ic += oc; }
    }
    if (lane == 63) { s_v[wv] = iv; s_c[wv] = ic; }
    __syncthreads();
    int bv = 0, bc = 0;
    for (int w = 0; w < wv; w++) { bv += s_v[w]; bc += s_c[w]; }
    if (j < nb) { offs[j] = bv + iv - v; chunk_off[j] = bc + ic - c; }
    if (threadIdx.x == SCAN_BLOCK - 1) { blk_tot[2 * blockIdx.x] = bv + iv; blk_tot[2 * blockIdx.x + 1] = bc + ic; }
}

__global__ __launch_bounds__(1024)
void k_scan_buckets(int nb, int *__restrict__ offs, int *__restrict__ chunk_off, const int *__restrict__ blk_tot)
{
    __shared__ int s_a[1024], s_b[1024];
    const int t = threadIdx.x;
    const int nblk = (nb + SCAN_BLOCK - 1) / SCAN_BLOCK;          // <= 1024 runs
    const int a = t < nblk ? blk_tot[2 * t] : 0, c = t < nblk ? blk_tot[2 * t + 1] : 0;
    s_a[t] = a; s_b[t] = c;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        int xa = t >= d ? s_a[t - d] : 0, xb = t >= d ? s_b[t - d] : 0;
        __syncthreads();
        s_a[t] += xa; s_b[t] += xb;
        __syncthreads();
    }
    const int tot_a = s_a[1023], tot_b = s_b[1023];
    __syncthreads();
    s_a[t] -= a; s_b[t] -= c;                    // exclusive: the base of run t
    __syncthreads();
    for (int j = t; j < nb; j += 1024) { offs[j] += s_a[j / SCAN_BLOCK]; chunk_off[j] += s_b[j / SCAN_BLOCK]; }
    if (t == 0) { offs[nb] = tot_a; chunk_off[nb] = tot_b; }
}

// Bucket of every chunk, so that an integrate workgroup finds its bucket with one load instead of a
// bisection of dependent loads: one wavefront per bucket writes its id over the bucket's chunk range.
__global__ void k_fill_chunks(const int *__restrict__ chunk_off, int *__restrict__ chunk_bucket)
{
    const int j = blockIdx.x;
    for (int c = chunk_off[j] + threadIdx.x; c < chunk_off[j + 1]; c += blockDim.x) chunk_bucket[c] = j;
}

void launch_scan(hipStream_t s, int nb, int nb_counted, int *hist, int *offs, int *cursor, int *chunk_off,
                 int *chunk_bucket)
{
    int *blk_tot = chunk_off + nb + 1;            // 2 x CPOL_SCAN_RUNS_MAX ints behind the chunk offsets
    hipLaunchKernelGGL(k_scan_shards, dim3((nb + SCAN_BLOCK - 1) / SCAN_BLOCK), dim3(SCAN_BLOCK), 0, s, nb, nb_counted, hist, offs, cursor,
                       chunk_off, blk_tot);
    hipLaunchKernelGGL(k_scan_buckets, dim3(1), dim3(1024), 0, s, nb, offs, chunk_off, blk_tot);
    hipLaunchKernelGGL(k_fill_chunks, dim3(nb), dim3(64), 0, s, chunk_off, chunk_bucket);
}

// Final position of every valid polarimetric unit = bucket start + start of its shard inside the bucket + the
// rank the PSD set-up drew.  One thread per position, its (up to four) classes one after the other.
// The thread also leaves the point's units in one byte (uflag: bit h = class h holds a unit, bit 4 + h = that unit's
// key lies below nb_form, i.e. it takes the combined columns): the reduction of a gate walks its points by these
// bytes, not by the keys of all classes.
__global__ void k_scatter(long nu, int n_cls, const cpol_key_t *__restrict__ key, const int *__restrict__ rank,
                          const int *__restrict__ offs, const int *__restrict__ cursor, int *__restrict__ sorted,
                          int nb_form, unsigned char *__restrict__ uflag)
{
    long u = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= nu) return;
    const int shard = (int)((u >> 8) & (CPOL_SORT_SHARDS - 1));
    int k[4];
#pragma unroll
    for (int h = 0; h < 4; h++) k[h] = h < n_cls ? key[(long)h * nu + u] : -1;
#pragma unroll
    for (int h = 0; h < 4; h++)
        if (k[h] >= 0) sorted[offs[k[h]] + cursor[k[h] * CPOL_SORT_SHARDS + shard] + rank[(long)h * nu + u]] = (int)((long)h * nu + u);
    unsigned flags = 0;
#pragma unroll
    for (int h = 0; h < 4; h++)
        if (k[h] >= 0) flags |= (1u | (k[h] < nb_form ? 16u : 0u)) << h;
    uflag[u] = (unsigned char)flags;
}

void launch_scatter(hipStream_t s, long nu, int n_cls, const cpol_key_t *key, const int *rank, const int *offs,
                    const int *cursor, int *sorted, int nb_form, unsigned char *uflag)
{
    hipLaunchKernelGGL(k_scatter, dim3((unsigned)((nu + 255) / 256)), dim3(256), 0, s, nu, n_cls, key, rank, offs, cursor,
                       sorted, nb_form, uflag);
}

// =============================================================================
// PSD integration (scattering_sz.py:76-103): for every unit of one (form, class, e-node,
// t-node) bucket   r[k] = sum_d  B'[d][k] * exp(-lambda * (D_d^nu - D_0^nu)),  B' = table column x D^mu,
// as an [units x 1024] . [1024 x NC] contraction on the f32 matrix cores
// (v_mfma_f32_4x4x1_16b_f32, NC / 4 instructions per 16 units x 4 diameters).  NC = 12: the twelve
// scattering components; 8: the seven combinations of them the observables are made of
// (get_pol_from_sz, scattering_sz.py:168-194); 14: the components and the two Doppler scheme-2 moments.
// The bucket's table slice lives in LDS; every lane generates one N(D) value per step and set of
// 16 units (A operand), so nothing is computed twice.
// exp(-lambda (D^nu - D_0^nu)) is an exact float32 zero beyond a unit's cutoff diameter
// (psd_cutoff_x; earlier where the class's tables are certified for it): the workgroup orders its
// units by that cutoff, stages only the part of the slice its largest cutoff needs and runs passes
// of 32 units, two sets of 16, each set up to the cutoff of its last unit.  A unit's terms past its
// own cutoff are set to zero, so its sums do not depend on which units share its pass.
// =============================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define INTEG_THREADS 512
#define INTEG_XCD_RUN 16         // consecutive chunks per XCD before the next XCD's run

// Table image (global and LDS alike, built by k_build_image):
//   img[((q*4*NC + k*NC + j) * 4 + s] = B'[d][j]   with d = 16 q + 4 s + k
// i.e. for the quad q of four consecutive steps s = 0..3, a lane with diameter residue k finds the four B
// operands of column j in one 16-byte word.  NC = 12 (the scattering components), 14 with the two Doppler
// scheme-2 moments, or 8 for the combined columns (where the words of a row with k >= 2 are stored with the
// two column groups exchanged, so that the 16 distinct words of a read fall into 16 different bank quarters).
//
// One chunk of <= CPOL_CHUNK_UNITS units of one bucket, integrated over an image of NC columns.  ROW floats per
// result row in psd_out; with ROW = 8 the columns from 8 on of a unit that takes all columns go to psd_hi (HI floats
// per unit: 4, or 8 when the Doppler moments are columns 12 and 13).
template <int NC, int ROW, int HI>
__device__ __forceinline__ void integrate_chunk(float *lds, const LutView lv, const float gstep, const bool cut_ok, const float cut_e,
                                                const int ie, const int it, const int start, int cnt, const int *__restrict__ sorted,
                                                float4 *urec, const float2 *__restrict__ lam, float *__restrict__ psd_out,
                                                float *__restrict__ psd_hi, unsigned long long *__restrict__ work)
{
    constexpr int SLICE_FLOATS = 1024 * NC;      // one (e, t) node of the permuted table
    constexpr int LDS_ZERO = SLICE_FLOATS;       // 64 zero floats (all banks) for the lanes of the spare columns
    constexpr int LDS_X = SLICE_FLOATS + 64;
    constexpr int UPT = CPOL_CHUNK_UNITS / INTEG_THREADS;       // units per thread in the chunk-local sort
    static_assert(CPOL_CHUNK_UNITS == UPT * INTEG_THREADS, "whole units per thread");
    // chunk-local counting sort of the units by the number of quads (16 diameters) their N(D) survives
    // (its 65 counters borrow the start of the table area, which is filled only afterwards: with them in a
    // region of their own the workgroup would pass the LDS size at which three fit a CU)
    // The sorted order goes to global memory as one 16-byte record per unit (slope, own quad count, slot), written
    // next to each other in cutoff order: a pass then fetches its 32 units with coalesced loads, no permutation
    // lives in LDS.
    int *s_hist = (int *)lds;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;

    // ---- D^nu first: it decides how much of the slice the chunk needs ----
    if (tid < 256) ((float4 *)(lds + LDS_X))[tid] = ((const float4 *)lv.xq)[tid];
    if (tid < 16) ((float4 *)(lds + LDS_ZERO))[tid] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < 68) s_hist[tid] = 0;
    // the first D^nu of every quad once more, next to each other: the bisection below reads them at whatever index
    // each lane has reached, and 16 floats apart those reads would share four LDS banks
    float *s_xq = lds + 128;
    if (tid >= 64 && tid < 128) s_xq[tid - 64] = lv.xq[(tid - 64) * 16];
    int myslot[UPT];
    float2 lu[UPT];
#pragma unroll
    for (int t = 0; t < UPT; t++) {
        myslot[t] = 0; lu[t] = make_float2(0.f, 0.f);
        if (tid + t * INTEG_THREADS < cnt) { myslot[t] = sorted[start + tid + t * INTEG_THREADS]; lu[t] = lam[myslot[t]]; }
    }
    __syncthreads();

    // ---- order the chunk's units by where their N(D) becomes an exact float32 zero ----
    // nqu = number of quads whose first D^nu does not exceed the unit's cutoff (bisection over 64 ascending values)
    int nqu[UPT], myrank[UPT];
#pragma unroll
    for (int t = 0; t < UPT; t++) {
        nqu[t] = 64; myrank[t] = 0;
        if (cut_ok) {
            const float xc = psd_cutoff_x(lu[t].x + lu[t].y, cut_e);
            int c = 0;
#pragma unroll
            for (int step = 64; step >= 1; step >>= 1)
                if (c + step <= 64 && s_xq[c + step - 1] <= xc) c += step;
            nqu[t] = c;
        }
        if (tid + t * INTEG_THREADS < cnt) myrank[t] = atomicAdd(&s_hist[nqu[t]], 1);
    }
    __syncthreads();
    if (wave == 0) {
        const int h = s_hist[lane];
        int incl = h;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { int o = __shfl_up(incl, m); if (lane >= m) incl += o; }
        const int n64 = s_hist[64];
        s_hist[lane] = incl - h;
        if (lane == 63) s_hist[64] = incl;
        // largest cutoff of the chunk = highest bin in use
        const unsigned long long used = __ballot(h > 0);
        if (lane == 0) s_hist[65] = n64 > 0 ? 64 : 63 - __clzll(used);
    }
    __syncthreads();
    const int nq_chunk = s_hist[65];
#pragma unroll
    for (int t = 0; t < UPT; t++)
        if (tid + t * INTEG_THREADS < cnt) {
            // the chunk's units in cutoff order, one 16-byte record each: slope (hi, lo), own quad count, slot
            urec[start + s_hist[nqu[t]] + myrank[t]] =
                make_float4(lu[t].x, lu[t].y, __int_as_float(nqu[t]), __int_as_float(myslot[t]));
        }
    __syncthreads();

    // ---- stage the quads the chunk needs: straight 16-byte copy of the pre-permuted image ----
    {
        const float4 *src = (const float4 *)(lv.table + (long)(ie * lv.n_t + it) * SLICE_FLOATS);
        float4 *dst = (float4 *)lds;
        // a quad is 16 * NC floats; the geometric path evaluates whole seed groups of four quads
        const int n4 = ((nq_chunk + 3) & ~3) * 4 * NC;
        constexpr int PER = SLICE_FLOATS / 4 / INTEG_THREADS;
        static_assert(SLICE_FLOATS % (4 * INTEG_THREADS) == 0, "slice must be a whole number of 16-byte rounds");
        // whole rounds of one 16-byte word per thread (the last one may reach past n4, never past the slice),
        // two loads in flight per thread
        const int rounds = (n4 + INTEG_THREADS - 1) / INTEG_THREADS;
        for (int i = 0; i < rounds; i += 2) {
            const float4 a = src[tid + i * INTEG_THREADS];
            const float4 b = src[tid + min(i + 1, PER - 1) * INTEG_THREADS];
            dst[tid + i * INTEG_THREADS] = a;
            if (i + 1 < rounds) dst[tid + (i + 1) * INTEG_THREADS] = b;
        }
    }
    __syncthreads();

    // ---- the passes: v_mfma_f32_4x4x1_16b_f32 ----
    // One instruction holds sixteen independent 4 x 4 x 1 products: block b = lane >> 2 takes A[b][i] from lane
    // 4 b + i, B[b][j] from lane 4 b + j and adds the outer product to D[b][i][j] = register i of lane 4 b + j.
    // Here a block is (ug, dr): a subgroup of four units (ug = lane >> 4, unit i = lane & 3 inside it) and a
    // diameter residue dr = (lane >> 2) & 3, so one instruction multiplies 16 units' N(D) values at four
    // diameters d = 16 q + 4 s + dr (step s of quad q) into four columns of the table, and NG = NC / 4
    // instructions (8 cycles each) take a set of 16 units through all columns: 24 matrix cycles per 64
    // (unit, diameter) terms with the twelve scattering components where the 16 x 16 x 4 shape, a quarter of
    // whose columns would be padding, needs 32.  The lane's B word of quad q and column group g is word
    // (4 q + dr) NC + 4 g + (lane & 3) of the image: its operands of the four steps in one 16-byte read, the
    // same word in the four subgroups (a broadcast), the 16 distinct words of a read in 16 different bank
    // quarters when NC = 12.  Every lane generates one N(D) value per step and set of 16 units, for the unit
    // (ug, lane & 3), so nothing is computed twice.  The four residues' partial sums (at most 256 float32
    // additions each over the whole axis) meet at the end of the pass.
    constexpr int NG = (NC + 3) / 4;
    const int ug = lane >> 4, dr = (lane >> 2) & 3, r4 = lane & 3;
    const float4 *B4 = (const float4 *)lds;
    const float4 *X4 = (const float4 *)(lds + LDS_X) + dr;
    int boff[NG];
#pragma unroll
    for (int g = 0; g < NG; g++) boff[g] = dr * NC + ((NC == CPOL_N_COMB && (dr & 2)) ? (4 * g + r4) ^ 4 : 4 * g + r4);
    // columns past NC (NC = 14: the last two of the fourth group): a zero word, the same for every quad
    const bool spare = 4 * (NG - 1) + r4 >= NC;
    const int bstride_last = spare ? 0 : 4 * NC;
    if (spare) boff[NG - 1] = LDS_ZERO / 4 + dr;
    const int npass = (cnt + 31) >> 5;
    unsigned wsum = 0, own = 0;
    // at the end of a pass the lane with residue dr stores the sums of unit (ug, dr) of either set; that unit's
    // record was read by the lanes (ug, any residue, dr)
    const bool dr_odd = dr & 1, dr_high = dr & 2;
    const int rec_lane = ug * 16 + dr * 5;
    auto mfma4 = [](float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); };

    // passes of 32 units in cutoff order, two sets of 16; the waves take them in snake order (0 1 2 ... 7 7 6 ... 0
    // ...) so that every wave gets cheap and expensive ones
    constexpr int NW = INTEG_THREADS / 64;
    for (int r8 = 0; r8 * NW < npass; r8++) {
        const int ps = __builtin_amdgcn_readfirstlane(r8 * NW + ((r8 & 1) ? NW - 1 - wave : wave));
        if (ps >= npass) continue;
        const int u0 = ps * 32;
        const int ua = u0 + ug * 4 + r4, ub = ua + 16;        // the lane's unit in either set
        // per unit lambda*log2(e) = hi + lo (two floats): exp(-lambda x) = exp2(-(hi x + lo x))
        float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
        if (ua < cnt) ra = urec[start + ua];
        if (ub < cnt) rb = urec[start + ub];
        const int nqa = __float_as_int(ra.z), nqb = __float_as_int(rb.z);      // quads the lane's units take part in
        const int slot_a = __shfl(__float_as_int(ra.w), rec_lane), slot_b = __shfl(__float_as_int(rb.w), rec_lane);
        const float ha = ra.x, la = ra.y, hb = rb.x, lb = rb.y;
        // Units come in cutoff order, so the first set ends no later than the second begins: both sets run up to
        // the quad count of the first set's last unit, the second alone from there to that of its own last unit.
        // Past the count of a set's first unit a lane sets its N(D) to zero once its own unit's count is reached:
        // a unit's sums cover its own quads whatever units share its pass.
        const int n_a = min(cnt - u0, 16), n_b = max(0, min(cnt - u0 - 16, 16));
        // (wave-uniform values, kept in scalar registers: the loops below are controlled by scalar instructions)
        const int a_lo = __builtin_amdgcn_readfirstlane(nqa);
        const int a_hi = __builtin_amdgcn_readlane(nqa, ((n_a - 1) >> 2) * 16 + ((n_a - 1) & 3));
        const int b_lo = n_b ? __builtin_amdgcn_readfirstlane(nqb) : 0;
        const int b_hi = n_b ? __builtin_amdgcn_readlane(nqb, ((n_b - 1) >> 2) * 16 + ((n_b - 1) & 3)) : 0;
        if (dr == 0) own += (unsigned)((ua < cnt ? nqa : 0) + (ub < cnt ? nqb : 0));      // the units' own quad counts
        f32x4 acca[NG], accb[NG];
#pragma unroll
        for (int g = 0; g < NG; g++) { acca[g] = f32x4{0.f, 0.f, 0.f, 0.f}; accb[g] = acca[g]; }
        if (gstep > 0.f) {
            // nu = 1 on a uniform diameter axis: N(D) is a geometric sequence.  A lane's four values of a quad
            // are 4 diameters apart (ratio R per unit), the same lane's values of the next quad 16 further
            // (ratio R^4): one exponential per unit seeds four quads, the rest are multiplications (at most five
            // roundings from the seed).  Seed groups are evaluated whole: one that is skipped starts beyond every
            // cutoff, so its seed is an exact zero.
            const float Ra = __builtin_amdgcn_exp2f(-(ra.x * gstep + ra.y * gstep)), Rb = __builtin_amdgcn_exp2f(-(rb.x * gstep + rb.y * gstep));
            const float R2a = Ra * Ra, R2b = Rb * Rb, R4a = R2a * R2a, R4b = R2b * R2b;
            const int nqa4 = (nqa + 3) & ~3, nqb4 = (nqb + 3) & ~3;
            const int a_hi4 = (a_hi + 3) & ~3, b_hi4 = (b_hi + 3) & ~3;
            wsum += (unsigned)a_hi4 * (unsigned)n_a + (unsigned)b_hi4 * (unsigned)n_b;
            // one seed group of one set: quads q0 .. q0 + 3
            auto seed_group = [&](int q0, float h, float l, float R, float R2, float R4, bool live, f32x4 *acc) {
                const float x0 = X4[q0 * 4].x;
                float n0 = __builtin_amdgcn_exp2f(__builtin_fmaf(-h, x0, -l * x0));
                n0 = live ? n0 : 0.f;          // a unit that has had its seed groups: zero seed, zero terms
                float n1 = n0 * R, n2 = n0 * R2;
                float n3 = n1 * R2;
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    float4 bq[NG];
#pragma unroll
                    for (int g = 0; g < NG; g++) bq[g] = B4[boff[g] + (q0 + m) * (g == NG - 1 ? bstride_last : 4 * NC)];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < NG; g++) acc[g] = mfma4(n0, bq[g].x, acc[g]);
#pragma unroll
                    for (int g = 0; g < NG; g++) acc[g] = mfma4(n1, bq[g].y, acc[g]);
#pragma unroll
                    for (int g = 0; g < NG; g++) acc[g] = mfma4(n2, bq[g].z, acc[g]);
#pragma unroll
                    for (int g = 0; g < NG; g++) acc[g] = mfma4(n3, bq[g].w, acc[g]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (m < 3) { n0 = n0 * R4; n1 = n1 * R4; n2 = n2 * R4; n3 = n3 * R4; }
                }
            };
            for (int q0 = 0; q0 < a_hi4; q0 += 4) {
                seed_group(q0, ha, la, Ra, R2a, R4a, q0 < nqa4, acca);
                if (n_b) seed_group(q0, hb, lb, Rb, R2b, R4b, true, accb);
            }
            for (int q0 = a_hi4; q0 < b_hi4; q0 += 4) seed_group(q0, hb, lb, Rb, R2b, R4b, q0 < nqb4, accb);
        } else {
            wsum += (unsigned)a_hi * (unsigned)n_a + (unsigned)b_hi * (unsigned)n_b;
            // Exponent argument -(hi + lo) x: the hi part by one fused multiply-add per value, the lo part (2^-24 of
            // it) with the quad's third abscissa instead of x itself - the difference, lo (x - x_mid), is below
            // the rounding of the argument at the quad's far end.  All N(D) values of the quad first (independent
            // registers), then the matrix instructions back to back: none waits for a transcendental result.
            auto quad2 = [&](int qd) {          // both sets
                float4 bq[NG];
#pragma unroll
                for (int g = 0; g < NG; g++) bq[g] = B4[boff[g] + qd * (g == NG - 1 ? bstride_last : 4 * NC)];
                const float4 xq = X4[qd * 4];
                const float xx[4] = {xq.x, xq.y, xq.z, xq.w};
                float na[4], nb_[4];
                {
                    float ta, tb, p0, p1, p2, p3, p4, p5, p6, p7;
                    // (the slopes enter negated through the instructions' source modifiers)
                    asm volatile("v_mul_f32 %0, -%12, %16\n\t"
                                 "v_mul_f32 %1, -%13, %16\n\t"
                                 "v_fma_f32 %2, -%10, %14, %0\n\t"
                                 "v_fma_f32 %3, -%11, %14, %1\n\t"
                                 "v_fma_f32 %4, -%10, %15, %0\n\t"
                                 "v_fma_f32 %5, -%11, %15, %1\n\t"
                                 "v_fma_f32 %6, -%10, %16, %0\n\t"
                                 "v_fma_f32 %7, -%11, %16, %1\n\t"
                                 "v_fma_f32 %8, -%10, %17, %0\n\t"
                                 "v_fma_f32 %9, -%11, %17, %1"
                                 : "=&v"(ta), "=&v"(tb), "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5),
                                   "=&v"(p6), "=&v"(p7)
                                 : "v"(ha), "v"(hb), "v"(la), "v"(lb), "v"(xx[0]), "v"(xx[1]), "v"(xx[2]), "v"(xx[3]));
                    na[0] = __builtin_amdgcn_exp2f(p0); nb_[0] = __builtin_amdgcn_exp2f(p1);
                    na[1] = __builtin_amdgcn_exp2f(p2); nb_[1] = __builtin_amdgcn_exp2f(p3);
                    na[2] = __builtin_amdgcn_exp2f(p4); nb_[2] = __builtin_amdgcn_exp2f(p5);
                    na[3] = __builtin_amdgcn_exp2f(p6); nb_[3] = __builtin_amdgcn_exp2f(p7);
                }
                if (qd >= a_lo) {              // wave-uniform: some units of the first set are past their own count
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) na[s4] = qd < nqa ? na[s4] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++) {
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        const float b = s4 == 0 ? bq[g].x : s4 == 1 ? bq[g].y : s4 == 2 ? bq[g].z : bq[g].w;
                        acca[g] = mfma4(na[s4], b, acca[g]);
                    }
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        const float b = s4 == 0 ? bq[g].x : s4 == 1 ? bq[g].y : s4 == 2 ? bq[g].z : bq[g].w;
                        accb[g] = mfma4(nb_[s4], b, accb[g]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            };
            auto quad1 = [&](int qd, float h, float l, int nq_own, int lo, f32x4 *acc) {      // one set
                float4 bq[NG];
#pragma unroll
                for (int g = 0; g < NG; g++) bq[g] = B4[boff[g] + qd * (g == NG - 1 ? bstride_last : 4 * NC)];
                const float4 xq = X4[qd * 4];
                const float t = -l * xq.z;
                float n[4] = {__builtin_amdgcn_exp2f(__builtin_fmaf(-h, xq.x, t)), __builtin_amdgcn_exp2f(__builtin_fmaf(-h, xq.y, t)),
                              __builtin_amdgcn_exp2f(__builtin_fmaf(-h, xq.z, t)), __builtin_amdgcn_exp2f(__builtin_fmaf(-h, xq.w, t))};
                if (qd >= lo) {
#pragma unroll
                    for (int s4 = 0; s4 < 4; s4++) n[s4] = qd < nq_own ? n[s4] : 0.f;
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s4 = 0; s4 < 4; s4++)
#pragma unroll
                    for (int g = 0; g < NG; g++) {
                        const float b = s4 == 0 ? bq[g].x : s4 == 1 ? bq[g].y : s4 == 2 ? bq[g].z : bq[g].w;
                        acc[g] = mfma4(n[s4], b, acc[g]);
                    }
                __builtin_amdgcn_sched_barrier(0);
            };
            if (n_b) {
                for (int qd = 0; qd < a_hi; qd++) quad2(qd);
                for (int qd = a_hi; qd < b_hi; qd++) quad1(qd, hb, lb, nqb, b_lo, accb);
            } else {
                for (int qd = 0; qd < a_hi; qd++) quad1(qd, ha, la, nqa, a_lo, acca);
            }
        }
        // The four residues' partial sums meet (rotations by eight and by four lanes inside each row of 16 lanes, an
        // order of additions that does not depend on the lane), then the lane with residue dr keeps register dr: unit
        // (ug, dr) of the set, column 4 g + (lane & 3).  A result row's 16-byte pieces are written by four neighbouring
        // lanes each; with twelve columns in rows of 16 floats the fourth piece is zeros, so that every row is written
        // whole.
        {
            auto finish = [&](int base, const f32x4 *acc, int uslot) {
                if (base >= cnt) return;                 // wave-uniform: the set is empty
                const bool live = base + ug * 4 + dr < cnt;
                float *row = psd_out + (long)uslot * ROW + r4;
#pragma unroll
                for (int g = 0; g < NG; g++) {
                    // (p_dr + p_dr+2) + (p_dr+1 + p_dr+3): the same value in all four lanes, whichever stores it.  One
                    // instruction per addition (the rotated operand through the DPP modifier), in one block: the
                    // instructions of a stage are independent, and a result is read three instructions after it is
                    // written (the wait states a DPP read of a fresh register needs); the leading s_nop covers the
                    // accumulators' way out of the matrix pipe.
                    float t[4];
                    asm volatile("s_nop 7\n\t"
                                 "v_add_f32_dpp %0, %4, %4 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %1, %5, %5 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %2, %6, %6 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %3, %7, %7 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                                 "s_nop 1\n\t"
                                 "v_add_f32_dpp %0, %0, %0 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %1, %1, %1 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %2, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "v_add_f32_dpp %3, %3, %3 row_ror:4 row_mask:0xf bank_mask:0xf\n\t"
                                 "s_nop 1"
                                 : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "=&v"(t[3])
                                 : "v"(acc[g][0]), "v"(acc[g][1]), "v"(acc[g][2]), "v"(acc[g][3]));
                    const float lo = dr_odd ? t[1] : t[0], hi = dr_odd ? t[3] : t[2];
                    const float val = dr_high ? hi : lo;
                    if (live) {
                        if (4 * g < ROW) row[4 * g] = val;
                        else psd_hi[(long)uslot * HI + (4 * g - ROW) + r4] = val;          // the columns from 8 on beside rows of eight floats
                    }
                }
                if (NG * 4 < ROW && live) {
#pragma unroll
                    for (int g = NG; g < ROW / 4; g++) row[4 * g] = 0.f;
                }
            };
            finish(u0, acca, slot_a);
            finish(u0 + 16, accb, slot_b);
        }
    }
    if (work) {                                       // wave-uniform count of evaluated unit x quad steps
        for (int off = 32; off > 0; off >>= 1) own += __shfl_xor(own, off);
        const int slot = ((blockIdx.x * NW + wave) & (CPOL_WORK_SLOTS - 1)) * CPOL_WORK_STRIDE;
        if (lane == 0) {
            atomicAdd(&work[slot], (unsigned long long)wsum);
            atomicAdd(&work[2 * CPOL_WORK_SLOTS * CPOL_WORK_STRIDE + slot], (unsigned long long)own);
        }
    }
}

// The chunks of the buckets [b_lo, b_hi).  NCA: columns of the image that planes [0, n_cls) are integrated over, the
// image of the combined columns when A_COMB; NCB: 0, or the columns of the image of the planes [n_cls, 2 n_cls) when
// one launch takes both forms of polarimetric unit (ClassTable::comb: NCA = 8 combined columns, NCB = 12).  ROW, HI:
// see integrate_chunk.  Three workgroups per CU up to 12 columns (48 KB slice), two with 14.
template <int NCA, int NCB, int ROW, int HI, bool A_COMB>
__global__ __launch_bounds__(INTEG_THREADS, ((NCA > NCB ? NCA : NCB) <= 12 ? 3 : 2) * (INTEG_THREADS / 256))
void k_integrate(ClassTable ct, int b_lo, int b_hi, const int *__restrict__ offs, const int *__restrict__ chunk_off,
                 const int *__restrict__ chunk_bucket, const int *__restrict__ sorted, float4 *urec, const float2 *__restrict__ lam,
                 float *__restrict__ psd_out, float *__restrict__ psd_hi, unsigned long long *__restrict__ work)
{
    constexpr int NCMAX = NCA > NCB ? NCA : NCB;
    __shared__ __attribute__((aligned(16))) float lds[1024 * NCMAX + 64 + 1024];

    // XCD-aware chunk order: the workgroups of one XCD (blockIdx % 8) take runs of INTEG_XCD_RUN consecutive
    // chunks, i.e. mostly the same table slice, so a slice is fetched into one L2; the runs go round the XCDs
    // because a chunk's cost depends on its class and node (cutoffs), and a contiguous eighth of the chunk
    // list per XCD would leave some XCDs with all the expensive ones.
    const int c_lo = chunk_off[b_lo];
    const unsigned total = (unsigned)(chunk_off[b_hi] - c_lo);
    const unsigned xcd = blockIdx.x % 8, kx = blockIdx.x / 8;
    const unsigned rel = ((kx / INTEG_XCD_RUN) * 8 + xcd) * INTEG_XCD_RUN + kx % INTEG_XCD_RUN;
    if (rel >= total) return;
    const int cid = c_lo + (int)rel;

    const int bkt = chunk_bucket[cid];              // bucket of this chunk (k_fill_chunks)
    const int start = offs[bkt] + (cid - chunk_off[bkt]) * CPOL_CHUNK_UNITS;
    int cnt = offs[bkt + 1] - start;
    if (cnt > CPOL_CHUNK_UNITS) cnt = CPOL_CHUNK_UNITS;

    const int it = bkt % ct.nt_max;
    const int ie = (bkt / ct.nt_max) % ct.ne_max;
    const int plane = bkt / (ct.nt_max * ct.ne_max);       // form * n_cls + class
    const int h = plane % ct.n_cls;
    const float gstep = ct.pol_geo_step[h];                // 4 x diameter step when N(D) is geometric
    const bool cut_ok = ct.cut_ok[h];                      // D^nu ascending: N(D) = 0 past a unit's cutoff
    const float cut_e = ct.cut_e[h];
    if (NCB && plane >= ct.n_cls)
        integrate_chunk<NCB ? NCB : NCA, ROW, HI>(lds, ct.lut[h], gstep, cut_ok, cut_e, ie, it, start, cnt, sorted, urec, lam, psd_out, psd_hi, work);
    else
        integrate_chunk<NCA, ROW, HI>(lds, A_COMB ? ct.lut_comb[h] : ct.lut[h], gstep, cut_ok, cut_e, ie, it, start, cnt, sorted, urec, lam, psd_out,
                                      psd_hi, work);
}

// nb: number of polarimetric buckets (the separate Doppler units' buckets follow them).
void launch_integrate(hipStream_t s, const ClassTable &ct, long max_chunks, int nb, const int *offs,
                      const int *chunk_off, const int *chunk_bucket, const int *sorted, float4 *urec, const float2 *lam,
                      float *psd_out, float *psd_hi, unsigned long long *work)
{
    if (max_chunks <= 0) return;
    max_chunks = (max_chunks + 8 * INTEG_XCD_RUN - 1) / (8 * INTEG_XCD_RUN) * (8 * INTEG_XCD_RUN);   // whole rounds of runs
    const dim3 grid((unsigned)max_chunks), block(INTEG_THREADS);
    const int nbf = ct.n_cls * ct.ne_max * ct.nt_max;          // buckets of one form
    if (ct.comb && !ct.dop_shared)
        hipLaunchKernelGGL((k_integrate<CPOL_N_COMB, 12, CPOL_N_COMB, 4, true>), grid, block, 0, s, ct, 0, nb, offs, chunk_off, chunk_bucket,
                           sorted, urec, lam, psd_out, psd_hi, work);
    else if (ct.comb) {
        // the Doppler moments ride on the polarimetric units: the combined image carries the second one, the units
        // that take all columns take the 14-column image, whose slice lets two workgroups share a CU where the
        // combined one lets three - so the two forms are launches of their own (the second nearly empty)
        hipLaunchKernelGGL((k_integrate<CPOL_N_COMB, 0, CPOL_N_COMB, 8, true>), grid, block, 0, s, ct, 0, nbf, offs, chunk_off, chunk_bucket,
                           sorted, urec, lam, psd_out, psd_hi, work);
        hipLaunchKernelGGL((k_integrate<14, 0, CPOL_N_COMB, 8, false>), grid, block, 0, s, ct, nbf, nb, offs, chunk_off, chunk_bucket,
                           sorted, urec, lam, psd_out, psd_hi, work);
    }
    else if (ct.ncol == 12)
        hipLaunchKernelGGL((k_integrate<12, 0, CPOL_N_COL, 0, false>), grid, block, 0, s, ct, 0, nb, offs, chunk_off, chunk_bucket, sorted, urec,
                           lam, psd_out, psd_hi, work);
    else
        hipLaunchKernelGGL((k_integrate<14, 0, CPOL_N_COL, 0, false>), grid, block, 0, s, ct, 0, nb, offs, chunk_off, chunk_bucket, sorted, urec,
                           lam, psd_out, psd_hi, work);
}

// =============================================================================
// Doppler scheme 2 for units whose PSD differs from the polarimetric one (2-moment input,
// doppler_sz.py:57,192,218-220): only the two moments  sum_d N rcs  and  sum_d N v rcs  are needed,
// two columns where the matrix-core tile has sixteen, so these units run on the vector ALUs:
// one lane per unit, the node's 12 KB compact image in LDS read as wave-uniform 16-byte words,
// packed-f32 exponent and accumulate, float64 flush every 128 diameters like k_integrate.
// Buckets [b_lo, b_hi) of the chunk table.
// =============================================================================
#define DOP_THREADS (CPOL_CHUNK_UNITS / 2)      // two units per lane: every LDS word read serves both

// plain path of one unit: one exponential per diameter, over the first nq8 quads (a multiple of 8) and no further
// than the unit's own quad count; the two moments of a diameter are accumulated as one packed pair
__device__ inline void dop_unit_plain(const float *lds, float2 l, int nq8, int own, double *o12, double *o13)
{
    const f32x2 *BB = (const f32x2 *)(lds + 1024);
    double s12 = 0., s13 = 0.;
    for (int d0 = 0; d0 < 16 * nq8; d0 += 128) {
        f32x2 a = {0.f, 0.f};
#pragma unroll 8
        for (int d = d0; d < d0 + 128; d++) {
            const float x = lds[d];
            float n = __builtin_amdgcn_exp2f(__builtin_fmaf(-l.x, x, -l.y * x));
            n = d < 16 * own ? n : 0.f;          // the unit's own quads only
            a = __builtin_elementwise_fma((f32x2){n, n}, BB[d], a);
        }
        s12 += (double)a.x;
        s13 += (double)a.y;
    }
    *o12 = s12; *o13 = s13;
}

__global__ __launch_bounds__(DOP_THREADS, 6)
void k_integrate_dop(ClassTable ct, int b_lo, int b_hi, const int *__restrict__ offs, const int *__restrict__ chunk_off,
                     const int *__restrict__ chunk_bucket, const int *__restrict__ sorted, const float2 *__restrict__ lam,
                     const double *__restrict__ n0d, float2 *__restrict__ dop_out, long slot0,
                     unsigned long long *__restrict__ work)
{
    __shared__ __attribute__((aligned(16))) float lds[CPOL_DOP_SLICE];
    __shared__ int s_hist[68];
    __shared__ float s_xq[64];           // first D^nu - D_0^nu of every quad, next to each other for the bisection
    // the chunk's units in cutoff order: slope (hi, lo), slot, quad count - the integration reads its units from
    // here, not through the index list again
    __shared__ float4 s_u[CPOL_CHUNK_UNITS];
    const int c_lo = chunk_off[b_lo];
    const unsigned total = (unsigned)(chunk_off[b_hi] - c_lo);
    // runs of INTEG_XCD_RUN consecutive chunks go round the XCDs, as in k_integrate
    const unsigned xcd = blockIdx.x % 8, kx = blockIdx.x / 8;
    const unsigned rel = ((kx / INTEG_XCD_RUN) * 8 + xcd) * INTEG_XCD_RUN + kx % INTEG_XCD_RUN;
    if (rel >= total) return;
    const int cid = c_lo + (int)rel;
    const int lo = chunk_bucket[cid];               // bucket of this chunk (k_fill_chunks)
    const int bkt = lo;
    const int rel0 = (cid - chunk_off[bkt]) * CPOL_CHUNK_UNITS;
    int cnt = offs[bkt + 1] - offs[bkt] - rel0;
    if (cnt > CPOL_CHUNK_UNITS) cnt = CPOL_CHUNK_UNITS;
    // a Doppler bucket holds the twins of the polarimetric bucket b_lo below it, in the same order: its units
    // are that bucket's entries of the sorted list moved up by slot0 (k_scatter writes no list of their own)
    const int start = offs[bkt - b_lo] + rel0;
    const int it = bkt % ct.nt_max;
    const int ie = (bkt / ct.nt_max) % ct.ne_max;
    const int h = (bkt / (ct.nt_max * ct.ne_max) - ct.n_pol_planes) % ct.n_cls;       // twin planes follow the polarimetric ones, form by form
    const LutView lv = ct.lut_dop[h];
    const int tid = threadIdx.x;
    {
        const float4 *src = (const float4 *)(lv.table + (long)(ie * lv.n_t + it) * CPOL_DOP_SLICE);
        float4 *dst = (float4 *)lds;
#pragma unroll
        for (int i = tid; i < CPOL_DOP_SLICE / 4; i += DOP_THREADS) dst[i] = src[i];
    }
    if (tid < 68) s_hist[tid] = 0;
    if (tid >= 64 && tid < 128) s_xq[tid - 64] = (lv.table + (long)(ie * lv.n_t + it) * CPOL_DOP_SLICE)[(tid - 64) * 16];
    __syncthreads();
    // ---- order the chunk's units by how many quads of 16 diameters they need (as k_integrate does): none for
    // a unit whose N0 is an exact zero (its moments are multiplied by it: the 1-moment snow intercept fed
    // with a number concentration for a temperature, doppler_sz.py:57, underflows for most gates), else up to
    // where its N(D) becomes an exact float32 zero.  A wavefront stops at the last quad any of its 128 units
    // needs. ----
    const bool cut_ok = ct.dop_cut_ok[h];
    int nqu[2], myrank[2], sl_[2];
    float2 lt_[2];
    double n0_[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {          // index, then slope and N0 side by side, for both units before any use
        const int i = tid + t * DOP_THREADS;
        sl_[t] = i < cnt ? sorted[start + i] + (int)slot0 : 0;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int i = tid + t * DOP_THREADS;
        lt_[t] = i < cnt ? lam[sl_[t]] : make_float2(0.f, 0.f);
        n0_[t] = i < cnt ? n0d[sl_[t]] : 0.;
    }
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int i = tid + t * DOP_THREADS;
        nqu[t] = 64; myrank[t] = 0;
        if (i < cnt) {
            if (n0_[t] == 0.) nqu[t] = 0;
            else if (cut_ok) {
                const float2 lt = lt_[t];
                const float xc = psd_cutoff_x(lt.x + lt.y, ct.dop_cut_e[h]);
                int c = 0;
#pragma unroll
                for (int step = 64; step >= 1; step >>= 1)
                    if (c + step <= 64 && s_xq[c + step - 1] <= xc) c += step;
                nqu[t] = c;
            }
            myrank[t] = atomicAdd(&s_hist[nqu[t]], 1);
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int hh = s_hist[tid];
        int incl = hh;
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) { int o = __shfl_up(incl, m); if (tid >= m) incl += o; }
        s_hist[tid] = incl - hh;
        if (tid == 63) s_hist[64] = incl;
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int i = tid + t * DOP_THREADS;
        if (i < cnt)
            s_u[s_hist[nqu[t]] + myrank[t]] = make_float4(lt_[t].x, lt_[t].y, __int_as_float(sl_[t]), __int_as_float(nqu[t]));
    }
    __syncthreads();
    // a wavefront takes 128 consecutive positions of that order, a lane positions p and p + 64 of them
    const int wbase = (tid >> 6) * 128, lane = tid & 63;
    const int pos[2] = {wbase + lane, wbase + 64 + lane};
    const bool have[2] = {pos[0] < cnt, pos[1] < cnt};
    if (wbase >= cnt) return;
    const int nq8 = __builtin_amdgcn_readfirstlane((__float_as_int(s_u[min(wbase + 127, cnt - 1)].w) + 7) & ~7);
    if (work && lane == 0)
        atomicAdd(&work[(CPOL_WORK_SLOTS + ((blockIdx.x * (DOP_THREADS / 64) + (tid >> 6)) & (CPOL_WORK_SLOTS - 1))) * CPOL_WORK_STRIDE],
                  (unsigned long long)nq8 * (unsigned)min(cnt - wbase, 128));
    int slot[2], own[2];
    float2 l[2];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const float4 ur = have[t] ? s_u[pos[t]] : make_float4(0.f, 0.f, 0.f, 0.f);
        slot[t] = __float_as_int(ur.z);
        own[t] = __float_as_int(ur.w);          // quads the unit itself takes part in
        l[t] = make_float2(ur.x, ur.y);
    }
    // below the exact-zero rule the terms past a unit's own count are small, not zero: they are left out per unit,
    // so that a unit's sums do not depend on which units share its wavefront
    const int own_lo = __builtin_amdgcn_readfirstlane(__float_as_int(s_u[wbase].w));
    if (nq8 == 0) {          // nothing to integrate for any unit of the wavefront
#pragma unroll
        for (int t = 0; t < 2; t++)
            if (have[t]) dop_out[(long)slot[t] - slot0] = make_float2(0.f, 0.f);
        return;
    }
    // The image words of a quad are the same for every lane of the wavefront: they are read from the image in memory
    // as wave-uniform scalar loads (constant address space) into scalar registers, not as 64 copies through LDS; LDS
    // keeps its copy of the node for the bisection above.
    typedef const __attribute__((address_space(4))) f32x4 cf4_t;
    typedef const __attribute__((address_space(4))) float cf1_t;
    const float *node_img = lv.table + (long)(ie * lv.n_t + it) * CPOL_DOP_SLICE;
    cf1_t *GX = (cf1_t *)(unsigned long long)node_img;
    cf4_t *BB4 = (cf4_t *)(unsigned long long)(node_img + 1024);      // word j of quad q: the moment pairs of diameters 16 q + 2 j, + 1
    double s12[2] = {0., 0.}, s13[2] = {0., 0.};
    const float gstep = ct.dop_geo_step[h];      // diameter step when N(D) is geometric, else 0
    if (gstep > 0.f) {
        // nu = 1 on a uniform axis: N(D_0 + i step) = N(D_0) R^i, so a quad's sum over 16 diameters is N(quad start)
        // times the polynomial sum_i B_i R^i, evaluated by Horner's rule on the packed pair of moments: one packed
        // FMA per diameter and unit.  N(quad start) comes from an exponential every fourth quad and R^16 in between.
        f32x2 Rv[2];
        float R16[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const float arg = l[t].x * gstep + l[t].y * gstep;
            const float R = __builtin_amdgcn_exp2f(-arg);
            Rv[t] = (f32x2){R, R};
            R16[t] = __builtin_amdgcn_exp2f(-16.f * arg);
        }
        for (int qo = 0; qo < nq8; qo += 8) {
            f32x2 acc[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int q4 = 0; q4 < 8; q4 += 4) {
                const float xs0 = GX[16 * (qo + q4)];
                float nq_[2];
#pragma unroll
                for (int t = 0; t < 2; t++) nq_[t] = __builtin_amdgcn_exp2f(__builtin_fmaf(-l[t].x, xs0, -l[t].y * xs0));
#pragma unroll
                for (int m = 0; m < 4; m++) {
                    cf4_t *w = BB4 + 8 * (qo + q4 + m);
                    f32x2 S[2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
                    for (int j = 7; j >= 0; j--) {
                        const f32x4 b = w[j];
                        const f32x2 hi = {b.z, b.w}, lo = {b.x, b.y};
#pragma unroll
                        for (int t = 0; t < 2; t++) {
                            S[t] = __builtin_elementwise_fma(S[t], Rv[t], hi);
                            S[t] = __builtin_elementwise_fma(S[t], Rv[t], lo);
                        }
                    }
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        float nn = nq_[t];
                        if (qo + q4 + m >= own_lo) nn = qo + q4 + m < own[t] ? nn : 0.f;      // wave-uniform test first
                        acc[t] = __builtin_elementwise_fma((f32x2){nn, nn}, S[t], acc[t]);
                        nq_[t] *= R16[t];
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 2; t++) { s12[t] += (double)acc[t].x; s13[t] += (double)acc[t].y; }
        }
    } else {
        for (int t = 0; t < 2; t++)
            if (have[t]) dop_unit_plain(lds, l[t], min(nq8, (own[t] + 7) & ~7), own[t], &s12[t], &s13[t]);
    }
    // unit-spacing trapezoid rule (doppler_sz.py:219-220): the sum minus half of the two end terms
    const float x0 = lds[0], xl = lds[1023];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        if (!have[t]) continue;
        const float na = __builtin_amdgcn_exp2f(__builtin_fmaf(-l[t].x, x0, -l[t].y * x0));
        const float nz = __builtin_amdgcn_exp2f(__builtin_fmaf(-l[t].x, xl, -l[t].y * xl));
        const double r12 = s12[t] - 0.5 * ((double)na * (double)lds[1024] + (double)nz * (double)lds[1024 + 2 * 1023]);
        const double r13 = s13[t] - 0.5 * ((double)na * (double)lds[1025] + (double)nz * (double)lds[1025 + 2 * 1023]);
        dop_out[(long)slot[t] - slot0] = make_float2((float)r12, (float)r13);     // indexed like the polarimetric twin
    }
}

void launch_integrate_dop(hipStream_t s, const ClassTable &ct, long max_chunks, int b_lo, int b_hi, const int *offs,
                          const int *chunk_off, const int *chunk_bucket, const int *sorted, const float2 *lam,
                          const double *n0d, float2 *dop_out, long slot0, unsigned long long *work)
{
    if (max_chunks <= 0) return;
    max_chunks = (max_chunks + 8 * INTEG_XCD_RUN - 1) / (8 * INTEG_XCD_RUN) * (8 * INTEG_XCD_RUN);   // whole rounds of runs
    hipLaunchKernelGGL(k_integrate_dop, dim3((unsigned)max_chunks), dim3(DOP_THREADS), 0, s, ct, b_lo, b_hi, offs,
                       chunk_off, chunk_bucket, sorted, lam, n0d, dop_out, slot0, work);
}

// =============================================================================
// per-ray reduction: GH-weighted sub-beam sum with the reference's float32 storage
// (scattering_sz.py:38,105), class sum (:109-110), observables (:164-196), the two
// range scans (:117-125) as wavefront scans in the reference's float32 order, mask (:144-151), Doppler
// scheme 1/2 averaging (doppler_sz.py:62-87), sensitivity cut (utilities.py:68-105).
// Two kernels: k_gate_reduce (one thread per gate) and k_ray_scan (one wavefront per ray).
// =============================================================================
// Doppler scheme 2 (doppler_sz.py:155-231) for one (ray, sub-beam, gate): RCS- and PSD-weighted mean fall
// speed over the classes from the two extra columns of the PSD integration, unit-spacing trapezoid rule
// (sum minus half of the end terms, :219-220), air-density correction and projection on the beam (:23,231).
// SHARED: the two moments ride on the polarimetric units (1-moment input); COMB: combined columns in use.
template <bool COMB, bool SHARED>
__device__ inline double doppler2_radial(const Dop2In &d2, const cpol_key_t *__restrict__ key, const float *__restrict__ psd_out,
                                         long nu, long u, long rb, int G, const unsigned char *__restrict__ uflag)
{
    double vh = 0., nn = 0.;
    const unsigned uf = uflag[u];                                       // bit h: class h holds a unit at the point
    if (!(uf & 15u)) return __builtin_nan("");       // no hydrometeors: the fall speed is 0 / 0 and the radial velocity a NaN whatever the wind is
    for (int h = 0; h < d2.n_cls; h++) {
        const long pslot = (long)h * nu + u;                            // polarimetric unit of the class
        const long slot = SHARED ? pslot : pslot + (long)d2.n_cls * nu;      // unit carrying the Doppler PSD
        if (!((uf >> h) & 1u)) continue;
        int k = SHARED ? (int)key[pslot] : 0;       // the table node, for the end terms of the columns of the matrix kernel
        double s12, s13;
        if (SHARED) {
            if (!COMB) { s12 = (double)psd_out[pslot * CPOL_N_COL + 12]; s13 = (double)psd_out[pslot * CPOL_N_COL + 13]; }
            else if (k < d2.nb_form) {
                // a unit in the combined form: column 7 holds the second moment, the first is 2 pi times column 0
                s12 = (double)(float)(2 * M_PI) * (double)psd_out[pslot * CPOL_N_COMB];
                s13 = (double)psd_out[pslot * CPOL_N_COMB + 7];
            }
            else { s12 = (double)d2.psd_hi[pslot * 8 + 4]; s13 = (double)d2.psd_hi[pslot * 8 + 5]; }      // columns 12, 13
        }
        else { const float2 t = d2.dop_out[pslot]; s12 = (double)t.x; s13 = (double)t.y; }
        if (SHARED) {
            // the MFMA columns hold plain sums: take half of the end terms off (unit-spacing trapezoid rule)
            const int it = k % d2.nt_max, ie = (k / d2.nt_max) % d2.ne_max;
            const float *img = d2.img[h] + (long)(ie * d2.n_t[h] + it) * d2.slice_floats;
            const float2 l = d2.lam[slot];
            const double l2 = (double)l.x + (double)l.y;
            const double e0 = exp2(-l2 * (double)d2.xq[h][0]), e1 = exp2(-l2 * (double)d2.xq[h][1023]);
            s12 -= 0.5 * (e0 * (double)img[d2.at0[0]] + e1 * (double)img[d2.at1[0]]);
            s13 -= 0.5 * (e0 * (double)img[d2.at0[1]] + e1 * (double)img[d2.at1[1]]);
        }
        double n0 = d2.n0d[slot];
        double a = n0 * s13, b = n0 * s12;
        vh += (a != a ? 0. : a);
        nn += (b != b ? 0. : b);
    }
    double vhyd = vh / nn;
    float rho = d2.vals[(long)CPOL_V_RHO * d2.plane + u];
    float rho0 = d2.vals[(long)CPOL_V_RHO * d2.plane + rb * G];
    vhyd *= (double)sqrtf(rho / rho0);
    float th = d2.elev[u] * (float)(M_PI / 180.0);
    double U = d2.vals[(long)CPOL_V_U * d2.plane + u], V = d2.vals[(long)CPOL_V_V * d2.plane + u];
    double W = d2.vals[(long)CPOL_V_W * d2.plane + u];
    return (U * d2.phi_sc[2 * rb] + V * d2.phi_sc[2 * rb + 1]) * (double)cosf(th) + (W - vhyd) * (double)sinf(th);
}

// Stage 1, one thread per (ray, gate): sub-beam and class sums, observables without the range scans.
// Launch shape by measurement (whole volume | an eighth of it, ms): 256 threads x 6 wavefronts per SIMD 1.18 | 0.236,
// 64 x 6 1.12 | 0.230, 64 x 8 1.20 | 0.207, 64 x 7 1.10 | 0.195: small workgroups drain evenly at the end of the grid, and
// the threads wait on dependent loads, so a seventh wavefront per SIMD is worth the 72-register limit.
#define GR_THREADS 64
#define GR_OCC 7
// COMB: combined columns in use (rows of eight floats, two forms of unit); SHARED: the Doppler scheme-2 moments ride on
// the polarimetric units.
template <bool COMB, bool SHARED>
__global__ __launch_bounds__(GR_THREADS, GR_OCC)
void k_gate_reduce(ObsParams op, int n_rays, long nu, const cpol_key_t *__restrict__ key, const double *__restrict__ coef,
                   const float *__restrict__ psd_out, const float *__restrict__ psd_hi, const signed char *__restrict__ mask,
                   const double *__restrict__ wsub, const double *__restrict__ vr, float *__restrict__ obs,
                   long obs_plane, long ray0, double *__restrict__ mask_out, double *__restrict__ rvel_out, Dop2In d2)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int G = op.n_gates, B = op.n_sub;
    if (t >= (long)n_rays * G) return;
    const int r = (int)(t / G), g = (int)(t % G);
    const float qnan = __builtin_nanf("");
    // the twelve sums of one unit: a row of 16 floats, or (combined columns in use) eight floats and four beside them
    auto twelve = [&](long slot, float *vv) {
        const float4 *p = (const float4 *)(psd_out + slot * op.row);
        const float4 v0 = p[0], v1 = p[1], v2 = COMB ? *(const float4 *)(psd_hi + slot * op.hi) : p[2];
        vv[0] = v0.x; vv[1] = v0.y; vv[2] = v0.z; vv[3] = v0.w; vv[4] = v1.x; vv[5] = v1.y; vv[6] = v1.z; vv[7] = v1.w;
        vv[8] = v2.x; vv[9] = v2.y; vv[10] = v2.z; vv[11] = v2.w;
    };
    // The quantities the observables are made of (scattering_sz.py:168-194):
    //   bq = s0 - s1 - s2 + s3, cq = s0 + s1 + s2 + s3, a1 = s4 + s7, a2 = s6 - s5, kd = s10 - s8, s11, s9
    float bq, cq, a1, a2, kd, s11, s9;
    // Does the gate hold a unit in the combined form (the PSD set-up marks its point with CPOL_MASK_COMB)?  The twelve
    // components of such a gate are never formed: the unit adds to each of them an amount of the size of a normal
    // float32 (ClassTable::comb), so scattering_sz.py:110 finds no exact zero to turn into NaN unless contributions
    // cancel exactly, and the gate's combinations are the sums of its units'.
    bool comb_gate = false;
    double msum = 0;
    for (int b = 0; b < B; b++) {
        const int m = mask[((long)r * B + b) * G + g];
        comb_gate = comb_gate || m == CPOL_MASK_COMB;
        msum += m == CPOL_MASK_COMB ? 0. : (double)m;
    }
    if (COMB && comb_gate) {
        // sub-beam after sub-beam, the classes of a point by its flags
        double C[7];
#pragma unroll
        for (int k = 0; k < 7; k++) C[k] = 0.;
        for (int b = 0; b < B; b++) {
            const long u = ((long)r * B + b) * G + g;
            const unsigned uf = op.uflag[u];         // bit h: class h holds a unit at the point; bit 4 + h: in the combined form
            for (int h = 0; h < op.n_cls; h++) {
                if (!((uf >> h) & 1u)) continue;
                const long slot = (long)h * nu + u;
                const double c = coef[slot];
                double gq[7];
                if ((uf >> (4 + h)) & 1u) {
                    const float4 *p = (const float4 *)(psd_out + slot * CPOL_N_COMB);
                    const float4 v0 = p[0], v1 = p[1];
                    gq[0] = v0.x; gq[1] = v0.y; gq[2] = v0.z; gq[3] = v0.w; gq[4] = v1.x; gq[5] = v1.y; gq[6] = v1.z;
                } else {
                    float vv[CPOL_N_SZ];
                    twelve(slot, vv);
                    const double z0 = vv[0], z1 = vv[1], z2 = vv[2], z3 = vv[3];
                    gq[0] = ((z0 - z1) - z2) + z3; gq[1] = ((z0 + z1) + z2) + z3;
                    gq[2] = (double)vv[4] + (double)vv[7]; gq[3] = (double)vv[6] - (double)vv[5];
                    gq[4] = (double)vv[10] - (double)vv[8]; gq[5] = vv[11]; gq[6] = vv[9];
                }
#pragma unroll
                for (int k = 0; k < 7; k++) {
                    const double v = gq[k] * c;
                    C[k] += (v != v ? 0. : v);           // nansum_arr: a NaN counts as zero (utilities.py:46-66)
                }
            }
        }
        bq = (float)C[0]; cq = (float)C[1]; a1 = (float)C[2]; a2 = (float)C[3]; kd = (float)C[4]; s11 = (float)C[5]; s9 = (float)C[6];
    } else {
        // every component as the reference accumulates it: float32 storage after every sub-beam (scattering_sz.py:38,
        // 105), float32 class sum (:109), exact zeros to NaN (:110)
        float tot[CPOL_N_SZ];
#pragma unroll
        for (int k = 0; k < CPOL_N_SZ; k++) tot[k] = 0.f;
        for (int h = 0; h < op.n_cls; h++) {
            float acc[CPOL_N_SZ];
#pragma unroll
            for (int k = 0; k < CPOL_N_SZ; k++) acc[k] = qnan;
            for (int b = 0; b < B; b++) {
                long u = ((long)r * B + b) * G + g;
                long slot = (long)h * nu + u;
                if (key[slot] < 0) continue;
                double c = coef[slot];
                float vv[CPOL_N_SZ];
                twelve(slot, vv);
#pragma unroll
                for (int k = 0; k < CPOL_N_SZ; k++) {
                    double v = (double)vv[k] * c;
                    double sm = (acc[k] != acc[k] ? 0. : (double)acc[k]) + (v != v ? 0. : v);
                    acc[k] = (float)sm;
                }
            }
#pragma unroll
            for (int k = 0; k < CPOL_N_SZ; k++) tot[k] += (acc[k] != acc[k]) ? 0.f : acc[k];
        }
        float s[CPOL_N_SZ];
#pragma unroll
        for (int k = 0; k < CPOL_N_SZ; k++) s[k] = (tot[k] == 0.f) ? qnan : tot[k];
        // float32 in the reference's order
        bq = ((s[0] - s[1]) - s[2]) + s[3];
        cq = ((s[0] + s[1]) + s[2]) + s[3];
        a1 = s[4] + s[7]; a2 = s[6] - s[5];
        kd = s[10] - s[8];
        s11 = s[11]; s9 = s[9];
    }
    double rv = __builtin_nan(""), sw = 0;
    for (int b = 0; b < B; b++) {
        long u = ((long)r * B + b) * G + g;
        if (op.doppler == 1 || op.doppler == 2) {
            double pw = op.doppler == 2 ? doppler2_radial<COMB, SHARED>(d2, key, psd_out, nu, u, ((long)r * B + b), G, op.uflag) : vr[u];
            double w = wsub[b];
            sw += (pw != pw) ? 0. : w;
            double tt = pw * w;
            rv = (rv != rv ? 0. : rv) + (tt != tt ? 0. : tt);
        }
    }
    // get_pol_from_sz, float32
    float xs_h = op.two_pi * bq, xs_v = op.two_pi * cq;
    float aq = a1 * a1 + a2 * a2;
    float delta = atan2f(-a2, -a1);             // arctan2(s5 - s6, -s4 - s7): both arguments are exact negations
    const long o = (ray0 + r) * (long)G + g;
    obs[CPOL_O_ZH * obs_plane + o] = op.zfac * xs_h;
    obs[CPOL_O_ZV * obs_plane + o] = op.zfac * xs_v;
    obs[CPOL_O_ZDR * obs_plane + o] = xs_h / xs_v;
    obs[CPOL_O_KDP * obs_plane + o] = op.kdpfac * kd + delta;        // KDP + delta_hv (scattering_sz.py:117)
    obs[CPOL_O_DELTA_HV * obs_plane + o] = delta;
    obs[CPOL_O_RHOHV * obs_plane + o] = sqrtf(aq / (bq * cq));
    obs[CPOL_O_AH * obs_plane + o] = op.afac * (op.two_wl * s11);
    obs[CPOL_O_AV * obs_plane + o] = op.afac * (op.two_wl * s9);
    if (mask_out) {
        double m = msum / (double)B;
        if (m >= 0 && m < 1) m = 0;
        mask_out[o] = m;
    }
    if (rvel_out && (op.doppler == 1 || op.doppler == 2)) rvel_out[o] = rv / sw;
}

// Stage 2, one wavefront per ray: PHIDP and the attenuation as sequential float32 scans along range (the
// reference's accumulation order and precision: acc_i = fl(acc_{i-1} (op) x_i), one rounding per gate, exactly what
// np.cumsum / np.cumprod do on a float32 array, utilities.py:18-24), then the sensitivity cut.
// The three scans (sum of 2 KDP, products of the two attenuation factors) of a chunk of 64 gates run side by side
// on lanes 0..2, each as a chain of 64 fused multiply-adds acc = fma(acc, m, a) over operands the 64 lanes put into
// LDS: (m, a) = (1, x) makes the step the correctly rounded sum acc + x, (x, 0) the correctly rounded product acc x
// (factors and products are positive, so no signed zero is at stake).
__global__ __launch_bounds__(64)
void k_ray_scan(ObsParams op, float *__restrict__ obs, long obs_plane, long ray0,
                double *__restrict__ rvel_out, double *__restrict__ spectrum, int n_vel)
{
    __shared__ float2 s_op[3][64];
    __shared__ float s_pre[3][64];
    const int r = blockIdx.x, lane = threadIdx.x;
    const int G = op.n_gates;
    const float qnan = __builtin_nanf("");
    float carry = lane == 0 ? 0.f : 1.f;          // lane 0: running sum; lanes 1, 2: running products
    for (int g0 = 0; g0 < G; g0 += 64) {
        const int g = g0 + lane;
        const bool act = g < G;
        const long o = (ray0 + r) * (long)G + (act ? g : 0);
        float zh = obs[CPOL_O_ZH * obs_plane + o], zv = obs[CPOL_O_ZV * obs_plane + o];
        const float kdpm = obs[CPOL_O_KDP * obs_plane + o];
        const float ah = obs[CPOL_O_AH * obs_plane + o], av = obs[CPOL_O_AV * obs_plane + o];
        // PHIDP = nan_cumsum(2 * KDP_m) * res / 1000  (scattering_sz.py:118)
        const float two_k = 2.f * kdpm;
        s_op[0][lane] = make_float2(1.f, (act && two_k == two_k) ? two_k : 0.f);
        if (op.att_corr) {
            // ZV *= nan_cumprod(10 ** (-0.1 * AV * res_km)), the same for ZH  (scattering_sz.py:120-123)
            // the factor is the correctly rounded float32 power (float64 evaluation, one rounding), what a libm
            // powf returns; NumPy's vectorised float32 power differs from it in the last bit for some 12 % of
            // arguments, depending on the CPU it runs on
            const float fh = (float)exp10((double)((-0.1f * ah) * op.res_km));
            const float fv = (float)exp10((double)((-0.1f * av) * op.res_km));
            s_op[1][lane] = make_float2((act && fh == fh) ? fh : 1.f, 0.f);
            s_op[2][lane] = make_float2((act && fv == fv) ? fv : 1.f, 0.f);
        }
        __syncthreads();
        if (lane == 0 || (op.att_corr && lane < 3)) {
            float acc = carry;
#pragma unroll 16
            for (int i = 0; i < 64; i++) {
                const float2 ma = s_op[lane][i];
                acc = fmaf(acc, ma.x, ma.y);
                s_pre[lane][i] = acc;
            }
            carry = acc;
        }
        __syncthreads();
        const float phidp = s_pre[0][lane] * op.res / 1000.f;
        bool att = false;
        float zdr = 0.f;
        if (op.att_corr) {
            zv *= s_pre[2][lane];
            zh *= s_pre[1][lane];
            zdr = zh / zv;
            att = true;
        }
        bool cut = false;
        if (op.sens_mode && act) {
            // utilities.py:74-81; mode 3 carries sensitivity + radar constant + SNR as one number
            double thr;
            if (op.sens_mode == 1) thr = op.sens0;
            else if (op.sens_mode == 2) thr = (op.sens0 - 20 * log10(op.sens1 / 1000.)) + 20 * log10((op.res_d * g) / 1000.);
            else thr = op.sens0 + 20 * log10((op.res_d * g) / 1000.);
            float zdb = 10.f * log10f(zh);
            cut = (double)zdb < thr;
        }
        if (act) {
            if (cut) {      // utilities.py:96-103: the SIMULATED_VARIABLES of the beam
                obs[CPOL_O_ZH * obs_plane + o] = qnan; obs[CPOL_O_ZV * obs_plane + o] = qnan;
                obs[CPOL_O_ZDR * obs_plane + o] = qnan; obs[CPOL_O_KDP * obs_plane + o] = qnan;
                obs[CPOL_O_RHOHV * obs_plane + o] = qnan; obs[CPOL_O_PHIDP * obs_plane + o] = qnan;
                if (rvel_out && op.doppler) rvel_out[o] = __builtin_nan("");
                if (spectrum)
                    for (int i = 0; i < n_vel; i++) spectrum[o * n_vel + i] = __builtin_nan("");
            } else {
                obs[CPOL_O_PHIDP * obs_plane + o] = phidp;
                if (att) {
                    obs[CPOL_O_ZH * obs_plane + o] = zh; obs[CPOL_O_ZV * obs_plane + o] = zv;
                    obs[CPOL_O_ZDR * obs_plane + o] = zdr;
                }
            }
        }
    }
}

void launch_observables(hipStream_t s, const ObsParams &op, int n_rays, long nu, const cpol_key_t *key,
                        const double *coef, const float *psd_out, const float *psd_hi, const signed char *mask,
                        const double *weight_per_sub, const double *vr, float *obs, long obs_plane,
                        long ray0, double *mask_out, double *rvel_out, const Dop2In &d2, double *spectrum, int n_vel)
{
    const long n = (long)n_rays * op.n_gates;
    const dim3 grid((unsigned)((n + GR_THREADS - 1) / GR_THREADS)), block(GR_THREADS);
#define CPOL_GR_LAUNCH(C, S) hipLaunchKernelGGL((k_gate_reduce<C, S>), grid, block, 0, s, op, n_rays, nu, key, coef, psd_out, psd_hi, mask, \
                                                weight_per_sub, vr, obs, obs_plane, ray0, mask_out, rvel_out, d2)
    const bool shared = op.doppler == 2 && d2.shared;
    if (op.comb) { if (shared) CPOL_GR_LAUNCH(true, true); else CPOL_GR_LAUNCH(true, false); }
    else { if (shared) CPOL_GR_LAUNCH(false, true); else CPOL_GR_LAUNCH(false, false); }
#undef CPOL_GR_LAUNCH
    hipLaunchKernelGGL(k_ray_scan, dim3(n_rays), dim3(64), 0, s, op, obs, obs_plane, ray0, rvel_out, spectrum,
                       n_vel);
}

// Diagnostic mode (interpolation.py:21-51, integrate_GH_pts): every model variable as the weighted sum over the
// sub-beams, formed like the reference's chain of np.nansum calls: float64, sub-beam after sub-beam, a NaN (the
// point lies outside the model: all variables of a sub-beam share the first one's mask, :272-278) counting as zero,
// each term value * weight / sum of the weights in that order.  One thread per (variable, ray, gate).
__global__ void k_diag(int n_rays, int n_sub, int n_gates, DiagVars dv, const float *__restrict__ vals, long plane,
                       const double *__restrict__ wsub, double sum_w, double *__restrict__ diag, long diag_plane, long ray0)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n = (long)n_rays * n_gates;
    if (t >= n) return;
    const int v = blockIdx.y;
    const int r = (int)(t / n_gates), g = (int)(t % n_gates);
    const float *src = vals + (long)dv.var[v] * plane;
    double acc = 0.;
    for (int b = 0; b < n_sub; b++) {
        const double y = (double)src[((long)r * n_sub + b) * n_gates + g] * wsub[b] / sum_w;
        acc += (y != y) ? 0. : y;
    }
    diag[(long)v * diag_plane + (ray0 + r) * (long)n_gates + g] = acc;
}

void launch_diag(hipStream_t s, int n_rays, int n_sub, int n_gates, const DiagVars &dv, const float *vals, long plane,
                 const double *wsub, double sum_w, double *diag, long diag_plane, long ray0)
{
    const long n = (long)n_rays * n_gates;
    hipLaunchKernelGGL(k_diag, dim3((unsigned)((n + 255) / 256), dv.n), dim3(256), 0, s, n_rays, n_sub, n_gates, dv, vals, plane,
                       wsub, sum_w, diag, diag_plane, ray0);
}

// geometry of the central sub-beam (scattering_sz.py:154-157)
__global__ void k_gather_central(int n_rays, int n_sub, int n_gates, const float *__restrict__ S,
                                 const float *__restrict__ H, const double *__restrict__ lat,
                                 const double *__restrict__ lon, float *dist, float *height, double *olat,
                                 double *olon, long ray0)
{
    long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)n_rays * n_gates) return;
    int r = (int)(t / n_gates), g = (int)(t % n_gates);
    long u = ((long)r * n_sub + n_sub / 2) * n_gates + g;
    long o = (ray0 + r) * (long)n_gates + g;
    if (dist) dist[o] = S[u];
    if (height) height[o] = H[u];
    if (olat) olat[o] = lat[u];
    if (olon) olon[o] = lon[u];
}

void launch_gather_central(hipStream_t s, int n_rays, int n_sub, int n_gates, const float *S, const float *H,
                           const double *lat, const double *lon, float *dist, float *height, double *olat,
                           double *olon, long ray0)
{
    long n = (long)n_rays * n_gates;
    hipLaunchKernelGGL(k_gather_central, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n_rays, n_sub, n_gates,
                       S, H, lat, lon, dist, height, olat, olon, ray0);
}

// doppler_c.c:11-33, one thread per velocity bin (three classes, stride 3)
__global__ void k_get_refl(float *refl, int nbins, const float *Da, const float *Db, const float *D, const float *rcs,
                           const float *N0, const float *mu, const float *lam, const float *nu, const float *step,
                           const float *dmin)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nbins) return;
    float out = 0.f;
    for (int j = 0; j < 3; j++) {
        if (lam[j] > 0) {
            float sum = 0;
            int ia = (int)((Da[i * 3 + j] - dmin[j]) / step[j]);
            int ib = (int)((Db[i * 3 + j] - dmin[j]) / step[j]);
            for (int k = ia; k < ib; k++) {
                double d = D[k * 3 + j];
                sum = (float)((double)sum + pow(d, (double)mu[j]) * exp(-(double)lam[j] * pow(d, (double)nu[j])) * (double)rcs[k * 3 + j]);
            }
            out += sum * step[j] * N0[j];
        }
    }
    refl[i] = out;
}

void launch_get_refl(hipStream_t s, float *refl, int nbins, const float *Da, const float *Db, const float *D,
                     const float *rcs, const float *N0, const float *mu, const float *lam, const float *nu,
                     const float *step, const float *dmin)
{
    hipLaunchKernelGGL(k_get_refl, dim3((nbins + 63) / 64), dim3(64), 0, s, refl, nbins, Da, Db, D, rcs, N0, mu, lam,
                       nu, step, dmin);
}

// Radar cross-section of every table entry, 2 pi (z0 - z1 - z2 + z3) in float32 (doppler_sz.py:255-258).
__global__ void k_build_rcs(const float *__restrict__ raw, float *__restrict__ rcs, long n)
{
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const float4 z = *(const float4 *)(raw + t * CPOL_N_SZ);
    rcs[t] = (float)(2 * M_PI) * (((z.x - z.y) - z.z) + z.w);
}

void launch_build_rcs(hipStream_t s, const float *raw, float *rcs, long n)
{
    hipLaunchKernelGGL(k_build_rcs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, raw, rcs, n);
}

// =============================================================================
// Doppler scheme 3: full Doppler spectrum (doppler_sz.py:233-306,337-367 and
// doppler_c.c:11-33).  One wavefront per (ray, gate); it walks the sub-beams, and for
// each: (1) the lanes evaluate the 3 x 1024 terms D^mu exp(-lambda D^nu) rcs in float64
// into LDS, (2) one lane per velocity bin converts the bin's fall-speed interval into a
// diameter interval per class and sums its terms with the reference's float accumulator,
// (3) the GH-weighted bin values are accumulated in float64.  Three classes (rain, snow,
// graupel), as the reference's get_refl hard-wires.
// =============================================================================
#ifndef SPEC_LONG_BIN
#define SPEC_LONG_BIN 256       // bins from this length on have their chains cut where the terms stop mattering
#endif
#define SPEC_EDGE_TILE 512       // bins per pass of the edge table: 513 edges x 3 floats fit the 8 KB term area
__global__ __launch_bounds__(64, 3)
void k_spectrum(ClassTable ct, SpecParams sp, long /*n_positions*/, const float *__restrict__ vals, long plane,
                const float *__restrict__ elev, const signed char *__restrict__ mask, const double *__restrict__ wsub,
                const double *__restrict__ phi_sc, const double *__restrict__ varray,
                const float *const *__restrict__ rcs_tab, const float *const *__restrict__ d_axis,
                double *__restrict__ spectrum, double *__restrict__ rvel, long ray0)
{
    extern __shared__ double sm[];
    // One class at a time goes through LDS (8 KB of terms instead of 24): the kernel is a chain of dependent
    // float additions per velocity bin, so what counts is how many wavefronts a CU can hold to overlap them.
    double *term = sm;                       // [1024] terms of the class in hand; with turbulence [3072]: weights and convolved bins
    double *acc = sm + (sp.turb ? 3 * 1024 : 1024);      // [n_vel]
    double *bs = acc + sp.n_vel;             // [n_vel] one sub-beam's weighted spectrum (turbulence broadening only)
    // per velocity bin: diameter index range [ia, ib) per class, float accumulator of the bin's reflectivity
    struct BinRec { short ia[3], ib[3]; float refl; };
    struct SpecSub { double hor, sth, Ww, sig; float lam32[3], n032[3], rinv; int node[3]; int active; };
    BinRec *rec = (BinRec *)(acc + sp.n_vel * (sp.turb ? 2 : 1));
    const int lane = threadIdx.x;
    const int G = sp.n_gates, B = sp.n_sub, NV = sp.n_vel;
    const int r = blockIdx.x / G, g = blockIdx.x % G;
    for (int i = lane; i < NV; i += 64) acc[i] = 0.;
    // What a sub-beam needs before its bins: PSD parameters, table nodes, fall-speed conversion, turbulence width.
    // One lane per sub-beam works these out side by side; the loop over the sub-beams then reads them from LDS.
    SpecSub *subs = (SpecSub *)(rec + NV);
    float *suf = (float *)(subs + B);        // [129] largest |term| from each block of eight diameters on, 0 past the axis
    for (int b = lane; b < B; b += 64) {
        const long rb = (long)r * B + b;
        const long u = rb * G + g;
        SpecSub ss;
        ss.active = mask[u] == 0 || mask[u] == CPOL_MASK_COMB;      // doppler_sz.py:271 (both codes: the point lies inside the model)
        const float T = vals[(long)CPOL_V_T * plane + u];
        const float e_raw = elev[u];
        const float e_lut = fold_elev(e_raw);
        for (int jc = 0; jc < 3; jc++) {
            const PsdView &p = ct.dop[jc];
            float q = vals[(long)p.q_var * plane + u];
            float Tp = vals[(long)p.t_var * plane + u];
            double l, n0;
            psd_params(p, q, 0.f, Tp, &l, &n0);
            if (q == 0.f) l = __builtin_nan("");
            ss.lam32[jc] = (float)l; ss.n032[jc] = (float)n0;
            // table node nearest to (elevation, temperature)
            const LutView &lv = ct.lut[jc];
            ss.node[jc] = nearest_node(e_lut, lv.e_min, lv.e_step, lv.n_e) * lv.n_t + nearest_node(T, lv.t_min, lv.t_step, lv.n_t);
        }
        // fall speed at the velocity bins (doppler_sz.py:337-346), NumPy promotion kept
        const float rho = vals[(long)CPOL_V_RHO * plane + u], rho0 = vals[(long)CPOL_V_RHO * plane + rb * G];
        ss.rinv = 1.0f / sqrtf(rho / rho0);
        const float th = e_raw * (float)(M_PI / 180.0);
        const double Uw = vals[(long)CPOL_V_U * plane + u], Vw = vals[(long)CPOL_V_V * plane + u];
        ss.Ww = vals[(long)CPOL_V_W * plane + u];
        // float32 sin/tan as the correctly rounded value: v/sin(theta) cancels against the wind term to a few
        // m/s out of hundreds at low elevation, so the last bit of these two decides which table term falls in
        // which velocity bin; the reference's result there depends on its libm in the same way
        ss.hor = (Uw * phi_sc[2 * rb] + Vw * phi_sc[2 * rb + 1]) / (double)(float)tan((double)th);
        ss.sth = (double)(float)sin((double)th);
        ss.sig = 0.;
        if (sp.turb) {
            // Standard deviation of the turbulence spread after Doviak and Zrnic (get_turb_std, doppler_sz.py:308-321,
            // with true division), in velocity bins
            const double edr = (double)vals[(long)CPOL_V_EDR * plane + u];
            const double rng = sp.range_res * ((double)g + 0.5);
            const double rt = rng * sp.sigma_theta;
            double tstd;
            if (sp.sigma_r < 0.1 * rt) tstd = cbrt((rng * edr * sp.sigma_theta * sp.kolm_a * sqrt(sp.kolm_a)) / 0.72);
            else {
                const double a135 = 1.35 * sp.kolm_a, ratio = rt / sp.sigma_r;
                const double den = 11. / 15. + 4. / 15. * ratio * ratio;
                tstd = cbrt((edr * sp.sigma_r * a135 * sqrt(a135)) * (den * sqrt(den)));      // x / den^(-3/2)
            }
            ss.sig = tstd / (varray[2] - varray[1]);
        }
        subs[b] = ss;
    }
    __syncthreads();
    for (int b = 0; b < B; b++) {
        const SpecSub &ss = subs[b];
        if (!ss.active) continue;                         // wave-uniform
        float lam32[3], n032[3];
        for (int jc = 0; jc < 3; jc++) { lam32[jc] = ss.lam32[jc]; n032[jc] = ss.n032[jc]; }
        const float rinv = ss.rinv;
        const double Ww = ss.Ww, hor = ss.hor, sth = ss.sth;
        const double w = wsub[b];
        // Diameter interval of every velocity bin and class.  An edge of the velocity axis belongs to two bins: its
        // three diameters (v / alpha)^(1 / beta), clipped, are worked out once per edge into the term area (free until
        // the first class's terms), SPEC_EDGE_TILE bins at a time, and the bins read their two edges from there.
        float *edge = (float *)term;                          // [tile + 1][3], NaN: negative fall speed at this edge
        bool has_long[3] = {false, false, false};             // a bin of the class takes SPEC_LONG_BIN diameters or more
        for (int i0 = 0; i0 < NV; i0 += SPEC_EDGE_TILE) {
            const int n_e = min(SPEC_EDGE_TILE + 1, NV - i0);             // edges i0 .. i0 + n_e - 1
            __syncthreads();
            for (int e = lane; e < n_e; e += 64) {
                const double wh = (double)rinv * (Ww + hor - varray[i0 + e] / sth);
                float d[3];
                for (int jc = 0; jc < 3; jc++) {
                    // (v / alpha)^(1 / beta): an exact square or fourth power where 1 / beta is 2 or 4, else through
                    // exp(y log x) - the float32 rounding hides its 1e-15
                    const double x = sp.inv_alpha[jc] * wh;
                    float dd = sp.inv_beta[jc] == 2. ? (float)(x * x)
                             : sp.inv_beta[jc] == 4. ? (float)((x * x) * (x * x)) : (float)powxy(x, sp.inv_beta[jc]);
                    // the reference clips the whole matrix with each later class's limits (:355-357)
                    for (int jl = jc; jl < 3; jl++) {
                        if (dd >= sp.d_max[jl]) dd = sp.d_max[jl];
                        if (dd <= sp.d_min[jl]) dd = sp.d_min[jl];
                    }
                    d[jc] = wh >= 0 ? dd : __builtin_nanf("");
                }
                edge[3 * e] = d[0]; edge[3 * e + 1] = d[1]; edge[3 * e + 2] = d[2];
            }
            __syncthreads();
            for (int i = i0 + lane; i < min(i0 + SPEC_EDGE_TILE, NV); i += 64) {
                BinRec br;
                for (int jc = 0; jc < 3; jc++) { br.ia[jc] = -1; br.ib[jc] = -1; }      // -1: the class adds nothing to this bin
                br.refl = 0.f;
                if (i < NV - 1) {
                    const float *e0 = edge + 3 * (i - i0), *e1 = e0 + 3;
                    if (e0[0] == e0[0] && e1[0] == e1[0]) {         // both fall speeds are >= 0
                        float Da[3], Db[3];
                        bool keep = false;
                        for (int jc = 0; jc < 3; jc++) {
                            Da[jc] = fminf(e0[jc], e1[jc]); Db[jc] = fmaxf(e0[jc], e1[jc]);
                            keep = keep || (Db[jc] - Da[jc]) != 0.0f;
                        }
                        if (keep) {
                            for (int jc = 0; jc < 3; jc++) {
                                if (lam32[jc] > 0.f) {
                                    int ia = (int)((Da[jc] - sp.d_min[jc]) / sp.step[jc]);
                                    int ib = (int)((Db[jc] - sp.d_min[jc]) / sp.step[jc]);
                                    br.ia[jc] = (short)min(max(ia, 0), 1024);
                                    br.ib[jc] = (short)min(max(ib, 0), 1024);
                                    has_long[jc] = has_long[jc] || (sp.chain_cut && br.ib[jc] - br.ia[jc] >= SPEC_LONG_BIN);
                                }
                            }
                        }
                    }
                }
                rec[i] = br;
            }
        }
        for (int jc = 0; jc < 3; jc++) {
            if (!(lam32[jc] > 0.f)) continue;                 // wave-uniform: the class adds nothing (doppler_c.c:20)
            // terms of the class at the sub-beam's table node
            const float *sl = rcs_tab[jc] + (long)ss.node[jc] * 1024;
            const double lamd = (double)lam32[jc];
            const float muj = sp.mu[jc];
            __syncthreads();                                  // the previous class's sums have read term[]
            if (sp.geo_step[jc] > 0.) {
                // nu = 1 on an equally spaced axis: a lane's diameters are 64 steps apart, so its exponentials
                // form a geometric sequence, E(D_0 + 64 i step) = E(D_0) R^i: two exponentials per lane and class
                // instead of sixteen.  The axis values are float32 roundings of the ideal ones; the difference
                // delta enters to first order, exp(-lambda (D + delta)) = exp(-lambda D) (1 - lambda delta), the
                // second order being below 1e-10 for |lambda delta| < 1e-5.
                const float *dx = d_axis[jc];
                const double D0 = (double)dx[0] + (double)lane * sp.geo_step[jc];
                double E = exp(-lamd * D0);
                const double R = exp(-lamd * 64. * sp.geo_step[jc]);
                // eight diameters' table values are fetched before the eight terms that use them
                for (int i0 = 0; i0 < 16; i0 += 8) {
                    float rc[8], da[8];
#pragma unroll
                    for (int j = 0; j < 8; j++) { rc[j] = sl[lane + 64 * (i0 + j)]; da[j] = dx[lane + 64 * (i0 + j)]; }
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int i = i0 + j, k = lane + 64 * i;
                        const double D = (double)da[j];
                        const double delta = D - (D0 + (double)(64 * i) * sp.geo_step[jc]);
                        const double dm = muj == 0.f ? 1. : (muj == 1.f ? D : (muj == 0.5f ? sqrt(D) : pow(D, (double)muj)));
                        term[k] = dm * (E - E * (lamd * delta)) * (double)rc[j];
                        E *= R;
                    }
                }
            } else {
                for (int k = lane; k < 1024; k += 64) {
                    const float rcs = sl[k];
                    double D = (double)d_axis[jc][k];
                    double dm = muj == 0.f ? 1. : (muj == 1.f ? D : pow(D, (double)muj));
                    double dn = sp.nu[jc] == 1.f ? D : pow(D, (double)sp.nu[jc]);
                    term[k] = dm * exp(-lamd * dn) * (double)rcs;
                }
            }
            __syncthreads();
            // Where the chain may stop: a term below 2^-26 of the float sum leaves it as it is (the sum's half step is at
            // least that, on either side, and a tie keeps the even sum), so once every term still to come is below
            // 2^-27 of the sum the rest of the bin changes nothing.  suf[m] = largest |term| of the diameters 8 m ..
            // 1023, rounded up; a NaN term counts as infinite (it must reach the sum).  Worked out only when a bin of the
            // class is long enough for it to pay.
            const bool long_bin = __any(has_long[jc]);
            if (long_bin) {
                float v[2];
#pragma unroll
                for (int h2 = 0; h2 < 2; h2++) {
                    double a = 0.;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const double t = term[8 * (lane + 64 * h2) + j];
                        a = t != t ? (double)__builtin_inff() : fmax(a, fabs(t));
                    }
                    v[h2] = (float)a * 1.000001f;
                }
                // suffix maxima over the 64 lanes of each half, then the upper half's total into the lower
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const float o0 = __shfl_down(v[0], off), o1 = __shfl_down(v[1], off);
                    if (lane + off < 64) { v[0] = fmaxf(v[0], o0); v[1] = fmaxf(v[1], o1); }
                }
                v[0] = fmaxf(v[0], __shfl(v[1], 0));
                suf[lane] = v[0]; suf[lane + 64] = v[1];
                if (lane == 0) suf[128] = 0.f;
                __syncthreads();
            }
            // the reference's float accumulator per bin (doppler_c.c:24-26), classes in order
            // Eight terms are fetched ahead of the eight dependent additions that use them: the chain then runs at
            // the latency of its own convert-add-convert, not at that of an LDS read per term.
            for (int i = lane; i < NV - 1; i += 64) {
                const int ia = rec[i].ia[jc], ib = rec[i].ib[jc];
                if (ia >= 0) {
                    float sum = 0.f;
                    for (int k = ia; k < ib; k += 8) {
                        if (long_bin && suf[k >> 3] < fabsf(sum) * 0x1p-27f) break;      // nothing from here on changes the sum
                        double t[8];
#pragma unroll
                        for (int j = 0; j < 8; j++) t[j] = term[min(k + j, 1023)];
#pragma unroll
                        for (int j = 0; j < 8; j++) {
                            const float s2 = (float)((double)sum + t[j]);
                            sum = k + j < ib ? s2 : sum;
                        }
                    }
                    rec[i].refl += sum * sp.step[jc] * n032[jc];
                }
            }
        }
        __syncthreads();
        if (sp.turb) {
            for (int i = lane; i < NV; i += 64) bs[i] = (double)(rec[i].refl * sp.zfac) * w;
        } else {
            for (int i = lane; i < NV; i += 64) acc[i] += (double)(rec[i].refl * sp.zfac) * w;
        }
        __syncthreads();
        if (sp.turb) {
            // Spectrum spread by turbulence (doppler_sz.py:94-96): gaussian_filter1d of the bins (reflecting edges,
            // radius int(4 sigma + 0.5), normalised weights) and rescaling to the power the sub-beam's spectrum
            // had before (turb_spectrum_spread, :323-335).
            const double sig = ss.sig;
            if (!(sig > 1e-15)) {                 // no spread (scipy skips such an axis); NaN falls here as well
                for (int i = lane; i < NV; i += 64) acc[i] += bs[i];
            } else {
                int R = (int)fmin(4.0 * sig + 0.5, 1023.0);                // weights live in term[0 .. R]
                const double c = -0.5 / (sig * sig);
                double ws = 0., p0 = 0.;
                for (int k = lane; k <= R; k += 64) {
                    const double wk = exp(c * (double)k * (double)k);
                    term[k] = wk;
                    ws += k == 0 ? wk : 2. * wk;
                }
                for (int i = lane; i < NV; i += 64) p0 += bs[i];
                for (int off = 32; off > 0; off >>= 1) { ws += __shfl_xor(ws, off); p0 += __shfl_xor(p0, off); }
                __syncthreads();
                double *cv = term + 1024;                                   // [n_vel <= 2048] convolved bins
                double p1 = 0.;
                for (int i = lane; i < NV; i += 64) {
                    double a = 0.;
                    for (int k = -R; k <= R; k++) {
                        int idx = i + k;
                        // half-sample symmetric extension: d c b a | a b c d | d c b a
                        while (idx < 0 || idx >= NV) idx = idx < 0 ? -1 - idx : 2 * NV - 1 - idx;
                        a += term[k < 0 ? -k : k] * bs[idx];
                    }
                    a /= ws;
                    cv[i] = a;
                    p1 += a;
                }
                for (int off = 32; off > 0; off >>= 1) p1 += __shfl_xor(p1, off);
                // an all-zero spectrum stays all zero (the reference's 0 / 0 would turn the gate into NaN)
                if (p1 != 0.)
                    for (int i = lane; i < NV; i += 64) acc[i] += cv[i] / p1 * p0;
            }
            __syncthreads();
        }
    }
    // spectrum and its first moment (doppler_sz.py:99)
    double s0 = 0., s1 = 0.;
    const long o = ((ray0 + r) * (long)G + g);
    for (int i = lane; i < NV; i += 64) {
        double a = acc[i];
        if (spectrum) spectrum[o * NV + i] = a;
        s0 += a; s1 += varray[i] * a;
    }
    for (int off = 32; off > 0; off >>= 1) { s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); }
    if (lane == 0 && rvel) rvel[o] = s1 / s0;
}

void launch_spectrum(hipStream_t s, const ClassTable &ct, const SpecParams &sp, int n_rays, long nu, const float *vals,
                     long plane, const float *elev, const signed char *mask, const double *wsub, const double *phi_sc,
                     const double *varray, const float *const *raw, const float *const *d_axis, double *spectrum,
                     double *rvel, long ray0)
{
    size_t lds = (size_t)((sp.turb ? 3 * 1024 : 1024) + sp.n_vel * (sp.turb ? 2 : 1)) * sizeof(double) + (size_t)sp.n_vel * 16
               + (size_t)sp.n_sub * 80 + 129 * sizeof(float);      // terms, accumulators, BinRec per bin, SpecSub per sub-beam, suffix maxima
    hipLaunchKernelGGL(k_spectrum, dim3((unsigned)((long)n_rays * sp.n_gates)), dim3(64), lds, s, ct, sp, nu, vals, plane,
                       elev, mask, wsub, phi_sc, varray, raw, d_axis, spectrum, rvel, ray0);
}
