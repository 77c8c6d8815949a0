/* TEST INFRASTRUCTURE ONLY -- plain-C restatement of the native pieces of the
 * reference hot path.  Linked by nothing in the product; loaded through ctypes by
 * oracle/cosmo_oracle.py, which only tests/, smoke() and bench.py's cpu_baseline
 * leg may import.
 *
 *  orc_interp     restates interpolation/interpolation_c.c:12-164 (get_all_radar_pts,
 *                 binary_search, trilinear_interp) gate by gate.  In the reference a
 *                 below-terrain / above-domain gate fills out[i..end] and the loop
 *                 goes on, so each gate's final value depends on that gate alone.
 *  orc_get_refl   restates doppler/doppler_c.c:11-33.
 *  orc_geod_fwd   WGS84 direct geodesic = what pyproj's Geod(ellps='WGS84').fwd
 *                 computes at interpolation/interpolation.py:322-325.  pyproj is a
 *                 third-party dependency absent from the reference tree (version
 *                 unpinned there); the published algorithm (C. F. F. Karney,
 *                 "Algorithms for geodesics", J. Geodesy 87, 2013, 6th-order series)
 *                 is restated and pinned by the 38x300 lat/lon known answers of
 *                 interpolation/ex_bemas.txt (tests/golden/ex_bemas.npz).
 *  orc_rotpole    rotated-pole transform standing in for pycosmo.WGS_to_COSMO
 *                 (interpolation/interpolation.py:355).  pycosmo is absent from the
 *                 reference snapshot: PARITY UNPINNED; standard COSMO formula.
 *
 * Build: gcc -O2 -ffp-contract=off (see Makefile) -- no FMA, no fast-math.
 */
#include <math.h>
#include <string.h>

/* ------------------------------------------------------------------ interp */
static int orc_level_search(const float *col, int stride, int dim, float key)
{   /* interpolation_c.c:108-135, heights descending with index */
    int high = 0, low = dim - 1, middle = 0;
    if (key > col[0]) return -1;
    else if (key < col[(long)(dim - 1) * stride]) return -2;
    while (high < low) {
        middle = (high + low) / 2;
        float v = col[(long)middle * stride];
        if (v == key) return middle;
        else if (low == middle || high == middle) return high;
        else if (v < key) low = middle;
        else if (v > key) high = middle;
    }
    return high;
}

void orc_interp(float *out, int n, const float *coords, const float *H,
                const float *data, const float *zlev, int nz, int ny, int nx,
                const float *llc, const float *res)
{
    long plane = (long)ny * nx;
    for (int i = 0; i < n; i++) {
        /* interpolation_c.c:43-57: position in grid units; [0] = lat-like (y), [1] = lon-like (x) */
        float p0 = (coords[2 * i + 0] - llc[1]) / res[1];
        float p1 = (coords[2 * i + 1] - llc[0]) / res[0];
        int iy = (int)floor(p0), ix = (int)floor(p1);
        float x = fmod(p0, 1.0), y = fmod(p1, 1.0);
        float dx = 1.0 - x, dy = 1.0 - y;
        long nb[4] = {(long)iy * nx + ix, (long)iy * nx + ix + 1,
                      (long)(iy + 1) * nx + ix, (long)(iy + 1) * nx + ix + 1};
        float t[4];
        for (int k = 0; k < 4; k++) t[k] = zlev[(long)(nz - 1) * plane + nb[k]];
        float topo = dx * dy * t[0] + x * t[2] * dy + dx * t[1] * y + x * y * t[3];   /* :61 */
        if (!(topo < H[i])) { out[i] = 0.0 / 0.0; continue; }                         /* :100-102 */
        float c[8], v[8];
        int above = 0;
        for (int j = 0; j < 4; j++) {
            int idx = orc_level_search(zlev + nb[j], (int)plane, nz, H[i]);
            int c1, c2;
            if (idx == -1) { above = 1; break; }                                      /* :70-73 */
            if (idx == -2) { c1 = nz - 3; c2 = nz - 2; }                              /* :74-77 */
            else { if (idx == nz - 2) idx--; c1 = idx; c2 = idx + 1; }                /* :79-81 */
            c[2 * j] = zlev[(long)c1 * plane + nb[j]];
            c[2 * j + 1] = zlev[(long)c2 * plane + nb[j]];
            v[2 * j] = data[(long)c1 * plane + nb[j]];
            v[2 * j + 1] = data[(long)c2 * plane + nb[j]];
        }
        if (above) { out[i] = -9999; continue; }                                      /* :92-94 */
        float w[4];
        for (int j = 0; j < 4; j++)                                                   /* :151 */
            w[j] = v[2 * j + 1] - (v[2 * j + 1] - v[2 * j]) / (c[2 * j] - c[2 * j + 1]) * (H[i] - c[2 * j + 1]);
        out[i] = dx * dy * w[0] + x * w[2] * dy + dx * w[1] * y + x * y * w[3];       /* :162 */
    }
}

/* ------------------------------------------------------------------ doppler */
void orc_get_refl(float *refl, int len, const float *Da, int nbins, const float *Db,
                  const float *D, int nD, const float *rcs, const float *N0, const float *mu,
                  const float *lambda, const float *nu, const float *step_D, const float *Dmin)
{   /* doppler_c.c:11-33; three classes hard-wired (stride 3) as in the reference */
    (void)nD;
    memset(refl, 0, len * sizeof(float));
    for (int i = 0; i < nbins; i++)
        for (int j = 0; j < 3; j++)
            if (lambda[j] > 0) {
                float sum = 0;
                int ia = (int)((Da[i * 3 + j] - Dmin[j]) / step_D[j]);
                int ib = (int)((Db[i * 3 + j] - Dmin[j]) / step_D[j]);
                for (int k = ia; k < ib; k++)
                    sum += pow(D[k * 3 + j], mu[j]) * exp(-lambda[j] * pow(D[k * 3 + j], nu[j])) * rcs[k * 3 + j];
                refl[i] += sum * step_D[j] * N0[j];
            }
}

/* ------------------------------------------------------------------ geodesic */
#define GEOD_A 6378137.0
#define GEOD_F (1.0 / 298.257223563)
#define DEG (M_PI / 180.0)

static double polyv(int n, const double *p, double x)
{
    double y = *p++;
    while (--n >= 0) y = y * x + *p++;
    return y;
}

static void norm2(double *s, double *c) { double r = hypot(*s, *c); *s /= r; *c /= r; }

static double ang_round(double x)
{
    const double z = 1 / 16.0;
    volatile double y;
    if (x == 0) return 0;
    y = fabs(x);
    y = y < z ? z - (z - y) : y;
    return x < 0 ? -y : y;
}

static double ang_normalize(double x)
{
    x = remainder(x, 360.0);
    return x != -180 ? x : 180;
}

static void sincosd(double x, double *sx, double *cx)
{   /* exact reduction to [-45,45] then quadrant fix-up */
    double r, s, c; int q;
    r = remquo(x, 90.0, &q);
    r *= DEG;
    s = sin(r); c = cos(r);
    switch ((unsigned)q & 3U) {
    case 0U: *sx = s;  *cx = c;  break;
    case 1U: *sx = c;  *cx = -s; break;
    case 2U: *sx = -s; *cx = -c; break;
    default: *sx = -c; *cx = s;  break;
    }
    if (x != 0) { *sx += 0.0; *cx += 0.0; }
}

static double atan2d(double y, double x)
{
    int q = 0; double ang;
    if (fabs(y) > fabs(x)) { double t = x; x = y; y = t; q = 2; }
    if (x < 0) { x = -x; ++q; }
    ang = atan2(y, x) / DEG;
    switch (q) {
    case 1: ang = (y >= 0 ? 180 : -180) - ang; break;
    case 2: ang = 90 - ang; break;
    case 3: ang = -90 + ang; break;
    }
    return ang;
}

static double sin_cos_series(double sinx, double cosx, const double *c, int n)
{   /* Clenshaw summation of sum_{k=1..n} c[k] sin(2 k x) */
    double ar, y0, y1;
    c += (n + 1);
    ar = 2 * (cosx - sinx) * (cosx + sinx);
    y0 = (n & 1) ? *--c : 0; y1 = 0;
    n /= 2;
    while (n--) {
        y1 = ar * y0 - y1 + *--c;
        y0 = ar * y1 - y0 + *--c;
    }
    return 2 * sinx * cosx * y0;
}

static double A1m1f(double eps)
{
    static const double co[] = {1, 4, 64, 0, 256};
    double t = polyv(3, co, eps * eps) / co[4];
    return (t + eps) / (1 - eps);
}

static void C1f(double eps, double *c)
{
    static const double co[] = {
        -1, 6, -16, 32,
        -9, 64, -128, 2048,
        9, -16, 768,
        3, -5, 512,
        -7, 1280,
        -7, 2048};
    double eps2 = eps * eps, d = eps; int o = 0, l;
    for (l = 1; l <= 6; ++l) {
        int m = (6 - l) / 2;
        c[l] = d * polyv(m, co + o, eps2) / co[o + m + 1];
        o += m + 2; d *= eps;
    }
}

static void C1pf(double eps, double *c)
{
    static const double co[] = {
        205, -432, 768, 1536,
        4005, -4736, 3840, 12288,
        -225, 116, 384,
        -7173, 2695, 7680,
        3467, 7680,
        38081, 61440};
    double eps2 = eps * eps, d = eps; int o = 0, l;
    for (l = 1; l <= 6; ++l) {
        int m = (6 - l) / 2;
        c[l] = d * polyv(m, co + o, eps2) / co[o + m + 1];
        o += m + 2; d *= eps;
    }
}

static void A3coeff(double n, double *A3x)
{
    static const double co[] = {
        -3, 128,
        -2, -3, 64,
        -1, -3, -1, 16,
        3, -1, -2, 8,
        1, -1, 2,
        1, 1};
    int o = 0, k = 0, j;
    for (j = 5; j >= 0; --j) {
        int m = 5 - j < j ? 5 - j : j;
        A3x[k++] = polyv(m, co + o, n) / co[o + m + 1];
        o += m + 2;
    }
}

static void C3coeff(double n, double *C3x)
{
    static const double co[] = {
        3, 128,
        2, 5, 128,
        -1, 3, 3, 64,
        -1, 0, 1, 8,
        -1, 1, 4,
        5, 256,
        1, 3, 128,
        -3, -2, 3, 64,
        1, -3, 2, 32,
        7, 512,
        -10, 9, 384,
        5, -9, 5, 192,
        7, 512,
        -14, 7, 512,
        21, 2560};
    int o = 0, k = 0, l, j;
    for (l = 1; l < 6; ++l)
        for (j = 5; j >= l; --j) {
            int m = 5 - j < j ? 5 - j : j;
            C3x[k++] = polyv(m, co + o, n) / co[o + m + 1];
            o += m + 2;
        }
}

void orc_geod_fwd(double lon1, double lat1, double azi1, const double *s12, int npts,
                  double *lon2, double *lat2)
{
    const double f = GEOD_F, f1 = 1 - f, e2 = f * (2 - f), ep2 = e2 / (f1 * f1);
    const double n = f / (2 - f), b = GEOD_A * f1;
    const double tiny = sqrt(2.2250738585072014e-308);
    double A3x[6], C3x[15], C1a[7], C1pa[7], C3a[6];
    double salp1, calp1, sbet1, cbet1, salp0, calp0, ssig1, csig1, somg1, comg1;
    double k2, eps, A1m1, B11, stau1, ctau1, A3c, B31, s, c;
    int i, o, l;

    A3coeff(n, A3x);
    C3coeff(n, C3x);
    azi1 = ang_normalize(azi1);
    sincosd(ang_round(azi1), &salp1, &calp1);
    sincosd(ang_round(lat1), &sbet1, &cbet1);
    sbet1 *= f1;
    norm2(&sbet1, &cbet1);
    cbet1 = cbet1 > tiny ? cbet1 : tiny;
    salp0 = salp1 * cbet1;
    calp0 = hypot(calp1, salp1 * sbet1);
    ssig1 = sbet1; somg1 = salp0 * sbet1;
    csig1 = comg1 = (sbet1 != 0 || calp1 != 0) ? cbet1 * calp1 : 1;
    norm2(&ssig1, &csig1);
    k2 = calp0 * calp0 * ep2;
    eps = k2 / (2 * (1 + sqrt(1 + k2)) + k2);
    A1m1 = A1m1f(eps);
    C1f(eps, C1a);
    B11 = sin_cos_series(ssig1, csig1, C1a, 6);
    s = sin(B11); c = cos(B11);
    stau1 = ssig1 * c + csig1 * s;
    ctau1 = csig1 * c - ssig1 * s;
    C1pf(eps, C1pa);
    A3c = -f * salp0 * polyv(5, A3x, eps);
    {   /* C3f */
        double mult = 1; o = 0;
        for (l = 1; l < 6; ++l) {
            int m = 6 - l - 1;
            mult *= eps;
            C3a[l] = mult * polyv(m, C3x + o, eps);
            o += m + 1;
        }
    }
    B31 = sin_cos_series(ssig1, csig1, C3a, 5);

    for (i = 0; i < npts; i++) {
        double tau12 = s12[i] / (b * (1 + A1m1));
        double B12, sig12, ssig12, csig12, ssig2, csig2, sbet2, cbet2, somg2, comg2, omg12, lam12, lon12;
        s = sin(tau12); c = cos(tau12);
        B12 = -sin_cos_series(stau1 * c + ctau1 * s, ctau1 * c - stau1 * s, C1pa, 6);
        sig12 = tau12 - (B12 - B11);
        ssig12 = sin(sig12); csig12 = cos(sig12);
        ssig2 = ssig1 * csig12 + csig1 * ssig12;
        csig2 = csig1 * csig12 - ssig1 * ssig12;
        sbet2 = calp0 * ssig2;
        cbet2 = hypot(salp0, calp0 * csig2);
        if (cbet2 == 0) cbet2 = csig2 = tiny;
        somg2 = salp0 * ssig2; comg2 = csig2;
        omg12 = atan2(somg2 * comg1 - comg2 * somg1, comg2 * comg1 + somg2 * somg1);
        lam12 = omg12 + A3c * (sig12 + (sin_cos_series(ssig2, csig2, C3a, 5) - B31));
        lon12 = lam12 / DEG;
        lon2[i] = ang_normalize(ang_normalize(lon1) + ang_normalize(lon12));
        lat2[i] = atan2d(sbet2, f1 * cbet2);
    }
}

/* ------------------------------------------------------------------ rotated pole */
void orc_rotpole(const double *lat, const double *lon, int n, double pole_lat_s, double pole_lon_s,
                 double *rlat, double *rlon)
{   /* north pole of the rotated grid from the southern-pole metadata */
    double pollat = -pole_lat_s, pollon = pole_lon_s - 180.0;
    double sp, cp;
    if (pollon < -180.0) pollon += 360.0;
    sp = sin(pollat * DEG); cp = cos(pollat * DEG);
    for (int i = 0; i < n; i++) {
        double phi = lat[i] * DEG, dl = (lon[i] - pollon) * DEG;
        double sphi = sin(phi), cphi = cos(phi), sdl = sin(dl), cdl = cos(dl);
        double a = cp * cphi * cdl + sp * sphi;
        double a1 = -sdl * cphi;
        double a2 = -sp * cphi * cdl + cp * sphi;
        rlat[i] = asin(a) / DEG;
        rlon[i] = atan2(a1, a2) / DEG;
    }
}
