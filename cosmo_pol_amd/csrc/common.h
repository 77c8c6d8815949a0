// Shared declarations of the HIP implementation (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/cosmopol.h"

#define CPOL_WAVE 64
#define CPOL_CHUNK_UNITS 1024      // units (gate x sub-beam x class) per integrate workgroup
typedef short cpol_key_t;         // bucket (class, elevation node, temperature node) of a unit, -1 = not valid; at most 16384 buckets
#define CPOL_SCAN_RUNS_MAX 1024    // runs of 256 buckets in the two-level bucket scan: at most 262144 buckets
#define CPOL_SORT_SHARDS 64         // counters per bucket: spreads same-address atomics of big buckets
#define CPOL_DOP_SLICE 3072       // floats per node of the compact Doppler image: 256 x 3 words of 4
#define CPOL_LINE_N 40            // doubles per (ray, sub-beam) record of the geodesic line constants
// work counters of the two integration kernels: CPOL_WORK_SLOTS each, one per 64-byte line, so that the wavefronts
// of a launch spread their atomics over many lines
#define CPOL_WORK_SLOTS 1024
#define CPOL_WORK_STRIDE 8
#define CPOL_N_COL 16             // floats per result row of a unit integrated over all columns: 12 scattering components (+ 2 Doppler moments) + spare
#define CPOL_MASK_COMB 4          // interpolation code (-1, 0, 1) of a point inside the model one of whose units takes the combined columns
#define CPOL_N_COMB 8             // columns of the combined image = floats per result row of a unit integrated over it: 7 combinations + a spare

struct LutDev {
    bool   present = false;
    int    n_e = 0, n_t = 0, n_d = 0;
    float  e_min = 0, e_step = 1, t_min = 0, t_step = 1;
    float  dD = 0;                 // D[1]-D[0] in float32 (scattering_sz.py:50)
    float *raw = nullptr;          // [n_e][n_t][1024][12] as uploaded
    float *rcs = nullptr;          // [n_e][n_t][1024] radar cross-section 2 pi (z0 - z1 - z2 + z3), built for the Doppler spectrum
    float *table = nullptr;        // [n_e][n_t] slices of 1024 x ncol, permuted and pre-multiplied by D^mu (see k_integrate)
    float *table_comb = nullptr;   // the same for the eight combined columns (see k_build_image), built when the configuration uses them
    float *node_floor = nullptr;   // [n_e][n_t]: smallest first-diameter term of the node's columns, 0 where a column changes sign (k_node_floor)
    long   nodes_comb = 0;         // nodes with a positive floor
    float *xq = nullptr;           // D^nu, permuted for the integrate kernel's ds_read_b128
    float *table_dop = nullptr;    // compact 3-column image of the separate Doppler units (k_integrate_dop)
    float *xq_dop = nullptr;
    float *d_axis = nullptr;       // [n_d]
    bool   axis_uniform = false;   // diameters equally spaced to float32 rounding
    double axis_step = 0;
    float  d0 = 0;                 // first diameter
    std::vector<float> d_host;     // the diameter axis as uploaded
    int    image_ncol = 0;         // what `table` was last built for
    float  cut_e = 152.f;          // term cutoff exponent the tables of this class are certified for (see k_cut_certificate)
    float  cut_e_dop = 152.f;      // the same for the separate Doppler units (their own PSD)
    cpol_psd psd, psd_dop;
};

// Kernel-side view of one class.
struct LutView {
    const float *table;
    const float *xq;
    const float *floor;            // per node, see LutDev::node_floor (views of the combined image only)
    int   n_e, n_t, n_d;
    float e_min, e_step, t_min, t_step;
    float dD;
};

struct PsdView {
    int    kind;
    double a, b, alpha, beta, mu, nu, lambda_factor, vel_factor, ntot_factor, n0_fixed, x_min, x_max;
    int    q_var, qn_var, t_var;
    double n0_unit, lam_unit;      // 1000^-(1+mu), 1000^-nu (hydrometeors.py:149,151)
    double log_lam_unit;
    // quotients of the constants above, worked out once on the host with the same float64 operations:
    double c_exp;                  // exponent of lambda: 1 / (4 + mu) | 1 / (b + 1) | -nu / b  (kind 0 | 1 | 2)
    double c_n, c_n_exp;           // kind 2: nu / ntot_factor, (mu + 1) / nu
    // base-2 forms for psd_fast: log2(lambda_factor); log2(log2 e) plus, for kind 2, log2(lam_unit); c_n * n0_unit
    double log2_lf, l2_const, cn_unit;
};

struct ClassTable {
    int     n_cls;
    int     ne_max, nt_max;        // bucket id = ((kind*n_cls + cls)*ne_max + ie)*nt_max + it
    int     n_kinds;               // 1, or 2 when Doppler scheme 2 needs units of its own (kind 1)
    int     ncol;                  // columns of the polarimetric image: 12, or 14 when the two Doppler scheme-2
                                   // moments ride on the polarimetric units (dop_shared)
    int     dop_shared;            // Doppler moments ride on the polarimetric units (identical PSD)
    // Two forms of polarimetric unit when comb = 1: planes [0, n_cls) are integrated over the seven combinations of
    // columns the observables are made of (lut_comb, 8 columns), planes [n_cls, 2 n_cls) over all twelve columns (lut).
    // A unit takes the first form when its contribution to every one of the twelve components is certain to be a
    // normal float32: every column of its node has one sign along the diameter axis and
    // coef * lut_comb[h].floor[node] >= 2^-100.
    int     comb;
    int     n_pol_planes;          // n_cls * (1 + comb)
    LutView lut_comb[CPOL_MAX_CLASSES];
    LutView lut[CPOL_MAX_CLASSES];
    LutView lut_dop[CPOL_MAX_CLASSES];
    double  pol_x0[CPOL_MAX_CLASSES];         // D_0^nu of the polarimetric PSD
    double  dop_x0[CPOL_MAX_CLASSES];         // D_0^nu of the Doppler PSD as float32 (separate Doppler units)
    float   dop_geo_step[CPOL_MAX_CLASSES];
    float   pol_geo_step[CPOL_MAX_CLASSES];   // same for the polarimetric PSD   // 4 x diameter step when the Doppler PSD has nu = 1 on a uniform axis, else 0
    PsdView psd[CPOL_MAX_CLASSES];
    PsdView dop[CPOL_MAX_CLASSES];
    int     cut_ok[CPOL_MAX_CLASSES];          // D^nu ascending along the diameter axis
    float   cut_e[CPOL_MAX_CLASSES];           // terms with N(D) < 2^-(cut_e - 2) N(D_0) are left out (152: exact float32 zeros only)
    int     dop_cut_ok[CPOL_MAX_CLASSES];      // same for the Doppler PSD's nu
    float   dop_cut_e[CPOL_MAX_CLASSES];       // cutoff exponent of the separate Doppler units
};

struct ModelView {
    const float *zcol;             // [ny][nx][nz]     heights, level index fastest
    const float *zc8;              // [ny][nx][8]      heights at levels (k (nz - 1)) / 7, k = 0..7
    const float *dcol;             // [ny][nx][nz][nvp] variables interleaved per level, nvp = nv rounded up to 4
    int   nz, ny, nx, nv, nvp;
    float llc0, llc1, res0, res1;  // (Lo1, La1), (dlon, dlat)
    float urc0, urc1;
    int   var_of_slot[CPOL_MAX_VARS];   // uploaded variable i -> cpol_var slot
    int   slot_of_var[CPOL_V_COUNT];    // the inverse, -1 for a variable that was not uploaded
    int   fast_div;                     // the model's values allow the shared-reciprocal division (k_model_range)
};

struct GeomParams {
    double lat0, lon0, alt;        // radar.coords
    double res;                    // radial resolution
    double earth_radius;           // get_earth_radius(coords[1]) (atm_refraction.py:114-117)
    double pole_lat_s, pole_lon_s;
    int    n_gates, n_sub;
    // refraction table (scheme 2): sorted total elevations and their (S, H, E) profiles; n_refr = 0 -> 4/3 model
    int    n_refr;
    const double *refr_keys;
    const float  *refr_S, *refr_H, *refr_E;
};

struct ObsParams {
    int    n_gates, n_sub, n_cls;
    float  res, res_km;            // radial resolution and resolution / 1000 as float32 operands
    float  two_pi, zfac, kdpfac, two_wl, afac;
    int    att_corr, doppler, sens_mode;
    int    row;                    // floats per result row of the PSD integration: 16, or 8 with combined columns
    int    comb, nb_form;          // combined columns in use; keys from nb_form on are units in the twelve-column form
    int    hi;                     // floats per unit in psd_hi
    const unsigned char *uflag;    // per point: bit h = class h holds a unit, bit 4 + h = that unit takes the combined columns
    double sens0, sens1, res_d;
};

struct cpol_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;         // device-to-host copies of cpol_run_rays, beside the kernels of the next batch
    std::vector<hipEvent_t> copy_events;       // two per batch: its front kernel / all its kernels have finished
    std::string err;
    bool have_cfg = false, have_model = false;
    cpol_config cfg;
    // model
    float *zcol = nullptr, *dcol = nullptr, *zc8 = nullptr;
    int nz = 0, ny = 0, nx = 0, nv = 0;
    int var_ids[CPOL_MAX_VARS];
    bool fast_div = false;             // see ModelView::fast_div
    float llc[2], urc[2], res[2];
    double pole[2];
    LutDev lut[CPOL_MAX_CLASSES];
    // quadrature
    int n_sub = 0;
    std::vector<double> az_off, el_off, weight;
    double *quad_dev = nullptr;    // [3][n_sub]: az_off, el_off, weight
    int n_vel = 0;                 // Doppler spectrum bins (constants.py:75 VARRAY)
    int turb = 0;                  // turbulence broadening of the spectrum
    double turb_beamwidth = 0., turb_a = 1.6;
    double *varray_dev = nullptr;
    const float **rawptr_dev = nullptr;   // [2][CPOL_MAX_CLASSES]: cross-section tables, diameter axes
    int n_refr = 0;                       // refraction table (cpol_set_refraction_table)
    double *refr_keys = nullptr;
    float *refr_tab = nullptr;            // [3][n_refr][n_gates]: S, H, E
    int refr_gates = 0;
    // Device workspaces of the kernels.  `heap` holds grow-only hipMalloc'ed buffers; `arena_work` is the same set
    // of buffers carved out of the per-device static scratch array for the entry point that must not allocate
    // (cpol_run_rays_static).  `w` points at the set in use.
    struct Buf { void *p = nullptr; size_t cap = 0; };
    struct Work {
        Buf b_S, b_H, b_E, b_lat, b_lon, b_coords, b_line, b_phi, b_vals, b_mask, b_key, b_lam, b_coef,
            b_sorted, b_urec, b_rank, b_uflag, b_psd, b_psd_hi, b_dop, b_cbkt, b_n0d, b_vr, b_hist, b_stage, b_az, b_el, b_status,
            o_obs, o_rvel, o_spec, o_mask, o_lat, o_lon, o_dist, o_height, o_diag, t_a, t_b, t_c, t_d, t_e;
    };
    Work heap, arena_work;
    Work *w = &heap;
    std::vector<Buf *> bufs;              // heap buffers to free
    char  *arena = nullptr;               // non-null while a call carves its buffers from the static scratch
    size_t arena_cap = 0, arena_off = 0;
    size_t max_batch_units = (size_t)32 << 20;     // (ray x sub-beam x gate) positions per batch
    int    host_batches = 4;                       // cpol_run_rays: batches per call
    double host_last_fraction = 0.25;              // share of the rays that the last batch takes
    // cpol_run_rays_async: two sets of device output buffers used in turn, and per set the event on the copy stream
    // after which its previous contents have left for the host
    Buf async_out[2][10];
    hipEvent_t async_done[2] = {nullptr, nullptr};
    int async_slot = 0;
    // optional stage timing (hipEvents on ctx->stream) and work counters
    bool   certified_cutoff = true;               // term cutoffs below the exact-zero one where the tables carry a certificate
    bool   combine_columns = true;                // integrate the combinations of columns the observables use where that is safe (ClassTable::comb)
    float *cert_dev = nullptr;                    // result word of k_cut_certificate
    unsigned long long *work_dev = nullptr;       // 3 x CPOL_WORK_SLOTS partial counts of unit x 16-diameter steps: evaluated by k_integrate, by k_integrate_dop, and the units' own counts in k_integrate (timing only)
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    struct Span { int stage; hipEvent_t a, b; };
    std::vector<Span> spans;
    double stage_ms[CPOL_N_STAGES] = {0};
    long   stage_launches[CPOL_N_STAGES] = {0};
    int   *valid_host = nullptr;                  // pinned: valid units of each batch
    size_t valid_cap = 0, valid_used = 0;
    double valid_units = 0, dop_units = 0, comb_units = 0;
};

#define HIPCHK(ctx, call)                                                                  \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e_);                \
            return CPOL_ERR_HIP;                                                           \
        }                                                                                  \
    } while (0)

// kernels.hip
void launch_transpose_levels(hipStream_t s, const float *src, float *dst, int nz, long ncell, int stride, int offset);
void launch_coarse_levels(hipStream_t s, const float *zcol, float *zc8, int nz, long ncell);
void launch_geom(hipStream_t s, const GeomParams &gp, int n_rays, const double *az, const double *el,
                 const double *quad, double *line, float *S, float *H, float *E, double *lat, double *lon,
                 float *coords, double *phi, const ModelView &mv, int *status);
void launch_interp(hipStream_t s, const ModelView &mv, long n, const float *coords, const float *H,
                   float *vals, long plane, signed char *mask, float *raw_out);
void launch_prep(hipStream_t s, const ClassTable &ct, long nu, int n_gates, const float *vals, long plane,
                 const float *elev, signed char *mask, const double *weight_per_sub, int n_sub,
                 const double *phi_sc, int doppler, cpol_key_t *key, int *rank, float2 *lam, double *coef, double *n0d,
                 int *hist, double *vr);
void launch_build_image(hipStream_t s, const float *raw, const float *d_axis, float *img, float *xq, long nslice,
                        int ncol, double mu, double nu, int with_dop, int zero_pol, double mu_d, double alpha_d,
                        double beta_d, double x_ref);
void launch_node_floor(hipStream_t s, const float *raw, const float *d_axis, long nslice, double mu, float *floor_out);
void launch_sincos_deg(hipStream_t s, int n, const double *deg, double *sc);
void launch_scan(hipStream_t s, int nb, int nb_counted, int *hist, int *offs, int *cursor, int *chunk_off,
                 int *chunk_bucket);
void launch_scatter(hipStream_t s, long nu, int n_cls, const cpol_key_t *key, const int *rank, const int *offs,
                    const int *cursor, int *sorted, int nb_form, unsigned char *uflag);
void launch_integrate(hipStream_t s, const ClassTable &ct, long max_chunks, int nb, const int *offs,
                      const int *chunk_off, const int *chunk_bucket, const int *sorted, float4 *urec, const float2 *lam,
                      float *psd_out, float *psd_hi, unsigned long long *work);
float cut_certificate(hipStream_t s, const float *raw, const float *d_axis, long nslice, double mu, double nu, double x_ref,
                      int with_dop, double mu_d, double alpha_d, double beta_d, double cut_e, float *worst_dev);
void launch_integrate_dop(hipStream_t s, const ClassTable &ct, long max_chunks, int b_lo, int b_hi, const int *offs,
                          const int *chunk_off, const int *chunk_bucket, const int *sorted, const float2 *lam,
                          const double *n0d, float2 *dop_out, long slot0, unsigned long long *work);
void launch_build_rcs(hipStream_t s, const float *raw, float *rcs, long n);
void launch_build_dop_image(hipStream_t s, const float *raw, const float *d_axis, float *img, float *xq, long nslice,
                            double mu_d, double nu_d, double alpha_d, double beta_d, double x_ref);
struct SpecParams {
    int    n_gates, n_sub, n_vel;
    float  zfac;                   // wavelength^4 / (pi^5 KW^2) as float32
    float  d_min[3], d_max[3], step[3];
    double inv_alpha[3], inv_beta[3];   // get_D_from_V: (1/alpha * v)^(1/beta)
    float  mu[3], nu[3];
    double geo_step[3];            // diameter step when nu = 1 on an equally spaced axis, else 0
    int    chain_cut;              // end a bin's float accumulation where the remaining terms cannot change the sum
    // turbulence broadening (off when turb == 0)
    int    turb;
    double sigma_r, sigma_theta;   // radial resolution [m], 3 dB beamwidth [rad]
    double kolm_a;                 // constants.A
    double range_res;              // gate g lies at range_res * (g + 0.5) (constants.py RANGE_RADAR)
};

void launch_spectrum(hipStream_t s, const ClassTable &ct, const SpecParams &sp, int n_rays, long nu, const float *vals,
                     long plane, const float *elev, const signed char *mask, const double *wsub, const double *phi_sc,
                     const double *varray, const float *const *raw, const float *const *d_axis, double *spectrum,
                     double *rvel, long ray0);
struct Dop2In {                    // inputs of the Doppler scheme-2 reduction inside k_gate_reduce
    const float  *vals; long plane;
    const float  *elev;
    const double *phi_sc;
    const float2 *lam;
    const double *n0d;
    const float2 *dop_out;         // results of the separate Doppler units, indexed like their polarimetric twins
    int n_cls, ne_max, nt_max, shared;
    int comb, nb_form;             // as in ObsParams
    const float *psd_hi;           // columns 8.. of the units that take all columns (shared: 12, 13 at [4], [5] of 8 floats)
    int slice_floats;              // floats per (e, t) node of that image
    int at0[2], at1[2];            // offsets of the two Doppler columns at d = 0 and d = 1023 inside a node
    const float *img[CPOL_MAX_CLASSES];   // table image holding the Doppler columns
    const float *xq[CPOL_MAX_CLASSES];
    int n_t[CPOL_MAX_CLASSES];
};
struct FrontOut {
    float  *E;                 // [nu] elevation at the point
    signed char *mask;         // [nu] interpolation code
    float  *vals; long plane;  // variable planes, written for the variables in write_mask
    unsigned write_mask;
    float  *dist, *height;     // outputs of the central sub-beam (any may be NULL), rows of n_gates from ray0 on
    double *lat, *lon;
    long   ray0;
    int    *status;            // per ray of the batch
    cpol_key_t *key; int *rank; float2 *lam; double *coef; double *n0d; int *hist;
};

void launch_model_range(hipStream_t s, const float *dcol, size_t n_val, const float *zcol, int nz, size_t n_z, unsigned *flag);
void launch_front(hipStream_t s, const GeomParams &gp, const ModelView &mv, const ClassTable &ct, int n_rays,
                  const double *az, const double *el, const double *quad, double *line, double *phi,
                  const double *wsub, int doppler, const FrontOut &fo);
void launch_vr1(hipStream_t s, const ClassTable &ct, long nu, int n_gates, const float *vals, long plane,
                const float *elev, const double *phi_sc, double *vr);
void launch_observables(hipStream_t s, const ObsParams &op, int n_rays, long nu, const cpol_key_t *key,
                        const double *coef, const float *psd_out, const float *psd_hi, const signed char *mask,
                        const double *weight_per_sub, const double *vr, float *obs, long obs_plane,
                        long ray0, double *mask_out, double *rvel_out, const Dop2In &d2, double *spectrum, int n_vel);
struct DiagVars { int n; int var[CPOL_MAX_VARS]; };      // uploaded variables in upload order -> cpol_var slot
void launch_diag(hipStream_t s, int n_rays, int n_sub, int n_gates, const DiagVars &dv, const float *vals, long plane,
                 const double *wsub, double sum_w, double *diag, long diag_plane, long ray0);
void launch_gather_central(hipStream_t s, int n_rays, int n_sub, int n_gates, const float *S, const float *H,
                           const double *lat, const double *lon, float *dist, float *height, double *olat,
                           double *olon, long ray0);
void launch_get_refl(hipStream_t s, float *refl, int nbins, const float *Da, const float *Db, const float *D,
                     const float *rcs, const float *N0, const float *mu, const float *lam, const float *nu,
                     const float *step, const float *dmin);
