/* cosmopol.h -- C ABI of libcosmopol_hip.so: the MI355X (gfx950) implementation of the
 * cosmo_pol per-gate forward-operator hot path.
 *
 * Plain C types only; every function returns an int status (CPOL_OK = 0) unless it
 * mirrors a reference symbol.  Pointers are HOST pointers unless a parameter name ends
 * in _dev.  The library owns all device memory inside a cpol_ctx; a ctx belongs to one
 * device and must not be used from two threads at once.  No host pointer is retained
 * after a call returns.
 *
 * Each entry point cites the reference interface it stands in for (paths relative to
 * the reference root).
 */
#ifndef COSMOPOL_H
#define COSMOPOL_H

#ifdef __cplusplus
extern "C" {
#endif

#define CPOL_OK 0
#define CPOL_ERR_ARG 1        /* bad argument / shape mismatch */
#define CPOL_ERR_HIP 2        /* a HIP runtime call failed (see cpol_last_error) */
#define CPOL_ERR_STATE 3      /* model / LUT / config not uploaded yet */
#define CPOL_ERR_DOMAIN 4     /* a gate lies outside the model domain (reference: IndexError,
                                 interpolation/interpolation.py:359-363) */
#define CPOL_ERR_REFRACTION 5 /* a sub-beam's elevation has no profile in the refraction table */

#define CPOL_MAX_CLASSES 4    /* R, S, G, H (scatter/scattering_sz.py:32-35) */
#define CPOL_MAX_VARS 16
#define CPOL_N_SZ 12          /* scattering components per diameter (lookup/compute_lut_sz.py:83-94) */
#define CPOL_N_OBS 9          /* ZH ZDR ZV KDP DELTA_HV PHIDP RHOHV AH AV (scattering_sz.py:130-139) */

typedef struct cpol_ctx cpol_ctx;

/* Variable slots of the model state (radar_operator.py:32-33, BASE_VARIABLES[_2MOM]). */
enum cpol_var {
    CPOL_V_U = 0, CPOL_V_V, CPOL_V_W, CPOL_V_QR, CPOL_V_QS, CPOL_V_QG, CPOL_V_RHO, CPOL_V_T,
    CPOL_V_QH, CPOL_V_QNH, CPOL_V_QNR, CPOL_V_QNS, CPOL_V_QNG,
    CPOL_V_EDR,           /* eddy dissipation rate, read by the turbulence broadening only (doppler_sz.py:36,95) */
    CPOL_V_COUNT
};

/* Order of the observables in every output array. */
enum cpol_obs {
    CPOL_O_ZH = 0, CPOL_O_ZDR, CPOL_O_ZV, CPOL_O_KDP, CPOL_O_DELTA_HV, CPOL_O_PHIDP,
    CPOL_O_RHOHV, CPOL_O_AH, CPOL_O_AV
};

/* POD mirror of the utilities/cfg.py CONFIG keys the hot path reads, plus the derived
 * constants of constants/constants.py:73-77. */
typedef struct {
    double radar_lat, radar_lon, radar_alt;   /* radar.coords */
    double range_res;                         /* radar.radial_resolution [m] */
    double earth_radius;                      /* refraction/atm_refraction.py:18-21 evaluated as at :114-117;
                                                 0 = let the library evaluate it */
    int    n_gates;                           /* len(arange(res/2, range, res)) */
    double wavelength_mm;                     /* constants.WAVELENGTH */
    double kw;                                /* constants.KW */
    int    attenuation_correction;            /* attenuation.correction */
    int    microphysics_scheme;               /* 1 or 2 */
    int    doppler_scheme;                    /* 0 = skip, 1, 2, 3 (3 needs cpol_set_velocity_bins and 1-moment input) */
    int    sensitivity_mode;                  /* 0 = no cut, 1 = [ZH], 2 = [ZH, range], 3 = [sensitivity, gain, SNR]
                                                 (utilities.py:74-81).  Mode 3: sensitivity[0] = sensitivity +
                                                 constants.RADAR_CONSTANT_DB (constants.py:80-82) + SNR, summed in
                                                 that order by the caller */
    double sensitivity[2];
} cpol_config;

/* Per-class particle-size-distribution constants (hydrometeors/hydrometeors.py:177-438,
 * constants/constants_1mom.py, constants_2mom.py).  psd_kind selects the set_psd branch:
 *   0: 1-moment rain / graupel   lambda = (lambda_factor / Q)^(1/(4+mu)),  N0 fixed
 *   1: 1-moment snow             N0(T), lambda = (a N0 lambda_factor / Q)^(1/(b+1))
 *   2: 2-moment (any class)      x = clamp(Q/(QN+eps)), lambda, N0 from (QN, Q)
 * The *_dop members describe the class as doppler_sz.py:57 instantiates it (always
 * 1-moment except hail). */
typedef struct {
    int    psd_kind;
    double a, b, alpha, beta, mu, nu;
    double lambda_factor, vel_factor, ntot_factor;
    double n0_fixed;
    double x_min, x_max;
    double d_min, d_max;
    int    q_var, qn_var;          /* cpol_var slots of the mass / number density arguments of set_psd (-1 if unused) */
    int    t_var;                  /* slot of the temperature argument of the 1-moment snow branch (normally CPOL_V_T;
                                      with 2-moment input doppler_sz.py:143 hands QNS_v to that argument) */
} cpol_psd;

typedef struct {
    int     n_e, n_t, n_d;          /* table is [n_e][n_t][n_d][12] float32 */
    float   e_min, e_step;          /* axes_limits[0], axes_step of lookup/lut.py:114-115 (float32) */
    float   t_min, t_step;
    const float *d_axis;            /* n_d diameters (float32) */
    const float *table;
} cpol_lut;

/* ---- life cycle ------------------------------------------------------------------- */
int  cpol_create(int device, cpol_ctx **ctx);
int  cpol_destroy(cpol_ctx *ctx);
const char *cpol_last_error(cpol_ctx *ctx);
/* HIP stream all kernels of this ctx are launched on (a hipStream_t), for event timing. */
void *cpol_stream(cpol_ctx *ctx);
/* Waits for every kernel and every device-to-host delivery queued by this ctx. */
int  cpol_synchronize(cpol_ctx *ctx);

/* ---- measurement -------------------------------------------------------------------- */
/* Stage timing with HIP events recorded on the ctx stream around each group of kernels
 * (no counterpart in the reference, which only has utilities/tictoc.py).  While enabled,
 * every cpol_run_rays* / cpol_observables_from_beams call adds to the totals; read them
 * after cpol_synchronize().  valid_units = number of (gate, sub-beam, class) units that
 * went through the PSD integration of the scattering components (stage INTEGRATE), the
 * quantity its arithmetic scales with; doppler_units = units of the Doppler scheme-2
 * moments that were integrated separately because their PSD differs (stage INTEGRATE_DOP;
 * 0 when the moments ride on the polarimetric units). */
/* cpol_run_rays* works the per-position front (geometry of the points, interpolation, PSD set-up) in one kernel and
 * books it under INTERP, the scans and the scatter of the bucket sort under PREP_SORT; GEOM and the PSD set-up share
 * of PREP_SORT are filled by cpol_geolocate / cpol_run_profiles, which run those parts on their own. */
#define CPOL_N_STAGES 7
enum cpol_stage { CPOL_ST_GEOM = 0, CPOL_ST_INTERP, CPOL_ST_PREP_SORT, CPOL_ST_INTEGRATE, CPOL_ST_OBSERVABLES,
                  CPOL_ST_INTEGRATE_DOP, CPOL_ST_SPECTRUM /* Doppler spectrum, doppler_scheme 3 */ };
int cpol_timing_enable(cpol_ctx *ctx, int on);          /* also resets the totals */
int cpol_timing_read(cpol_ctx *ctx, double stage_ms[CPOL_N_STAGES], long stage_launches[CPOL_N_STAGES],
                     double *valid_units, double *doppler_units);
/* Number of (unit, diameter) terms stage INTEGRATE evaluated while timing was enabled.  A unit's
 * N(D) = exp(-lambda D^nu) underflows to an exact float32 zero beyond some diameter; units are ordered by
 * that diameter inside their bucket and a wavefront stops at the last 16-diameter step any of its units
 * still needs, so this is less than 1024 x valid_units (no counterpart in the reference, which evaluates
 * all 1024 terms in float64, scattering_sz.py:95-103).  doppler_unit_diameters (may be NULL): the same for
 * stage INTEGRATE_DOP, where a unit whose N0 is an exact zero needs no term at all.  own_unit_diameters (may be
 * NULL): the terms up to each unit's own stopping point, i.e. what stage INTEGRATE would evaluate if no unit had to
 * run on to the stopping point of the last unit of its group of 32. */
int cpol_timing_work(cpol_ctx *ctx, double *evaluated_unit_diameters, double *doppler_unit_diameters,
                     double *own_unit_diameters);
/* How many of the valid units counted by cpol_timing_read were integrated over the combined columns (see
 * cpol_set_column_combination) while timing was enabled.  No counterpart in the reference. */
int cpol_timing_forms(cpol_ctx *ctx, double *combined_units);
/* Which terms of the PSD integration (scattering_sz.py:95-103: all 1024 diameters, float64) are evaluated.  The
 * terms whose N(D) = exp(-lambda D^nu) underflows to an exact float32 zero are always left out.  With
 * certified != 0 (the default) a class may stop earlier, at N(D) < 2^-E N(D_0), E in {48 .. 128}: when its tables
 * are uploaded the library computes, for every table node, scattering component and stopping point, the largest
 * share of the sum that any lambda can leave beyond that point, and takes the smallest E for which this share is
 * below 2^-28 (a sixteenth of float32's resolution) everywhere; tables that admit no such E keep the exact-zero
 * rule.  certified = 0 switches to the exact-zero rule for every class.  cpol_get_term_cutoff returns E per class
 * (150 = exact zeros only, 0 = no table uploaded).  The same switch governs a shortcut of the Doppler spectrum
 * (doppler_c.c:24-26, a float accumulator per velocity bin): with certified != 0 the accumulation of a long bin
 * ends where every term still to come is below 2^-27 of the float sum, which none of them can change - the
 * spectrum is bit for bit the one of certified = 0.  No counterpart in the reference. */
int cpol_set_term_cutoff(cpol_ctx *ctx, int certified);
int cpol_get_term_cutoff(cpol_ctx *ctx, float exponent[CPOL_MAX_CLASSES]);
/* Over which columns the PSD integration runs.  The observables are functions of seven linear combinations of the
 * twelve scattering components (get_pol_from_sz, scattering_sz.py:168-194: s0 - s1 - s2 + s3, s0 + s1 + s2 + s3,
 * s4 + s7, s6 - s5, s10 - s8, s11, s9), and integration is linear, so a unit (gate x sub-beam x class) may be
 * integrated over those seven columns of the table instead of all twelve.  What the reference does with the twelve
 * sums that this would lose is the rule of scattering_sz.py:110 - a component whose sum over sub-beams and classes is
 * an exact zero becomes NaN - and the float32 storage of each component after every sub-beam (:38,105) where that
 * storage underflows.  With on != 0 (the default) a unit therefore takes the combined columns only when neither can
 * touch it: every column of its table node has one sign along the whole diameter axis, so that the first diameter's
 * term bounds the magnitude of each of its twelve sums from below, and the unit's coefficient times the smallest such
 * term of the node is at least 2^-100 - its contribution to every component is then a normal float32 far from the
 * subnormals.  Every other unit is integrated over all twelve columns, and a gate that holds no unit of the first
 * kind is reduced exactly as the reference reduces it.  A gate that holds one is reduced from the combinations,
 * accumulated in float64: none of its components can be an exact zero unless nonzero contributions cancel exactly,
 * which needs unrelated floats to be exact negatives of each other and is the one case not reproduced; its
 * observables differ from the reference's float32 accumulation of the components by that accumulation's own rounding
 * (a few 1e-7 of a component).  on = 0: all units over all twelve columns.  When the two Doppler scheme-2 moments ride
 * on the polarimetric units (1-moment input, doppler_sz.py:192,218-220) the spare eighth column carries the second,
 * sum N rcs v, and the first, sum N rcs, is 2 pi times the first combination, rcs = 2 pi (s0 - s1 - s2 + s3).
 * cpol_get_column_combination reports per class how many table nodes qualify (all 0: no unit takes the combined
 * columns).  No counterpart in the reference. */
int cpol_set_column_combination(cpol_ctx *ctx, int on);
int cpol_get_column_combination(cpol_ctx *ctx, long nodes_per_class[CPOL_MAX_CLASSES]);

/* 1 when the uploaded model lets the vertical interpolation (interpolation_c.c:151, four float32 divisions per
 * point, column and variable) share one refined reciprocal per level pair - the same bits as the division as long
 * as every variable is zero or finite with a magnitude in [2^-78, 2^85] and every level is 2^-10 .. 2^14 m thick,
 * which cpol_upload_model checks on the device; 0 when a value lies outside and every quotient is a full division;
 * negative when no model has been uploaded.  No counterpart in the reference: a diagnostic of this library. */
int cpol_model_range_certified(cpol_ctx *ctx);
/* Upper bound of (ray x sub-beam x gate) positions processed per internal batch; bounds the
 * workspace the library allocates (default 32 Mi positions, about 26 GB with 2-moment input and Doppler scheme 2,
 * allocated only as far as a call needs it; at most 128 Mi). */
int cpol_set_batch_units(cpol_ctx *ctx, long max_units);
/* Batch schedule of cpol_run_rays, which delivers the outputs to host memory the way the reference's pool hands the
 * beams back to the parent (radar_operator.py:255-256): the outputs of one batch of rays travel to the host beside the
 * kernels of the next, so only the last batch's copy is not hidden.  A call is split into n_batches batches, the last
 * one taking last_fraction of the rays (about 1 Mi positions at least), the others equal shares of the rest and never
 * more than cpol_set_batch_units allows; n_batches = 1: no split.  The results do not depend on it.  No counterpart in
 * the reference. */
int cpol_set_host_batching(cpol_ctx *ctx, int n_batches, double last_fraction);

/* ---- state -------------------------------------------------------------------------- */
/* utilities/cfg.py CONFIG + constants.update() (radar_operator.py:39-40). */
int cpol_set_config(cpol_ctx *ctx, const cpol_config *cfg);

/* Model state: what radar_operator.py:86-151 (load_model_file) keeps in dic_vars.
 * var_ids[i] is the cpol_var slot of data[i]; every array is [nz][ny][nx] float32 with
 * level 0 = model top.  llc/urc = (Lo1, La1)/(Lo2, La2), res = (dlon, dlat),
 * pole = (Latitude_of_southern_pole, Longitude_of_southern_pole). */
int cpol_upload_model(cpol_ctx *ctx, int nvars, const int *var_ids, const float *const *data,
                      const float *zlev, int nz, int ny, int nx,
                      const float llc[2], const float urc[2], const float res[2], const double pole[2]);

/* One hydrometeor class: its lookup table (lookup/lut.py:79-245, `lut_sz[h]`) and its PSD
 * constants for the polarimetric path (psd) and for the Doppler path (psd_doppler).
 * cls = 0..3 for R, S, G, H. */
int cpol_upload_lut(cpol_ctx *ctx, int cls, const cpol_lut *lut, const cpol_psd *psd,
                    const cpol_psd *psd_doppler);

/* Velocity bins of the Doppler spectrum: constants.VARRAY = arange(-NVEL, NVEL + VRES, VRES)
 * (constants/constants.py:73-75).  Needed for doppler_scheme 3. */
int cpol_set_velocity_bins(cpol_ctx *ctx, int n_vel, const double *varray);

/* Turbulence broadening of the Doppler spectrum (doppler_sz.py:94-96, get_turb_std :308-321,
 * turb_spectrum_spread :323-335).  With enable != 0 and doppler_scheme 3 every sub-beam's weighted spectrum is
 * convolved along the velocity axis with a Gaussian whose standard deviation follows Doviak and Zrnic from the
 * eddy dissipation rate (model variable CPOL_V_EDR, which must have been uploaded), the gate's range, the
 * radial resolution and the 3 dB beamwidth, and rescaled to its power before the convolution.  The reference
 * cannot reach this code (it asks for scheme 4 inside the scheme-3 branch) and its formula relies on Python-2
 * integer division; this entry point implements the formula with true division, one standard deviation per gate,
 * the filter being scipy.ndimage.gaussian_filter1d's (reflecting edges, truncated at four standard deviations).
 * beamwidth_deg = radar.3dB_beamwidth, kolmogorov_a = constants.A (1.6). */
int cpol_set_turbulence(cpol_ctx *ctx, int enable, double beamwidth_deg, double kolmogorov_a);

/* Antenna quadrature: the kept sub-beams of interpolation/interpolation.py:260-279 in
 * reference order, as offsets (degrees) added to the ray's azimuth / elevation and
 * normalised weights.  The central sub-beam is index n_sub/2 (scattering_sz.py:25). */
int cpol_set_quadrature(cpol_ctx *ctx, int n_sub, const double *az_off, const double *el_off,
                        const double *weight);

/* Refraction scheme 2 (refraction/atm_refraction.py:45-84): the ray equation through the model's
 * refractivity profile depends on the total elevation of a sub-beam only, so the host solves it once
 * per distinct (ray elevation + vertical sub-beam offset) and passes the result here: n_profiles
 * strictly ascending keys (degrees, exactly el + el_off in float64) and S, H, E as
 * [n_profiles][n_gates] float32 in the reference's units.  Rays whose sub-beams find no key get
 * CPOL_ERR_REFRACTION in ray_status.  n_profiles = 0 returns to the built-in 4/3 earth-radius
 * model (atm_refraction.py:110-123), which is the default. */
int cpol_set_refraction_table(cpol_ctx *ctx, int n_profiles, const double *elevation_keys, const float *S,
                              const float *H, const float *E);

/* ---- the hot path ------------------------------------------------------------------- */
/* One call = what the pool workers of radar_operator.py:237-256 do for n_rays rays:
 * get_profiles_GH -> get_doppler_velocity -> get_radar_observables (+ cut_at_sensitivity
 * when sensitivity_mode != 0).  Outputs (any may be NULL) are [.. n_rays][n_gates]:
 *   obs   float32 [CPOL_N_OBS][n_rays][n_gates]   linear units, NaN where undefined
 *   rvel  float64 [n_rays][n_gates]               (doppler_scheme 1 or 2)
 *   mask  float64 [n_rays][n_gates]               scattering_sz.py:144-151
 *   lat, lon float64, dist, height float32 [n_rays][n_gates]   central sub-beam geometry
 * ray_status int32 [n_rays]: CPOL_OK, CPOL_ERR_DOMAIN or CPOL_ERR_REFRACTION per ray.
 * cpol_run_rays copies to host buffers.  cpol_run_rays_dev works on device memory: members
 * of dev_out that are non-NULL on entry are caller-owned device buffers and are filled in
 * place; NULL members are served from ctx-owned buffers (valid until the next call) whose
 * addresses are stored into dev_out.  Work is queued on the ctx stream; call
 * cpol_synchronize() before reading. */
typedef struct {
    float  *obs;
    double *rvel;
    double *mask;
    double *lat, *lon;
    float  *dist, *height;
    int    *ray_status;
    double *dspectrum;      /* float64 [n_rays][n_gates][n_vel], doppler_scheme 3 only (doppler_sz.py:91-97) */
    double *diag;           /* float64 [n_vars][n_rays][n_gates], the reference's diagnostic mode (radar_operator.py:245-246,
                               interpolation.py:21-51 integrate_GH_pts): every uploaded model variable, in upload order,
                               as the weighted NaN-aware sum over the sub-beams, sum_b nan0(v_b w_b / sum w).  Unlike
                               the other members NULL means "not wanted", also for cpol_run_rays_dev */
} cpol_outputs;

int cpol_run_rays(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                  cpol_outputs *host_out);
int cpol_run_rays_dev(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                      cpol_outputs *dev_out);

/* cpol_run_rays for a caller that simulates one scan after the other (a sequence of model states, the radars of a
 * network): the call returns as soon as its kernels and its device-to-host copies are queued, so the outputs of one
 * scan travel to the host beside the kernels of the next - the pool of radar_operator.py:253-256 kept busy across
 * scans.  host_out must stay valid, and should be page-locked (a copy into pageable memory makes the call wait for
 * it), until cpol_synchronize() has returned, which waits for the kernels and the deliveries of every call so far;
 * ray_status is to be read after that.  azimuth / elevation may be host or device arrays and are read before the
 * call returns.  Two sets of device output buffers are used in turn, so any number of calls may be queued: the
 * third waits on the device for the first one's delivery. */
int cpol_run_rays_async(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                        cpol_outputs *host_out);
/* Waits until the outputs of the most recent cpol_run_rays_async call (calls_back = 0) or of the one before it
 * (calls_back = 1) are in host memory: with two host buffer sets a consumer queues scan k, waits for scan k - 1 and
 * reads it while scan k runs. */
int cpol_wait_async(cpol_ctx *ctx, int calls_back);

/* The same hot path for callers that cannot allocate: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.  Same computation and same outputs as
 * cpol_run_rays_dev, bit for bit, but the call allocates nothing: every workspace is a piece of one array of
 * 80 MiB that the library keeps in the device memory of each GPU it runs on until the process ends.  For that:
 *   - n_rays x sub-beams x n_gates must not exceed CPOL_STATIC_MAX_POSITIONS (CPOL_ERR_ARG otherwise).  That is the
 *     only size limit: a position takes at most 726 bytes of the array (2-moment input with Doppler scheme 2), 47.5 MB
 *     at the limit, and outputs served from the array at most 76 bytes per gate; the sort counters take 520 bytes per
 *     bucket (lookup-table node x unit kind x unit form), and the library accepts at most 32768 buckets (17 MB; 15 MB
 *     for the reference's 46 x 39-node tables of four classes with Doppler units of their own and two forms of
 *     polarimetric unit), so everything together stays below the 80 MiB whatever the tables;
 *   - azimuth_dev / elevation_dev are DEVICE arrays [n_rays] and are read in place;
 *   - cpol_set_config, cpol_upload_lut, cpol_upload_model and cpol_set_quadrature (which allocate and build the
 *     table images) must have been called before; doppler_scheme 3 is not served (CPOL_ERR_ARG);
 *   - members of dev_out that are NULL on entry are served from the same array and stay valid until the next
 *     call of this entry point on that GPU; stage timing is not recorded for these calls.
 * All calls on one GPU, from any ctx, share that GPU's array: calls that overlap on one GPU race, so run them one
 * after the other (same stream, or synchronise in between). */
#define CPOL_STATIC_MAX_POSITIONS 65536
int cpol_run_rays_static(cpol_ctx *ctx, int n_rays, const double *azimuth_dev, const double *elevation_dev,
                         cpol_outputs *dev_out);

/* ---- stage-level entry points (unit parity against the reference functions) ---------- */
/* refraction/atm_refraction.py:110-123 (ref_4_3), pyproj Geod.fwd at
 * interpolation/interpolation.py:316-325 and the rotated-pole transform of :355 for every
 * (ray, sub-beam, gate).  Arrays are [n_rays][n_sub][n_gates]; coords_rot is [..][2] =
 * (rlat, rlon) float32, the exact operand the reference hands to get_all_radar_pts. */
int cpol_geolocate(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                   float *dist, float *height, float *elev, double *lat, double *lon,
                   float *coords_rot, int *ray_status);

/* interpolation/interpolation_c.c:12 (get_all_radar_pts) on the uploaded model for n points
 * and every uploaded variable: out is [nvars][n] in upload order; -9999 above the domain,
 * NaN below terrain, exactly like the reference. */
int cpol_interp_points(cpol_ctx *ctx, int n, const float *coords_rot, const float *heights, float *out);

/* scatter/scattering_sz.py:17 (get_radar_observables) on caller-supplied sub-beam profiles,
 * i.e. on the list[Beam] that get_profiles_GH returns.  values is [CPOL_V_COUNT][n_sub][n_gates]
 * (slots that the scheme does not use are ignored), elev/mask are [n_sub][n_gates], weight
 * [n_sub].  obs float32 [CPOL_N_OBS][n_gates], mask_out float64 [n_gates].
 * With doppler_scheme 1, 2 or 3, rvel float64 [n_gates] receives doppler_sz.py:26
 * (get_doppler_velocity); az_sub [n_sub] is GH_pt[0] of each sub-beam.  With scheme 3,
 * dspectrum float64 [n_gates][n_vel] (may be NULL) receives the Doppler spectrum. */
int cpol_observables_from_beams(cpol_ctx *ctx, int n_sub, int n_gates, const float *values,
                                const float *elev, const double *mask, const double *weight,
                                const double *az_sub, float *obs, double *mask_out, double *rvel,
                                double *dspectrum);

/* ---- drop-in replacements of the reference's SWIG/C symbols --------------------------- */
/* interpolation/interpolation_c.c:8,12 -- identical signature; runs on device 0 through a
 * process-wide ctx (the model arrays are uploaded on every call, as SWIG converts them on
 * every call in the reference, interpolation_c_wrap.c:3693-3762). */
float *get_all_radar_pts(float *output, int len, float *coords_rad_pts, int crp_x, int crp_y,
                         float *radar_heights, int rh_x, float *model_data, int md_x, int md_y, int md_z,
                         float *model_heights, int mh_x, int mh_y, int mh_z,
                         float *llc_cosmo, int llc_cosmo_x, float *res_cosmo, int res_cosmo_x);

/* doppler/doppler_c.c:9,11 -- identical signature (three classes, stride 3). */
float *get_refl(float *refl, int len, float *Da, int len_D_bins_x, int o2, float *Db, int o3, int o4,
                float *D, int o5, int o6, float *rcs, int o8, int o9, float *N0, int o10,
                float *mu, int o11, float *lambda, int o12, float *nu, int o13,
                float *step_D, int o14, float *Dmin, int o15);

#ifdef __cplusplus
}
#endif
#endif /* COSMOPOL_H */
