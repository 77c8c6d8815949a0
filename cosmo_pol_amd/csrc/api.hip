// C ABI of libcosmopol_hip.so (declared in include/cosmopol.h): context, uploads and
// the orchestration of the kernels in kernels.hip.  No Python / torch types here.
#include "common.h"
#include <math.h>
#include <string.h>
#include <mutex>
#include <algorithm>

static int fail(cpol_ctx *ctx, int code, const char *msg);

// Make sure workspace b holds at least `bytes`.  Normally a grow-only hipMalloc'ed buffer; while a call runs on
// the static scratch (ctx->arena set) the buffer is the next 256-byte aligned piece of that array instead and
// nothing is allocated.
static int ensure(cpol_ctx *ctx, cpol_ctx::Buf &b, size_t bytes)
{
    if (bytes == 0) bytes = 16;
    if (ctx->arena) {
        if (b.cap >= bytes) return CPOL_OK;
        const size_t off = (ctx->arena_off + 255) & ~(size_t)255;
        if (off + bytes > ctx->arena_cap)
            return fail(ctx, CPOL_ERR_ARG, "the call needs more workspace than the static scratch array holds");
        b.p = ctx->arena + off;
        b.cap = bytes;
        ctx->arena_off = off + bytes;
        return CPOL_OK;
    }
    if (b.cap >= bytes) return CPOL_OK;
    if (b.p) HIPCHK(ctx, hipFree(b.p));
    b.p = nullptr; b.cap = 0;
    size_t want = bytes + bytes / 8;
    HIPCHK(ctx, hipMalloc(&b.p, want));
    b.cap = want;
    bool known = false;
    for (auto *q : ctx->bufs) known |= (q == &b);
    if (!known) ctx->bufs.push_back(&b);
    return CPOL_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != CPOL_OK) return rc_; } while (0)

static int build_class_table(cpol_ctx *ctx, ClassTable &ct, bool allow_build);

// ---- stage timing -------------------------------------------------------------------
static hipEvent_t next_event(cpol_ctx *ctx)
{
    if (ctx->ev_used == ctx->ev_pool.size()) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        ctx->ev_pool.push_back(e);
    }
    return ctx->ev_pool[ctx->ev_used++];
}

struct StageTimer {
    cpol_ctx *ctx; int stage; hipEvent_t a = nullptr;
    StageTimer(cpol_ctx *c, int st) : ctx(c), stage(st)
    {
        // no events (their creation allocates) while a call runs on the static scratch
        if (ctx->timing && !ctx->arena && (a = next_event(ctx))) (void)hipEventRecord(a, ctx->stream);
    }
    ~StageTimer()
    {
        if (!a) return;
        hipEvent_t b = next_event(ctx);
        if (!b) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->spans.push_back({stage, a, b});
    }
};

static void timing_collect(cpol_ctx *ctx)
{
    for (auto &sp : ctx->spans) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess) {
            ctx->stage_ms[sp.stage] += ms;
            ctx->stage_launches[sp.stage] += 1;
        }
    }
    ctx->spans.clear();
    ctx->ev_used = 0;
    for (size_t i = 0; i < ctx->valid_used; i++) {
        ctx->valid_units += ctx->valid_host[3 * i];
        ctx->dop_units += ctx->valid_host[3 * i + 1] - ctx->valid_host[3 * i];
        ctx->comb_units += ctx->valid_host[3 * i + 2];
    }
    ctx->valid_used = 0;
}

extern "C" int cpol_timing_enable(cpol_ctx *ctx, int on)
{
    if (!ctx) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    timing_collect(ctx);
    const size_t work_bytes = 3 * CPOL_WORK_SLOTS * CPOL_WORK_STRIDE * sizeof(unsigned long long);
    if (on && !ctx->work_dev) HIPCHK(ctx, hipMalloc((void **)&ctx->work_dev, work_bytes));
    if (ctx->work_dev) HIPCHK(ctx, hipMemset(ctx->work_dev, 0, work_bytes));
    ctx->timing = on != 0;
    for (int i = 0; i < CPOL_N_STAGES; i++) { ctx->stage_ms[i] = 0; ctx->stage_launches[i] = 0; }
    ctx->valid_units = 0; ctx->dop_units = 0; ctx->comb_units = 0;
    return CPOL_OK;
}

extern "C" int cpol_timing_read(cpol_ctx *ctx, double stage_ms[CPOL_N_STAGES], long stage_launches[CPOL_N_STAGES],
                                double *valid_units, double *doppler_units)
{
    if (!ctx) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    timing_collect(ctx);
    for (int i = 0; i < CPOL_N_STAGES; i++) {
        if (stage_ms) stage_ms[i] = ctx->stage_ms[i];
        if (stage_launches) stage_launches[i] = ctx->stage_launches[i];
    }
    if (valid_units) *valid_units = ctx->valid_units;
    if (doppler_units) *doppler_units = ctx->dop_units;
    return CPOL_OK;
}

extern "C" int cpol_timing_work(cpol_ctx *ctx, double *evaluated_unit_diameters, double *doppler_unit_diameters,
                                double *own_unit_diameters)
{
    if (!ctx || !evaluated_unit_diameters) return CPOL_ERR_ARG;
    *evaluated_unit_diameters = 0.;
    if (doppler_unit_diameters) *doppler_unit_diameters = 0.;
    if (own_unit_diameters) *own_unit_diameters = 0.;
    if (!ctx->work_dev) return CPOL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<unsigned long long> w(3 * CPOL_WORK_SLOTS * CPOL_WORK_STRIDE);
    HIPCHK(ctx, hipMemcpy(w.data(), ctx->work_dev, w.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double tot[3] = {0., 0., 0.};
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < CPOL_WORK_SLOTS; i++) tot[k] += (double)w[(size_t)(k * CPOL_WORK_SLOTS + i) * CPOL_WORK_STRIDE];
    *evaluated_unit_diameters = 16. * tot[0];
    if (doppler_unit_diameters) *doppler_unit_diameters = 16. * tot[1];
    if (own_unit_diameters) *own_unit_diameters = 16. * tot[2];
    return CPOL_OK;
}

extern "C" int cpol_timing_forms(cpol_ctx *ctx, double *combined_units)
{
    if (!ctx || !combined_units) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    timing_collect(ctx);
    *combined_units = ctx->comb_units;
    return CPOL_OK;
}

extern "C" int cpol_set_term_cutoff(cpol_ctx *ctx, int certified)
{
    if (!ctx) return CPOL_ERR_ARG;
    ctx->certified_cutoff = certified != 0;
    for (auto &l : ctx->lut) l.image_ncol = 0;          // images and certificates are redone at the next set-up / run
    if (ctx->have_cfg) {
        bool all = true;
        const int n_cls = ctx->cfg.microphysics_scheme == 1 ? 3 : 4;
        for (int h = 0; h < n_cls; h++) all = all && ctx->lut[h].present;
        if (all) {
            HIPCHK(ctx, hipSetDevice(ctx->device));
            ClassTable ct;
            return build_class_table(ctx, ct, true);
        }
    }
    return CPOL_OK;
}

extern "C" int cpol_set_column_combination(cpol_ctx *ctx, int on)
{
    if (!ctx) return CPOL_ERR_ARG;
    ctx->combine_columns = on != 0;
    for (auto &l : ctx->lut) l.image_ncol = 0;          // images are redone at the next set-up / run
    if (ctx->have_cfg) {
        bool all = true;
        const int n_cls = ctx->cfg.microphysics_scheme == 1 ? 3 : 4;
        for (int h = 0; h < n_cls; h++) all = all && ctx->lut[h].present;
        if (all) {
            HIPCHK(ctx, hipSetDevice(ctx->device));
            ClassTable ct;
            return build_class_table(ctx, ct, true);
        }
    }
    return CPOL_OK;
}

extern "C" int cpol_get_column_combination(cpol_ctx *ctx, long nodes_per_class[CPOL_MAX_CLASSES])
{
    if (!ctx || !nodes_per_class) return CPOL_ERR_ARG;
    for (int h = 0; h < CPOL_MAX_CLASSES; h++) nodes_per_class[h] = 0;
    if (!ctx->have_cfg) return CPOL_OK;
    const int n_cls = ctx->cfg.microphysics_scheme == 1 ? 3 : 4;
    for (int h = 0; h < n_cls; h++)
        if (!ctx->lut[h].present) return CPOL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ClassTable ct;
    TRY(build_class_table(ctx, ct, true));
    if (ct.comb)
        for (int h = 0; h < n_cls; h++) nodes_per_class[h] = ctx->lut[h].nodes_comb;
    return CPOL_OK;
}

extern "C" int cpol_model_range_certified(cpol_ctx *ctx)
{
    if (!ctx) return CPOL_ERR_ARG;
    if (!ctx->have_model) return fail(ctx, CPOL_ERR_STATE, "no model uploaded");
    return ctx->fast_div ? 1 : 0;
}

extern "C" int cpol_get_term_cutoff(cpol_ctx *ctx, float exponent[CPOL_MAX_CLASSES])
{
    if (!ctx || !exponent) return CPOL_ERR_ARG;
    for (int h = 0; h < CPOL_MAX_CLASSES; h++) exponent[h] = ctx->lut[h].present ? ctx->lut[h].cut_e - 2.f : 0.f;
    return CPOL_OK;
}

extern "C" int cpol_set_batch_units(cpol_ctx *ctx, long max_units)
{
    if (!ctx || max_units < 1) return CPOL_ERR_ARG;
    // unit slots (2 kinds x classes x positions) are addressed with 32-bit integers
    if (max_units > (128L << 20)) return fail(ctx, CPOL_ERR_ARG, "at most 128 Mi positions per batch");
    ctx->max_batch_units = (size_t)max_units;
    return CPOL_OK;
}

extern "C" int cpol_set_host_batching(cpol_ctx *ctx, int n_batches, double last_fraction)
{
    if (!ctx || n_batches < 1 || n_batches > 64 || !(last_fraction >= 0.) || last_fraction > 1.) return CPOL_ERR_ARG;
    ctx->host_batches = n_batches;
    ctx->host_last_fraction = last_fraction;
    return CPOL_OK;
}

static int fail(cpol_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

// Workspace of cpol_run_rays_static, the entry point that must not allocate: one array per GPU the library runs
// on, held until the process ends.  `scratch` is a host-side stand-in for the symbol, not a device address: the
// address on each device comes from hipGetSymbolAddress (once per device, in cpol_create) and only that pointer
// is ever handed to kernels, memsets or copies.
#define CPOL_STATIC_SCRATCH_BYTES ((size_t)80 << 20)
static __device__ __align__(256) unsigned char scratch[CPOL_STATIC_SCRATCH_BYTES];
#define CPOL_MAX_DEVICES 64
static void *g_scratch_ptr[CPOL_MAX_DEVICES];      // device address of `scratch` on each device, nullptr = not asked yet
static std::mutex g_scratch_mutex;

extern "C" int cpol_create(int device, cpol_ctx **out)
{
    if (!out) return CPOL_ERR_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n || device >= CPOL_MAX_DEVICES) return CPOL_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return CPOL_ERR_HIP;
    {
        // every GPU holds its own copy of the array at its own address
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        if (!g_scratch_ptr[device]) {
            void *p = nullptr;
            if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || !p) return CPOL_ERR_HIP;
            g_scratch_ptr[device] = p;
        }
    }
    cpol_ctx *ctx = new cpol_ctx();
    ctx->device = device;
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return CPOL_ERR_HIP; }
    memset(&ctx->cfg, 0, sizeof(ctx->cfg));
    *out = ctx;
    return CPOL_OK;
}

extern "C" int cpol_destroy(cpol_ctx *ctx)
{
    if (!ctx) return CPOL_OK;
    // teardown: nothing useful can be done about a failing free, so the statuses are dropped on purpose
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    for (auto *b : ctx->bufs) if (b->p) (void)hipFree(b->p);
    if (ctx->zcol) (void)hipFree(ctx->zcol);
    if (ctx->zc8) (void)hipFree(ctx->zc8);
    if (ctx->dcol) (void)hipFree(ctx->dcol);
    if (ctx->quad_dev) (void)hipFree(ctx->quad_dev);
    if (ctx->varray_dev) (void)hipFree(ctx->varray_dev);
    if (ctx->work_dev) (void)hipFree(ctx->work_dev);
    if (ctx->cert_dev) (void)hipFree(ctx->cert_dev);
    if (ctx->refr_keys) (void)hipFree(ctx->refr_keys);
    if (ctx->refr_tab) (void)hipFree(ctx->refr_tab);
    if (ctx->rawptr_dev) (void)hipFree((void *)ctx->rawptr_dev);
    for (auto &l : ctx->lut) {
        float *owned[] = {l.raw, l.rcs, l.table, l.table_comb, l.node_floor, l.xq, l.table_dop, l.xq_dop, l.d_axis};
        for (float *q : owned) if (q) (void)hipFree(q);
    }
    for (auto e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (auto e : ctx->copy_events) (void)hipEventDestroy(e);
    for (auto e : ctx->async_done) if (e) (void)hipEventDestroy(e);
    if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
    if (ctx->valid_host) (void)hipHostFree(ctx->valid_host);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return CPOL_OK;
}

extern "C" const char *cpol_last_error(cpol_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
extern "C" void *cpol_stream(cpol_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

extern "C" int cpol_synchronize(cpol_ctx *ctx)
{
    if (!ctx) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));      // deliveries of cpol_run_rays_async
    timing_collect(ctx);
    return CPOL_OK;
}

static int build_class_table(cpol_ctx *ctx, ClassTable &ct, bool allow_build);

// Build the lookup-table images as soon as the configuration and every table of its microphysics scheme are
// known, so that the hot-path calls find them ready (and the allocation-free one can rely on that).
static int prepare_images(cpol_ctx *ctx)
{
    if (!ctx->have_cfg) return CPOL_OK;
    const int n_cls = ctx->cfg.microphysics_scheme == 1 ? 3 : 4;
    for (int h = 0; h < n_cls; h++) if (!ctx->lut[h].present) return CPOL_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ClassTable ct;
    return build_class_table(ctx, ct, true);
}

extern "C" int cpol_set_config(cpol_ctx *ctx, const cpol_config *cfg)
{
    if (!ctx || !cfg) return CPOL_ERR_ARG;
    if (cfg->n_gates <= 0 || cfg->range_res <= 0) return fail(ctx, CPOL_ERR_ARG, "n_gates and range_res must be positive");
    if (cfg->microphysics_scheme != 1 && cfg->microphysics_scheme != 2) return fail(ctx, CPOL_ERR_ARG, "microphysics_scheme must be 1 or 2");
    if (cfg->sensitivity_mode < 0 || cfg->sensitivity_mode > 3) return fail(ctx, CPOL_ERR_ARG, "sensitivity_mode must be 0, 1, 2 or 3");
    if (cfg->doppler_scheme < 0 || cfg->doppler_scheme > 3)
        return fail(ctx, CPOL_ERR_ARG, "doppler_scheme must be 0 (off), 1, 2 or 3");
    if (cfg->doppler_scheme == 3 && cfg->microphysics_scheme != 1)
        return fail(ctx, CPOL_ERR_ARG, "doppler_scheme 3 needs 1-moment input: the reference's get_refl handles exactly three "
                                       "classes (doppler/doppler_c.c:18)");
    ctx->cfg = *cfg;
    if (ctx->cfg.earth_radius == 0) {
        // atm_refraction.py:18-21 evaluated at coords[1] taken as radians (:114-117)
        double lat = cfg->radar_lon, a = 6378.1370 * 1000, b = 6356.7523 * 1000;
        double n1 = a * a * cos(lat), n2 = b * b * sin(lat), d1 = a * cos(lat), d2 = b * sin(lat);
        ctx->cfg.earth_radius = sqrt((n1 * n1 + n2 * n2) / (d1 * d1 + d2 * d2));
    }
    ctx->have_cfg = true;
    return prepare_images(ctx);
}

extern "C" int cpol_upload_model(cpol_ctx *ctx, int nvars, const int *var_ids, const float *const *data,
                                 const float *zlev, int nz, int ny, int nx, const float llc[2],
                                 const float urc[2], const float res[2], const double pole[2])
{
    if (!ctx || !var_ids || !data || !zlev || !llc || !urc || !res || !pole) return CPOL_ERR_ARG;
    if (nvars < 1 || nvars > CPOL_V_COUNT || nz < 3 || ny < 2 || nx < 2) return fail(ctx, CPOL_ERR_ARG, "bad model shape");
    for (int i = 0; i < nvars; i++)
        if (var_ids[i] < 0 || var_ids[i] >= CPOL_V_COUNT || !data[i]) return fail(ctx, CPOL_ERR_ARG, "bad variable id");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    long ncell = (long)ny * nx;
    size_t nvox = (size_t)ncell * nz;
    if (ctx->zcol) { HIPCHK(ctx, hipFree(ctx->zcol)); ctx->zcol = nullptr; }
    if (ctx->zc8) { HIPCHK(ctx, hipFree(ctx->zc8)); ctx->zc8 = nullptr; }
    if (ctx->dcol) { HIPCHK(ctx, hipFree(ctx->dcol)); ctx->dcol = nullptr; }
    ctx->have_model = false;
    HIPCHK(ctx, hipMalloc((void **)&ctx->zcol, nvox * sizeof(float)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->zc8, (size_t)ncell * 8 * sizeof(float)));
    // a level's variables are padded to whole 16-byte words (k_interp reads them as such and keeps 32-bit word offsets)
    const int nvp = (nvars + 3) & ~3;
    if ((double)nvox * (nvp / 4) >= 2147483648.) return fail(ctx, CPOL_ERR_ARG, "model too large: nz*ny*nx*ceil(nvars/4) must stay below 2^31");
    HIPCHK(ctx, hipMalloc((void **)&ctx->dcol, nvox * nvp * sizeof(float)));
    if (nvp != nvars) HIPCHK(ctx, hipMemsetAsync(ctx->dcol, 0, nvox * nvp * sizeof(float), ctx->stream));
    TRY(ensure(ctx, ctx->w->b_stage, nvox * sizeof(float)));
    float *stage = (float *)ctx->w->b_stage.p;
    HIPCHK(ctx, hipMemcpyAsync(stage, zlev, nvox * sizeof(float), hipMemcpyDefault, ctx->stream));
    launch_transpose_levels(ctx->stream, stage, ctx->zcol, nz, ncell, 1, 0);
    launch_coarse_levels(ctx->stream, ctx->zcol, ctx->zc8, nz, ncell);
    for (int i = 0; i < nvars; i++) {
        HIPCHK(ctx, hipMemcpyAsync(stage, data[i], nvox * sizeof(float), hipMemcpyDefault, ctx->stream));
        launch_transpose_levels(ctx->stream, stage, ctx->dcol, nz, ncell, nvp, i);
        ctx->var_ids[i] = var_ids[i];
    }
    {   // does every division of the vertical interpolation stay in the range of the shared-reciprocal form?
        unsigned *flag = nullptr, bad = 1;
        HIPCHK(ctx, hipMalloc((void **)&flag, sizeof(unsigned)));
        HIPCHK(ctx, hipMemsetAsync(flag, 0, sizeof(unsigned), ctx->stream));
        launch_model_range(ctx->stream, ctx->dcol, nvox * nvp, ctx->zcol, nz, nvox, flag);
        const hipError_t e = hipMemcpyAsync(&bad, flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
        const hipError_t e2 = hipStreamSynchronize(ctx->stream);
        (void)hipFree(flag);
        HIPCHK(ctx, e);
        HIPCHK(ctx, e2);
        ctx->fast_div = bad == 0;
    }
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->nz = nz; ctx->ny = ny; ctx->nx = nx; ctx->nv = nvars;
    for (int k = 0; k < 2; k++) { ctx->llc[k] = llc[k]; ctx->urc[k] = urc[k]; ctx->res[k] = res[k]; ctx->pole[k] = pole[k]; }
    ctx->have_model = true;
    return CPOL_OK;
}

static void fill_psd(PsdView &v, const cpol_psd &p)
{
    v.kind = p.psd_kind; v.a = p.a; v.b = p.b; v.alpha = p.alpha; v.beta = p.beta; v.mu = p.mu; v.nu = p.nu;
    v.lambda_factor = p.lambda_factor; v.vel_factor = p.vel_factor; v.ntot_factor = p.ntot_factor;
    v.n0_fixed = p.n0_fixed; v.x_min = p.x_min; v.x_max = p.x_max; v.q_var = p.q_var; v.qn_var = p.qn_var;
    v.t_var = (p.t_var >= 0 && p.t_var < CPOL_V_COUNT) ? p.t_var : CPOL_V_T;
    v.n0_unit = pow(1000., -(1 + p.mu));
    v.lam_unit = pow(1000., -p.nu);
    v.log_lam_unit = log(v.lam_unit);
    v.c_exp = p.psd_kind == 0 ? 1. / (4. + p.mu) : p.psd_kind == 1 ? 1. / (p.b + 1) : -p.nu / p.b;
    v.c_n = p.nu / p.ntot_factor;
    v.c_n_exp = (p.mu + 1) / p.nu;
    v.log2_lf = log2(p.lambda_factor);
    v.l2_const = log2(1.4426950408889634) + (p.psd_kind == 2 ? log2(v.lam_unit) : 0.);
    v.cn_unit = v.c_n * v.n0_unit;
}

extern "C" int cpol_upload_lut(cpol_ctx *ctx, int cls, const cpol_lut *lut, const cpol_psd *psd,
                               const cpol_psd *psd_doppler)
{
    if (!ctx || !lut || !psd || cls < 0 || cls >= CPOL_MAX_CLASSES) return CPOL_ERR_ARG;
    if (lut->n_d != 1024) return fail(ctx, CPOL_ERR_ARG, "lookup tables must have 1024 diameters (compute_lut_sz.py:37)");
    if (lut->n_e < 1 || lut->n_t < 1 || !lut->table || !lut->d_axis) return fail(ctx, CPOL_ERR_ARG, "bad lookup table");
    if (psd->psd_kind < 0 || psd->psd_kind > 2 || psd->q_var < 0 || psd->q_var >= CPOL_V_COUNT ||
        (psd->psd_kind == 2 && (psd->qn_var < 0 || psd->qn_var >= CPOL_V_COUNT)))
        return fail(ctx, CPOL_ERR_ARG, "bad psd descriptor");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    LutDev &L = ctx->lut[cls];
    float **owned[] = {&L.raw, &L.rcs, &L.table, &L.table_comb, &L.node_floor, &L.xq, &L.table_dop, &L.xq_dop, &L.d_axis};
    for (float **q : owned) if (*q) { HIPCHK(ctx, hipFree(*q)); }
    L = LutDev();
    size_t nt = (size_t)lut->n_e * lut->n_t * lut->n_d * CPOL_N_SZ;
    HIPCHK(ctx, hipMalloc((void **)&L.raw, nt * sizeof(float)));
    HIPCHK(ctx, hipMalloc((void **)&L.d_axis, 1024 * sizeof(float)));
    HIPCHK(ctx, hipMemcpy(L.raw, lut->table, nt * sizeof(float), hipMemcpyDefault));
    std::vector<float> dax(1024);
    HIPCHK(ctx, hipMemcpy(dax.data(), lut->d_axis, 1024 * sizeof(float), hipMemcpyDefault));
    HIPCHK(ctx, hipMemcpy(L.d_axis, dax.data(), 1024 * sizeof(float), hipMemcpyHostToDevice));
    L.n_e = lut->n_e; L.n_t = lut->n_t; L.n_d = lut->n_d;
    L.e_min = lut->e_min; L.e_step = lut->e_step; L.t_min = lut->t_min; L.t_step = lut->t_step;
    L.dD = dax[1] - dax[0];
    // uniform axis (np.linspace, compute_lut_sz.py:104): D_d = D_0 + d * step to within float32 rounding
    {
        const double d0 = dax[0], st = ((double)dax[1023] - d0) / 1023.;
        double dev = 0;
        for (int d = 0; d < 1024; d++) dev = fmax(dev, fabs((double)dax[d] - (d0 + d * st)) / fabs((double)dax[d]));
        L.axis_uniform = dev < 2.5e-7;
        L.axis_step = st;
        L.d0 = dax[0];
    }
    L.d_host = dax;
    L.psd = *psd;
    L.psd_dop = psd_doppler ? *psd_doppler : *psd;
    L.present = true;
    return prepare_images(ctx);
}

extern "C" int cpol_set_turbulence(cpol_ctx *ctx, int enable, double beamwidth_deg, double kolmogorov_a)
{
    if (!ctx) return CPOL_ERR_ARG;
    if (enable && (!(beamwidth_deg > 0) || !(kolmogorov_a > 0)))
        return fail(ctx, CPOL_ERR_ARG, "turbulence broadening needs a positive beamwidth and Kolmogorov constant");
    ctx->turb = enable ? 1 : 0;
    ctx->turb_beamwidth = beamwidth_deg;
    ctx->turb_a = kolmogorov_a;
    return CPOL_OK;
}

extern "C" int cpol_set_velocity_bins(cpol_ctx *ctx, int n_vel, const double *varray)
{
    if (!ctx || n_vel < 2 || n_vel > 4096 || !varray) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (ctx->varray_dev) { HIPCHK(ctx, hipFree(ctx->varray_dev)); ctx->varray_dev = nullptr; }
    HIPCHK(ctx, hipMalloc((void **)&ctx->varray_dev, n_vel * sizeof(double)));
    HIPCHK(ctx, hipMemcpy(ctx->varray_dev, varray, n_vel * sizeof(double), hipMemcpyDefault));
    ctx->n_vel = n_vel;
    return CPOL_OK;
}

extern "C" int cpol_set_refraction_table(cpol_ctx *ctx, int n_profiles, const double *keys, const float *S,
                                         const float *H, const float *E)
{
    if (!ctx || n_profiles < 0) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->refr_keys) { HIPCHK(ctx, hipFree(ctx->refr_keys)); ctx->refr_keys = nullptr; }
    if (ctx->refr_tab) { HIPCHK(ctx, hipFree(ctx->refr_tab)); ctx->refr_tab = nullptr; }
    ctx->n_refr = 0;
    if (n_profiles == 0) return CPOL_OK;
    if (!ctx->have_cfg) return fail(ctx, CPOL_ERR_STATE, "cpol_set_config has not been called");
    if (!keys || !S || !H || !E) return fail(ctx, CPOL_ERR_ARG, "refraction table: NULL array");
    for (int i = 1; i < n_profiles; i++)
        if (!(keys[i] > keys[i - 1])) return fail(ctx, CPOL_ERR_ARG, "refraction table: keys must be strictly ascending");
    const size_t G = ctx->cfg.n_gates, n = (size_t)n_profiles * G;
    HIPCHK(ctx, hipMalloc((void **)&ctx->refr_keys, n_profiles * sizeof(double)));
    HIPCHK(ctx, hipMalloc((void **)&ctx->refr_tab, 3 * n * sizeof(float)));
    HIPCHK(ctx, hipMemcpy(ctx->refr_keys, keys, n_profiles * sizeof(double), hipMemcpyDefault));
    HIPCHK(ctx, hipMemcpy(ctx->refr_tab, S, n * sizeof(float), hipMemcpyDefault));
    HIPCHK(ctx, hipMemcpy(ctx->refr_tab + n, H, n * sizeof(float), hipMemcpyDefault));
    HIPCHK(ctx, hipMemcpy(ctx->refr_tab + 2 * n, E, n * sizeof(float), hipMemcpyDefault));
    ctx->n_refr = n_profiles;
    ctx->refr_gates = (int)G;
    return CPOL_OK;
}

extern "C" int cpol_set_quadrature(cpol_ctx *ctx, int n_sub, const double *az_off, const double *el_off,
                                   const double *weight)
{
    if (!ctx || n_sub < 1 || n_sub > 4096 || !az_off || !el_off || !weight) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ctx->az_off.assign(az_off, az_off + n_sub);
    ctx->el_off.assign(el_off, el_off + n_sub);
    ctx->weight.assign(weight, weight + n_sub);
    if (ctx->quad_dev) HIPCHK(ctx, hipFree(ctx->quad_dev));
    HIPCHK(ctx, hipMalloc((void **)&ctx->quad_dev, 3 * n_sub * sizeof(double)));
    std::vector<double> q(3 * n_sub);
    for (int i = 0; i < n_sub; i++) { q[i] = az_off[i]; q[n_sub + i] = el_off[i]; q[2 * n_sub + i] = weight[i]; }
    HIPCHK(ctx, hipMemcpy(ctx->quad_dev, q.data(), q.size() * sizeof(double), hipMemcpyHostToDevice));
    ctx->n_sub = n_sub;
    return CPOL_OK;
}

// ------------------------------------------------------------------------------------
static bool same_psd(const cpol_psd &a, const cpol_psd &b)
{
    return a.psd_kind == b.psd_kind && a.a == b.a && a.b == b.b && a.alpha == b.alpha && a.beta == b.beta &&
           a.mu == b.mu && a.nu == b.nu && a.lambda_factor == b.lambda_factor && a.n0_fixed == b.n0_fixed &&
           a.x_min == b.x_min && a.x_max == b.x_max && a.q_var == b.q_var && a.qn_var == b.qn_var &&
           (a.psd_kind != 1 || a.t_var == b.t_var || (a.t_var < 0 && b.t_var < 0));
}

// Reference point of the exponent argument: the smallest D^nu of the diameter axis (D_0^nu for the reference's
// ascending axes), so that exp(-lambda (D^nu - x_ref)) never exceeds 1.
static double xref_of(const LutDev &L, double nu)
{
    double m = pow((double)L.d0, nu);
    for (float d : L.d_host) { double x = pow((double)d, nu); if (x < m) m = x; }
    return m;
}

// Class table of the configured schemes; (re)builds the lookup-table images when the column
// layout they were built for differs from what the configuration needs.  With allow_build = false nothing is
// allocated or built: stale or missing images are an error (the entry point that must not allocate).
static int build_class_table(cpol_ctx *ctx, ClassTable &ct, bool allow_build)
{
    memset(&ct, 0, sizeof(ct));
    ct.n_cls = ctx->cfg.microphysics_scheme == 1 ? 3 : 4;
    const bool dop2 = ctx->cfg.doppler_scheme == 2;
    bool shared = dop2;
    for (int h = 0; h < ct.n_cls; h++) {
        if (!ctx->lut[h].present) return fail(ctx, CPOL_ERR_STATE, "a lookup table of the microphysics scheme is missing");
        shared = shared && same_psd(ctx->lut[h].psd, ctx->lut[h].psd_dop);
    }
    ct.dop_shared = shared ? 1 : 0;
    ct.ncol = shared ? 14 : 12;
    ct.n_kinds = (dop2 && !shared) ? 2 : 1;
    // the combined columns (ClassTable::comb); when the Doppler moments ride on the polarimetric units, the spare
    // column of the combined image carries the second one
    const bool want_comb = ctx->combine_columns;
    for (int h = 0; h < ct.n_cls; h++) {
        LutDev &L = ctx->lut[h];
        const long nslice = (long)L.n_e * L.n_t;
        const size_t img_bytes = (size_t)nslice * 1024 * ct.ncol * sizeof(float);
        const int want_dop_img = ct.n_kinds == 2 ? 1 : 0;
        const int tag = ct.ncol * 4 + (shared ? 2 : 0) + want_dop_img + (want_comb ? 256 : 0);
        if (L.image_ncol != tag) {
            if (!allow_build)
                return fail(ctx, CPOL_ERR_STATE, "the lookup-table images of this configuration have not been built yet: "
                                                 "call cpol_set_config and cpol_upload_lut (which build them) first");
            if (L.table) { HIPCHK(ctx, hipFree(L.table)); L.table = nullptr; }
            if (L.table_dop) { HIPCHK(ctx, hipFree(L.table_dop)); L.table_dop = nullptr; }
            if (L.table_comb) { HIPCHK(ctx, hipFree(L.table_comb)); L.table_comb = nullptr; }
            if (L.node_floor) { HIPCHK(ctx, hipFree(L.node_floor)); L.node_floor = nullptr; }
            if (!L.xq) HIPCHK(ctx, hipMalloc((void **)&L.xq, 1024 * sizeof(float)));
            if (!L.xq_dop) HIPCHK(ctx, hipMalloc((void **)&L.xq_dop, 1024 * sizeof(float)));
            HIPCHK(ctx, hipMalloc((void **)&L.table, img_bytes));
            launch_build_image(ctx->stream, L.raw, L.d_axis, L.table, L.xq, nslice, ct.ncol, L.psd.mu, L.psd.nu,
                               shared ? 1 : 0, 0, L.psd_dop.mu, L.psd_dop.alpha, L.psd_dop.beta, xref_of(L, L.psd.nu));
            if (want_dop_img) {
                HIPCHK(ctx, hipMalloc((void **)&L.table_dop, (size_t)nslice * CPOL_DOP_SLICE * sizeof(float)));
                launch_build_dop_image(ctx->stream, L.raw, L.d_axis, L.table_dop, L.xq_dop, nslice, L.psd_dop.mu,
                                       L.psd_dop.nu, L.psd_dop.alpha, L.psd_dop.beta, (double)(float)xref_of(L, L.psd_dop.nu));
            }
            if (!ctx->cert_dev) HIPCHK(ctx, hipMalloc((void **)&ctx->cert_dev, 4 * sizeof(float)));
            L.nodes_comb = 0;
            if (want_comb) {
                HIPCHK(ctx, hipMalloc((void **)&L.table_comb, (size_t)nslice * 1024 * CPOL_N_COMB * sizeof(float)));
                HIPCHK(ctx, hipMalloc((void **)&L.node_floor, (size_t)nslice * sizeof(float)));
                launch_build_image(ctx->stream, L.raw, L.d_axis, L.table_comb, L.xq, nslice, CPOL_N_COMB, L.psd.mu, L.psd.nu,
                                   shared ? 1 : 0, 0, L.psd_dop.mu, L.psd_dop.alpha, L.psd_dop.beta, xref_of(L, L.psd.nu));
                launch_node_floor(ctx->stream, L.raw, L.d_axis, nslice, L.psd.mu, L.node_floor);
                std::vector<float> fl((size_t)nslice);
                HIPCHK(ctx, hipMemcpyAsync(fl.data(), L.node_floor, fl.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                for (float f : fl) L.nodes_comb += f > 0.f ? 1 : 0;
            }
            HIPCHK(ctx, hipGetLastError());
            // term cutoff of the class: the smallest exponent its tables are certified for, else exact zeros only
            L.cut_e = 152.f;
            L.cut_e_dop = 152.f;
            {
                bool asc = L.d_host.size() == 1024 && L.d_host[0] > 0.f;
                for (int d = 1; asc && d < 1024; d++) asc = L.d_host[d] > L.d_host[d - 1];
                const double cand[] = {48., 56., 64., 72., 80., 96., 112., 128.};
                if (ctx->certified_cutoff && asc && L.psd.nu > 0.) {
                    for (double e : cand) {
                        const float worst = cut_certificate(ctx->stream, L.raw, L.d_axis, nslice, L.psd.mu, L.psd.nu,
                                                            xref_of(L, L.psd.nu), shared ? 1 : 0, L.psd_dop.mu, L.psd_dop.alpha,
                                                            L.psd_dop.beta, e, ctx->cert_dev);
                        if (worst <= 3.7252903e-09f) { L.cut_e = (float)(e + 2.); break; }      // 2^-28
                    }
                }
                // the separate Doppler units integrate the two moments under their own PSD: a certificate of their own
                if (ctx->certified_cutoff && asc && want_dop_img && L.psd_dop.nu > 0.) {
                    for (double e : cand) {
                        const float worst = cut_certificate(ctx->stream, L.raw, L.d_axis, nslice, L.psd_dop.mu, L.psd_dop.nu,
                                                            (double)(float)xref_of(L, L.psd_dop.nu), 2, L.psd_dop.mu, L.psd_dop.alpha,
                                                            L.psd_dop.beta, e, ctx->cert_dev);
                        if (worst <= 3.7252903e-09f) { L.cut_e_dop = (float)(e + 2.); break; }
                    }
                }
            }
            L.image_ncol = tag;
        }
        LutView &v = ct.lut[h];
        v.table = L.table; v.xq = L.xq; v.n_e = L.n_e; v.n_t = L.n_t; v.n_d = L.n_d;
        v.e_min = L.e_min; v.e_step = L.e_step; v.t_min = L.t_min; v.t_step = L.t_step; v.dD = L.dD;
        ct.lut_dop[h] = v;
        ct.lut_dop[h].table = L.table_dop; ct.lut_dop[h].xq = L.xq_dop;
        ct.lut_comb[h] = v;
        ct.lut_comb[h].table = L.table_comb; ct.lut_comb[h].floor = L.node_floor;
        // exp(-lambda D) over equally spaced D is a geometric sequence: k_integrate_dop then needs one
        // exponential per 64 diameters
        ct.pol_x0[h] = xref_of(L, L.psd.nu);
        ct.dop_x0[h] = (double)(float)xref_of(L, L.psd_dop.nu);
        // N(D) = 0 in float32 beyond a unit's cutoff needs D^nu ascending along the axis
        bool asc = L.d_host.size() == 1024 && L.d_host[0] > 0.f;
        for (int d = 1; asc && d < 1024; d++) asc = L.d_host[d] > L.d_host[d - 1];
        ct.cut_ok[h] = (asc && L.psd.nu > 0.) ? 1 : 0;
        ct.cut_e[h] = L.cut_e;
        ct.dop_cut_ok[h] = (asc && L.psd_dop.nu > 0.) ? 1 : 0;
        ct.dop_cut_e[h] = L.cut_e_dop;
        ct.pol_geo_step[h] = (L.axis_uniform && L.psd.nu == 1.0) ? (float)(4. * L.axis_step) : 0.f;
        ct.dop_geo_step[h] = (L.axis_uniform && L.psd_dop.nu == 1.0) ? (float)L.axis_step : 0.f;
        fill_psd(ct.psd[h], L.psd);
        fill_psd(ct.dop[h], L.psd_dop);
        if (L.n_e > ct.ne_max) ct.ne_max = L.n_e;
        if (L.n_t > ct.nt_max) ct.nt_max = L.n_t;
    }
    // two forms of polarimetric unit when any node qualifies for the combined columns
    bool any = false;
    for (int h = 0; h < ct.n_cls; h++) any = any || (want_comb && ctx->lut[h].table_comb && ctx->lut[h].nodes_comb > 0);
    // at most 32768 buckets (unit kind x form x class x elevation node x temperature node): their sort counters are
    // part of the workspace (17 MB at the limit), and a polarimetric bucket's number must fit cpol_key_t
    const long pol_nodes = (long)ct.n_cls * ct.ne_max * ct.nt_max;
    if ((long)ct.n_kinds * pol_nodes > 32768)
        return fail(ctx, CPOL_ERR_ARG, "more than 32768 lookup-table nodes (unit kind x class x elevation x temperature)");
    ct.comb = (any && (long)ct.n_kinds * 2 * pol_nodes <= 32768) ? 1 : 0;
    ct.n_pol_planes = ct.n_cls * (1 + ct.comb);
    return CPOL_OK;
}

static void build_obs_params(const cpol_ctx *ctx, int n_sub, int n_gates, int n_cls, ObsParams &op)
{
    const cpol_config &c = ctx->cfg;
    double wl = c.wavelength_mm;
    op.n_gates = n_gates; op.n_sub = n_sub; op.n_cls = n_cls;
    op.res = (float)c.range_res;
    op.res_km = (float)(c.range_res / 1000.);
    op.two_pi = (float)(2 * M_PI);
    op.zfac = (float)(wl * wl * wl * wl / (pow(M_PI, 5) * c.kw * c.kw));
    op.kdpfac = (float)(1e-3 * (180.0 / M_PI) * wl);
    op.two_wl = (float)(2 * wl);
    op.afac = (float)4.343e-3;
    op.att_corr = c.attenuation_correction;
    op.doppler = c.doppler_scheme;
    op.sens_mode = c.sensitivity_mode;
    op.sens0 = c.sensitivity[0]; op.sens1 = c.sensitivity[1];
    op.res_d = c.range_res;
}

static void build_model_view(const cpol_ctx *ctx, ModelView &mv)
{
    memset(&mv, 0, sizeof(mv));
    mv.zcol = ctx->zcol; mv.zc8 = ctx->zc8; mv.dcol = ctx->dcol;
    mv.nz = ctx->nz; mv.ny = ctx->ny; mv.nx = ctx->nx; mv.nv = ctx->nv; mv.nvp = (ctx->nv + 3) & ~3;
    mv.llc0 = ctx->llc[0]; mv.llc1 = ctx->llc[1]; mv.res0 = ctx->res[0]; mv.res1 = ctx->res[1];
    mv.urc0 = ctx->urc[0]; mv.urc1 = ctx->urc[1];
    mv.fast_div = ctx->fast_div ? 1 : 0;
    for (int v = 0; v < CPOL_V_COUNT; v++) mv.slot_of_var[v] = -1;
    for (int i = 0; i < ctx->nv; i++) { mv.var_of_slot[i] = ctx->var_ids[i]; mv.slot_of_var[ctx->var_ids[i]] = i; }
}

// Workspaces of run_physics for nu points.
static int ensure_physics(cpol_ctx *ctx, const ClassTable &ct, long nu)
{
    const int n_planes = ct.n_kinds * ct.n_cls;
    const long nslots = nu * n_planes;
    const int nb = ct.n_kinds * ct.n_pol_planes * ct.ne_max * ct.nt_max;
    TRY(ensure(ctx, ctx->w->b_key, nslots * sizeof(cpol_key_t)));
    TRY(ensure(ctx, ctx->w->b_lam, nslots * sizeof(float2)));
    TRY(ensure(ctx, ctx->w->b_coef, nslots * sizeof(double)));
    TRY(ensure(ctx, ctx->w->b_sorted, (size_t)nu * ct.n_cls * sizeof(int)));      // polarimetric units; their Doppler twins follow them
    TRY(ensure(ctx, ctx->w->b_urec, (size_t)nu * ct.n_cls * sizeof(float4)));          // the chunks' units in cutoff order
    TRY(ensure(ctx, ctx->w->b_rank, nslots * sizeof(int)));
    TRY(ensure(ctx, ctx->w->b_uflag, (size_t)nu));
    // result rows of the polarimetric units: 16 floats, or with combined columns 8 and, for the units that take all
    // twelve columns, 4 more beside them
    TRY(ensure(ctx, ctx->w->b_psd, (size_t)nu * ct.n_cls * (ct.comb ? CPOL_N_COMB : CPOL_N_COL) * sizeof(float)));
    if (ct.comb) TRY(ensure(ctx, ctx->w->b_psd_hi, (size_t)nu * ct.n_cls * (ct.dop_shared ? 8 : 4) * sizeof(float)));
    if (ct.n_kinds == 2) TRY(ensure(ctx, ctx->w->b_dop, (size_t)nu * ct.n_cls * sizeof(float2)));
    TRY(ensure(ctx, ctx->w->b_cbkt, (size_t)((nslots + CPOL_CHUNK_UNITS - 1) / CPOL_CHUNK_UNITS + nb + 1) * sizeof(int)));
    // counters, shard cursors, bucket offsets, chunk offsets, and the run totals of the two-level scan (launch_scan)
    TRY(ensure(ctx, ctx->w->b_hist, (size_t)((2 * CPOL_SORT_SHARDS + 2) * (nb + 1) + 2 * CPOL_SCAN_RUNS_MAX) * sizeof(int)));
    if (ctx->cfg.doppler_scheme) TRY(ensure(ctx, ctx->w->b_vr, nu * sizeof(double)));
    if (ctx->cfg.doppler_scheme == 2) TRY(ensure(ctx, ctx->w->b_n0d, nslots * sizeof(double)));
    return CPOL_OK;
}

// Clear the (bucket, shard) counters that the PSD set-up fills.
static int clear_hist(cpol_ctx *ctx, const ClassTable &ct)
{
    const int nb = ct.n_kinds * ct.n_pol_planes * ct.ne_max * ct.nt_max;
    HIPCHK(ctx, hipMemsetAsync(ctx->w->b_hist.p, 0, (size_t)CPOL_SORT_SHARDS * (nb + 1) * sizeof(int), ctx->stream));
    return CPOL_OK;
}

// PSD set-up -> bucket sort -> integration -> per-ray reduction for nr rays whose
// sub-beam profiles are already on the device.  prepped: the PSD set-up has run already (k_front, after
// ensure_physics and clear_hist), only Doppler scheme 1's radial velocity is still to be computed.
static int run_physics(cpol_ctx *ctx, const ClassTable &ct, int nr, int B, int G, const float *vals, long plane,
                       const float *elev, const signed char *mask, const double *wsub, const double *phi,
                       float *obs, long obs_plane, long ray0, double *mask_out, double *rvel_out, double *spectrum,
                       bool prepped = false)
{
    const long nu = (long)nr * B * G;
    const int nb = ct.n_kinds * ct.n_pol_planes * ct.ne_max * ct.nt_max;
    const int nb_pol = ct.n_pol_planes * ct.ne_max * ct.nt_max;       // the separate Doppler units' buckets follow
    TRY(ensure_physics(ctx, ct, nu));
    int *hist = (int *)ctx->w->b_hist.p, *cursor = hist + CPOL_SORT_SHARDS * (nb + 1), *offs = cursor + CPOL_SORT_SHARDS * (nb + 1),
        *chunk_off = offs + (nb + 1);
    {
    StageTimer t_(ctx, CPOL_ST_PREP_SORT);
    if (!prepped) {
        TRY(clear_hist(ctx, ct));
        launch_prep(ctx->stream, ct, nu, G, vals, plane, elev, const_cast<signed char *>(mask), wsub, B, phi, ctx->cfg.doppler_scheme == 3 ? 0 : ctx->cfg.doppler_scheme,
                    (cpol_key_t *)ctx->w->b_key.p, (int *)ctx->w->b_rank.p, (float2 *)ctx->w->b_lam.p, (double *)ctx->w->b_coef.p,
                    (double *)ctx->w->b_n0d.p, hist, (double *)ctx->w->b_vr.p);
    } else if (ctx->cfg.doppler_scheme == 1)
        launch_vr1(ctx->stream, ct, nu, G, vals, plane, elev, phi, (double *)ctx->w->b_vr.p);
    launch_scan(ctx->stream, nb, nb_pol, hist, offs, cursor, chunk_off, (int *)ctx->w->b_cbkt.p);
    // (keys below n_cls x nodes are units in the combined form when there are two forms)
    launch_scatter(ctx->stream, nu, ct.n_cls, (const cpol_key_t *)ctx->w->b_key.p, (const int *)ctx->w->b_rank.p, offs, cursor,
                   (int *)ctx->w->b_sorted.p, ct.comb ? ct.n_cls * ct.ne_max * ct.nt_max : 0, (unsigned char *)ctx->w->b_uflag.p);
    }
    const bool count_work = ctx->timing && !ctx->arena;
    if (count_work) {
        // number of valid units of this batch (offs[nb]) for the work counters
        if (ctx->valid_used == ctx->valid_cap) {
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            timing_collect(ctx);
            if (!ctx->valid_host) {
                HIPCHK(ctx, hipHostMalloc((void **)&ctx->valid_host, 3 * 4096 * sizeof(int), hipHostMallocDefault));
                ctx->valid_cap = 4096;
            }
        }
        int *vh = ctx->valid_host + 3 * ctx->valid_used++;
        HIPCHK(ctx, hipMemcpyAsync(vh, offs + nb_pol, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(vh + 1, offs + nb, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        // units in the combined form: the buckets of the planes [0, n_cls) when there are two forms
        vh[2] = 0;
        if (ct.comb) HIPCHK(ctx, hipMemcpyAsync(vh + 2, offs + ct.n_cls * ct.ne_max * ct.nt_max, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    }
    // upper bound of the chunk count of either kind (the exact count lives on the device)
    const long max_chunks = (nu * ct.n_cls + CPOL_CHUNK_UNITS - 1) / CPOL_CHUNK_UNITS + nb_pol;
    {
    StageTimer t_(ctx, CPOL_ST_INTEGRATE);
    launch_integrate(ctx->stream, ct, max_chunks, nb_pol, offs, chunk_off, (const int *)ctx->w->b_cbkt.p, (const int *)ctx->w->b_sorted.p,
                     (float4 *)ctx->w->b_urec.p, (const float2 *)ctx->w->b_lam.p, (float *)ctx->w->b_psd.p, (float *)ctx->w->b_psd_hi.p,
                     count_work ? ctx->work_dev : nullptr);
    }
    if (ct.n_kinds == 2) {
        StageTimer t_(ctx, CPOL_ST_INTEGRATE_DOP);
        launch_integrate_dop(ctx->stream, ct, max_chunks, nb_pol, nb, offs, chunk_off, (const int *)ctx->w->b_cbkt.p, (const int *)ctx->w->b_sorted.p,
                             (const float2 *)ctx->w->b_lam.p, (const double *)ctx->w->b_n0d.p, (float2 *)ctx->w->b_dop.p, (long)ct.n_cls * nu,
                             count_work ? ctx->work_dev : nullptr);
    }
    ObsParams op;
    build_obs_params(ctx, B, G, ct.n_cls, op);
    op.row = ct.comb ? CPOL_N_COMB : CPOL_N_COL;
    op.comb = ct.comb;
    op.nb_form = ct.n_cls * ct.ne_max * ct.nt_max;
    op.hi = ct.dop_shared ? 8 : 4;
    op.uflag = (const unsigned char *)ctx->w->b_uflag.p;
    Dop2In d2;
    memset(&d2, 0, sizeof(d2));
    if (ctx->cfg.doppler_scheme == 2) {
        d2.vals = vals; d2.plane = plane; d2.elev = elev; d2.phi_sc = phi;
        d2.lam = (const float2 *)ctx->w->b_lam.p; d2.n0d = (const double *)ctx->w->b_n0d.p;
        d2.dop_out = (const float2 *)ctx->w->b_dop.p;
        d2.n_cls = ct.n_cls; d2.ne_max = ct.ne_max; d2.nt_max = ct.nt_max; d2.shared = ct.dop_shared;
        d2.comb = ct.comb; d2.nb_form = op.nb_form; d2.psd_hi = (const float *)ctx->w->b_psd_hi.p;
        if (ct.dop_shared) {       // columns 12, 13 of the 14-column image; d = 1023 is (q, k, s) = (63, 3, 3)
            d2.slice_floats = 1024 * 14;
            d2.at0[0] = 12 * 4; d2.at0[1] = 13 * 4;
            d2.at1[0] = (((63 * 4 + 3) * 14 + 12) << 2) + 3; d2.at1[1] = (((63 * 4 + 3) * 14 + 13) << 2) + 3;
        } else {                   // the moment pairs of the compact image follow its 1024 exponents
            d2.slice_floats = CPOL_DOP_SLICE;
            d2.at0[0] = 1024; d2.at0[1] = 1025;
            d2.at1[0] = 1024 + 2 * 1023; d2.at1[1] = 1025 + 2 * 1023;
        }
        for (int h = 0; h < ct.n_cls; h++) {
            const LutView &v = ct.dop_shared ? ct.lut[h] : ct.lut_dop[h];
            d2.img[h] = v.table; d2.xq[h] = v.xq; d2.n_t[h] = v.n_t;
        }
    }
    if (ctx->cfg.doppler_scheme == 3) {
        StageTimer ts_(ctx, CPOL_ST_SPECTRUM);
        if (ctx->n_vel < 2) return fail(ctx, CPOL_ERR_STATE, "doppler_scheme 3: cpol_set_velocity_bins has not been called");
        // the spectrum needs the radar cross-section only: 4 KB per table node instead of the 48 KB of all twelve
        // components, built once per uploaded table
        for (int h = 0; h < 3; h++) {
            LutDev &L = ctx->lut[h];
            if (!L.rcs) {
                const long n = (long)L.n_e * L.n_t * 1024;
                HIPCHK(ctx, hipMalloc((void **)&L.rcs, n * sizeof(float)));
                launch_build_rcs(ctx->stream, L.raw, L.rcs, n);
            }
        }
        // device array of the cross-section table / diameter-axis pointers of the three classes
        const float *hp[2 * CPOL_MAX_CLASSES] = {nullptr};
        for (int h = 0; h < 3; h++) { hp[h] = ctx->lut[h].rcs; hp[CPOL_MAX_CLASSES + h] = ctx->lut[h].d_axis; }
        if (!ctx->rawptr_dev) HIPCHK(ctx, hipMalloc((void **)&ctx->rawptr_dev, sizeof(hp)));
        HIPCHK(ctx, hipMemcpyAsync((void *)ctx->rawptr_dev, hp, sizeof(hp), hipMemcpyHostToDevice, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));      // hp is a stack buffer
        SpecParams sp;
        memset(&sp, 0, sizeof(sp));
        sp.n_gates = G; sp.n_sub = B; sp.n_vel = ctx->n_vel; sp.zfac = op.zfac;
        sp.chain_cut = ctx->certified_cutoff ? 1 : 0;
        if (ctx->turb) {
            bool have_edr = false;
            for (int i = 0; i < ctx->nv; i++) have_edr = have_edr || ctx->var_ids[i] == CPOL_V_EDR;
            if (!have_edr && ctx->have_model)
                return fail(ctx, CPOL_ERR_STATE, "turbulence broadening: the model state has no eddy dissipation rate (CPOL_V_EDR)");
            if (ctx->n_vel > 2048) return fail(ctx, CPOL_ERR_ARG, "turbulence broadening handles at most 2048 velocity bins");
            sp.turb = 1;
            sp.sigma_r = ctx->cfg.range_res;
            sp.sigma_theta = ctx->turb_beamwidth * (M_PI / 180.0);
            sp.kolm_a = ctx->turb_a;
            sp.range_res = ctx->cfg.range_res;
        }
        for (int h = 0; h < 3; h++) {
            const cpol_psd &pd = ctx->lut[h].psd_dop;
            sp.d_min[h] = (float)pd.d_min; sp.d_max[h] = (float)pd.d_max; sp.step[h] = ctx->lut[h].dD;
            sp.inv_alpha[h] = 1.0 / pd.alpha; sp.inv_beta[h] = 1.0 / pd.beta;
            sp.mu[h] = (float)pd.mu; sp.nu[h] = (float)pd.nu;
            sp.geo_step[h] = (ctx->lut[h].axis_uniform && (float)pd.nu == 1.f) ? ctx->lut[h].axis_step : 0.;
        }
        launch_spectrum(ctx->stream, ct, sp, nr, nu, vals, plane, elev, mask, wsub, phi, ctx->varray_dev,
                        ctx->rawptr_dev, ctx->rawptr_dev + CPOL_MAX_CLASSES, spectrum, rvel_out, ray0);
    }
    StageTimer t_(ctx, CPOL_ST_OBSERVABLES);
    launch_observables(ctx->stream, op, nr, nu, (const cpol_key_t *)ctx->w->b_key.p, (const double *)ctx->w->b_coef.p,
                       (const float *)ctx->w->b_psd.p, (const float *)ctx->w->b_psd_hi.p, mask, wsub, (const double *)ctx->w->b_vr.p, obs, obs_plane, ray0,
                       mask_out, rvel_out, d2, spectrum, ctx->n_vel);
    HIPCHK(ctx, hipGetLastError());
    return CPOL_OK;
}

static int check_ready(cpol_ctx *ctx, bool need_model, bool need_quad)
{
    if (!ctx) return CPOL_ERR_ARG;
    if (!ctx->have_cfg) return fail(ctx, CPOL_ERR_STATE, "cpol_set_config has not been called");
    if (ctx->n_refr > 0 && ctx->refr_gates != ctx->cfg.n_gates)
        return fail(ctx, CPOL_ERR_STATE, "the refraction table was built for a different number of gates");
    if (need_model && !ctx->have_model) return fail(ctx, CPOL_ERR_STATE, "no model uploaded");
    if (need_quad && ctx->n_sub < 1) return fail(ctx, CPOL_ERR_STATE, "cpol_set_quadrature has not been called");
    return CPOL_OK;
}

static void geom_params(const cpol_ctx *ctx, GeomParams &gp)
{
    gp.lat0 = ctx->cfg.radar_lat; gp.lon0 = ctx->cfg.radar_lon; gp.alt = ctx->cfg.radar_alt;
    gp.res = ctx->cfg.range_res; gp.earth_radius = ctx->cfg.earth_radius;
    gp.pole_lat_s = ctx->pole[0]; gp.pole_lon_s = ctx->pole[1];
    gp.n_gates = ctx->cfg.n_gates; gp.n_sub = ctx->n_sub;
    gp.n_refr = ctx->n_refr; gp.refr_keys = ctx->refr_keys;
    const size_t n = (size_t)ctx->n_refr * ctx->refr_gates;
    gp.refr_S = ctx->refr_tab; gp.refr_H = ctx->refr_tab + n; gp.refr_E = ctx->refr_tab + 2 * n;
}

// points_too: the per-point arrays that the stand-alone geometry kernel writes (k_front keeps them in registers)
static int ensure_geom(cpol_ctx *ctx, long nrb, long nu, bool points_too = true)
{
    if (points_too) {
        TRY(ensure(ctx, ctx->w->b_S, nu * sizeof(float)));
        TRY(ensure(ctx, ctx->w->b_H, nu * sizeof(float)));
        TRY(ensure(ctx, ctx->w->b_lat, nu * sizeof(double)));
        TRY(ensure(ctx, ctx->w->b_lon, nu * sizeof(double)));
        TRY(ensure(ctx, ctx->w->b_coords, nu * 2 * sizeof(float)));
    }
    TRY(ensure(ctx, ctx->w->b_E, nu * sizeof(float)));
    TRY(ensure(ctx, ctx->w->b_line, (size_t)nrb * CPOL_LINE_N * sizeof(double)));
    TRY(ensure(ctx, ctx->w->b_phi, (size_t)nrb * 2 * sizeof(double)));
    return CPOL_OK;
}

// The hot path for n_rays rays.  no_alloc = false: the workspaces are the ctx's grow-only heap buffers and the
// rays go through in batches.  no_alloc = true (cpol_run_rays_static): ctx->arena is set, every workspace is a
// piece of the static scratch, azimuth / elevation are device arrays used in place, and nothing is allocated.
// Queue device-to-host copies of rays [r0, r0 + nr) on the copy stream, behind the event that marks the point of
// the compute stream at which they are final.  part 0: the central sub-beam's geometry (lat, lon, dist, height),
// which the front kernel of a batch writes; part 1: everything the last kernels of the batch write.
static int copy_batch_to_host(cpol_ctx *ctx, const cpol_outputs &d, const cpol_outputs &h, int n_rays, int r0, int nr,
                              hipEvent_t done, int part)
{
    const size_t G = ctx->cfg.n_gates, RG = (size_t)n_rays * G, o = (size_t)r0 * G, n = (size_t)nr * G;
    hipStream_t s = ctx->copy_stream;
    HIPCHK(ctx, hipStreamWaitEvent(s, done, 0));
    if (part == 0) {
        if (h.lat) HIPCHK(ctx, hipMemcpyAsync(h.lat + o, d.lat + o, n * sizeof(double), hipMemcpyDeviceToHost, s));
        if (h.lon) HIPCHK(ctx, hipMemcpyAsync(h.lon + o, d.lon + o, n * sizeof(double), hipMemcpyDeviceToHost, s));
        if (h.dist) HIPCHK(ctx, hipMemcpyAsync(h.dist + o, d.dist + o, n * sizeof(float), hipMemcpyDeviceToHost, s));
        if (h.height) HIPCHK(ctx, hipMemcpyAsync(h.height + o, d.height + o, n * sizeof(float), hipMemcpyDeviceToHost, s));
        return CPOL_OK;
    }
    // (one copy per field: a single hipMemcpy2DAsync of the nine rows measured 18 GB/s on rows of 7 MB where these
    // copies move 53 GB/s)
    if (h.obs)
        for (int f = 0; f < CPOL_N_OBS; f++)
            HIPCHK(ctx, hipMemcpyAsync(h.obs + f * RG + o, d.obs + f * RG + o, n * sizeof(float), hipMemcpyDeviceToHost, s));
    if (h.rvel && d.rvel) HIPCHK(ctx, hipMemcpyAsync(h.rvel + o, d.rvel + o, n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (h.dspectrum && d.dspectrum)
        HIPCHK(ctx, hipMemcpyAsync(h.dspectrum + o * ctx->n_vel, d.dspectrum + o * ctx->n_vel, n * ctx->n_vel * sizeof(double),
                                   hipMemcpyDeviceToHost, s));
    if (h.mask) HIPCHK(ctx, hipMemcpyAsync(h.mask + o, d.mask + o, n * sizeof(double), hipMemcpyDeviceToHost, s));
    if (h.ray_status) HIPCHK(ctx, hipMemcpyAsync(h.ray_status + r0, d.ray_status + r0, nr * sizeof(int), hipMemcpyDeviceToHost, s));
    if (h.diag && d.diag)
        for (int v = 0; v < ctx->nv; v++)
            HIPCHK(ctx, hipMemcpyAsync(h.diag + v * RG + o, d.diag + v * RG + o, n * sizeof(double), hipMemcpyDeviceToHost, s));
    return CPOL_OK;
}

// True when p lies in page-locked host memory (hipHostMalloc / hipHostRegister): copies into it are asynchronous.
static bool is_pinned_host(const void *p)
{
    if (!p) return false;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeHost;
}

// host (may be NULL): host buffers that receive the outputs batch by batch, the copy of one batch on a second
// stream beside the kernels of the next.
static int run_rays_impl(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                         cpol_outputs *out, bool no_alloc, const cpol_outputs *host = nullptr, bool async_delivery = false)
{
    ClassTable ct;
    TRY(build_class_table(ctx, ct, !no_alloc));
    // every variable the scheme reads must have been uploaded
    {
        bool have[CPOL_V_COUNT] = {false};
        for (int i = 0; i < ctx->nv; i++) have[ctx->var_ids[i]] = true;
        bool ok = have[CPOL_V_T];
        for (int h = 0; h < ct.n_cls; h++) {
            ok = ok && have[ct.psd[h].q_var] && (ct.psd[h].qn_var < 0 || have[ct.psd[h].qn_var]) && have[ct.psd[h].t_var];
            if (ctx->cfg.doppler_scheme)
                ok = ok && have[ct.dop[h].q_var] && (ct.dop[h].qn_var < 0 || have[ct.dop[h].qn_var]) && have[ct.dop[h].t_var];
        }
        if (ctx->cfg.doppler_scheme) ok = ok && have[CPOL_V_U] && have[CPOL_V_V] && have[CPOL_V_W] && have[CPOL_V_RHO];
        if (!ok) return fail(ctx, CPOL_ERR_STATE, "the uploaded model lacks a variable the configured schemes read");
    }
    const int B = ctx->n_sub, G = ctx->cfg.n_gates;
    const long RG = (long)n_rays * G;
    cpol_ctx::Work &w = *ctx->w;
    if (!out->obs) { TRY(ensure(ctx, w.o_obs, (size_t)CPOL_N_OBS * RG * sizeof(float))); out->obs = (float *)w.o_obs.p; }
    if (!out->rvel) { TRY(ensure(ctx, w.o_rvel, RG * sizeof(double))); out->rvel = (double *)w.o_rvel.p; }
    if (!out->mask) { TRY(ensure(ctx, w.o_mask, RG * sizeof(double))); out->mask = (double *)w.o_mask.p; }
    if (!out->lat) { TRY(ensure(ctx, w.o_lat, RG * sizeof(double))); out->lat = (double *)w.o_lat.p; }
    if (!out->lon) { TRY(ensure(ctx, w.o_lon, RG * sizeof(double))); out->lon = (double *)w.o_lon.p; }
    if (!out->dist) { TRY(ensure(ctx, w.o_dist, RG * sizeof(float))); out->dist = (float *)w.o_dist.p; }
    if (!out->height) { TRY(ensure(ctx, w.o_height, RG * sizeof(float))); out->height = (float *)w.o_height.p; }
    if (!out->ray_status) { TRY(ensure(ctx, w.b_status, n_rays * sizeof(int))); out->ray_status = (int *)w.b_status.p; }
    if (ctx->cfg.doppler_scheme == 3) {
        if (ctx->n_vel < 2) return fail(ctx, CPOL_ERR_STATE, "doppler_scheme 3: cpol_set_velocity_bins has not been called");
        if (!out->dspectrum) { TRY(ensure(ctx, w.o_spec, (size_t)RG * ctx->n_vel * sizeof(double))); out->dspectrum = (double *)w.o_spec.p; }
    } else out->dspectrum = nullptr;
    // diagnostic mode: wanted when the caller (or the host caller behind a ctx-owned device buffer) asks for it
    const bool want_diag = out->diag || (host && host->diag);
    if (want_diag && !out->diag) { TRY(ensure(ctx, w.o_diag, (size_t)ctx->nv * RG * sizeof(double))); out->diag = (double *)w.o_diag.p; }
    DiagVars dv;
    memset(&dv, 0, sizeof(dv));
    double diag_sum_w = 0.;
    if (want_diag) {
        dv.n = ctx->nv;
        for (int i = 0; i < ctx->nv; i++) dv.var[i] = ctx->var_ids[i];
        for (double wq : ctx->weight) diag_sum_w += wq;              // sum_weights += GH_weight, in sub-beam order (interpolation.py:29-31)
    }
    const double *az_dev = azimuth, *el_dev = elevation;
    if (!no_alloc) {
        // host or device arrays: a copy of either kind puts them on the device
        TRY(ensure(ctx, w.b_az, n_rays * sizeof(double)));
        TRY(ensure(ctx, w.b_el, n_rays * sizeof(double)));
        HIPCHK(ctx, hipMemcpyAsync(w.b_az.p, azimuth, n_rays * sizeof(double), hipMemcpyDefault, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(w.b_el.p, elevation, n_rays * sizeof(double), hipMemcpyDefault, ctx->stream));
        az_dev = (const double *)w.b_az.p; el_dev = (const double *)w.b_el.p;
    }
    int *status = out->ray_status;
    HIPCHK(ctx, hipMemsetAsync(status, 0, n_rays * sizeof(int), ctx->stream));

    long per_ray = (long)B * G;
    int rays_per_batch = (int)(ctx->max_batch_units / per_ray);
    if (rays_per_batch < 1) rays_per_batch = 1;
    if (rays_per_batch > n_rays || no_alloc) rays_per_batch = n_rays;
    // Batch schedule.  Device outputs: as few batches as the workspace limit allows.  Host delivery: the outputs
    // of a batch leave on the copy stream beside the kernels of the next, so only the last batch's copy is not
    // hidden: host_batches batches, the last one a small one (host_last_fraction of the rays, but about 1 Mi
    // positions at least: a batch has fixed costs), the others equal.
    std::vector<int> sched;
    {
        int nb_host = 1, last = 0;
        if (host && ctx->host_batches > 1) {
            const int min_rays = (int)std::max<long>(1, ((1L << 20) + per_ray - 1) / per_ray);
            last = std::max(min_rays, (int)(n_rays * ctx->host_last_fraction));
            nb_host = ctx->host_batches;
            if (last * nb_host > n_rays) { nb_host = 1; last = 0; }      // too small a call to split
        }
        const int body = n_rays - last;
        int per = nb_host > 1 ? (body + nb_host - 2) / (nb_host - 1) : body;
        if (per > rays_per_batch) per = rays_per_batch;
        for (int left = body; left > 0; left -= per) sched.push_back(left < per ? left : per);
        if (last) sched.push_back(last);
    }
    long nu_max = 0;
    for (int nr : sched) nu_max = std::max(nu_max, (long)nr * per_ray);
    const int nr_max = (int)(nu_max / per_ray);
    TRY(ensure_geom(ctx, (long)nr_max * B, nu_max, false));
    TRY(ensure(ctx, w.b_vals, (size_t)CPOL_V_COUNT * nu_max * sizeof(float)));
    TRY(ensure(ctx, w.b_mask, nu_max));
    GeomParams gp;
    geom_params(ctx, gp);
    ModelView mv;
    build_model_view(ctx, mv);
    const double *wsub = ctx->quad_dev + 2 * B;
    // variables that Doppler scheme 1's k_vr1 reads back
    unsigned vr1_mask = 1u << CPOL_V_RHO | 1u << CPOL_V_U | 1u << CPOL_V_V | 1u << CPOL_V_W;
    for (int h = 0; h < ct.n_cls && ctx->cfg.doppler_scheme == 1; h++) {
        vr1_mask |= 1u << ct.dop[h].q_var | 1u << ct.dop[h].t_var;
        if (ct.dop[h].qn_var >= 0) vr1_mask |= 1u << ct.dop[h].qn_var;
    }
    // What reaches the host: rvel only with a Doppler scheme (without one the device buffer is never written).
    cpol_outputs hd;
    memset(&hd, 0, sizeof(hd));
    bool early_geometry = false;
    if (host) {
        hd = *host;
        if (!ctx->cfg.doppler_scheme) hd.rvel = nullptr;
        if (!ctx->copy_stream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        // a copy into pageable memory blocks the host until it is done: such copies are queued only once the GPU
        // has its next batch; page-locked buffers take the geometry as soon as the front kernel has written it
        early_geometry = is_pinned_host(hd.lat) && is_pinned_host(hd.lon) && is_pinned_host(hd.dist) && is_pinned_host(hd.height)
                         && is_pinned_host(hd.obs);
        while (ctx->copy_events.size() < 2 * sched.size()) {
            hipEvent_t e;
            HIPCHK(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->copy_events.push_back(e);
        }
    }
    int prev_r0 = -1, prev_nr = 0, n_batch = 0;
    hipEvent_t prev_done = nullptr;
    int rc = CPOL_OK;
    auto batches = [&]() -> int {
        int r0 = 0;
        for (int nr : sched) {
            long nu = (long)nr * per_ray;
            {
                // geometry of the points, interpolation and PSD set-up in one kernel; the variables that a later
                // kernel reads are the only ones that reach memory
                StageTimer t_(ctx, CPOL_ST_INTERP);
                TRY(ensure_physics(ctx, ct, nu));
                TRY(clear_hist(ctx, ct));
                FrontOut fo;
                memset(&fo, 0, sizeof(fo));
                fo.E = (float *)w.b_E.p; fo.mask = (signed char *)w.b_mask.p;
                fo.vals = (float *)w.b_vals.p; fo.plane = nu_max;
                const int ds = ctx->cfg.doppler_scheme;
                fo.write_mask = (ds == 3 || want_diag) ? ~0u
                              : ds == 2 ? (1u << CPOL_V_RHO | 1u << CPOL_V_U | 1u << CPOL_V_V | 1u << CPOL_V_W)
                              : ds == 1 ? vr1_mask : 0u;
                fo.dist = out->dist; fo.height = out->height; fo.lat = out->lat; fo.lon = out->lon;
                fo.ray0 = r0; fo.status = status + r0;
                fo.key = (cpol_key_t *)w.b_key.p; fo.rank = (int *)w.b_rank.p; fo.lam = (float2 *)w.b_lam.p;
                fo.coef = (double *)w.b_coef.p; fo.n0d = (double *)w.b_n0d.p; fo.hist = (int *)w.b_hist.p;
                launch_front(ctx->stream, gp, mv, ct, nr, az_dev + r0, el_dev + r0, ctx->quad_dev, (double *)w.b_line.p,
                             (double *)w.b_phi.p, wsub, ds == 3 ? 0 : ds, fo);
            }
            if (host && early_geometry) {
                // the previous batch's results first (their event comes first on the compute stream), then this
                // batch's geometry
                if (prev_done) { TRY(copy_batch_to_host(ctx, *out, hd, n_rays, prev_r0, prev_nr, prev_done, 1)); prev_done = nullptr; }
                hipEvent_t fr = ctx->copy_events[2 * n_batch];
                HIPCHK(ctx, hipEventRecord(fr, ctx->stream));
                TRY(copy_batch_to_host(ctx, *out, hd, n_rays, r0, nr, fr, 0));
            }
            TRY(run_physics(ctx, ct, nr, B, G, (const float *)w.b_vals.p, nu_max, (const float *)w.b_E.p,
                            (const signed char *)w.b_mask.p, wsub, (const double *)w.b_phi.p,
                            out->obs, RG, r0, out->mask, out->rvel, out->dspectrum, true));
            if (want_diag)
                launch_diag(ctx->stream, nr, B, G, dv, (const float *)w.b_vals.p, nu_max, wsub, diag_sum_w, out->diag, RG, r0);
            if (host) {
                hipEvent_t done = ctx->copy_events[2 * n_batch + 1];
                HIPCHK(ctx, hipEventRecord(done, ctx->stream));
                if (prev_done) {       // pageable host buffers: this batch's kernels are queued, now the previous batch may block the host
                    TRY(copy_batch_to_host(ctx, *out, hd, n_rays, prev_r0, prev_nr, prev_done, 0));
                    TRY(copy_batch_to_host(ctx, *out, hd, n_rays, prev_r0, prev_nr, prev_done, 1));
                }
                prev_done = done; prev_r0 = r0; prev_nr = nr;
            }
            n_batch++;
            r0 += nr;
        }
        HIPCHK(ctx, hipGetLastError());
        if (host && prev_done) {
            if (!early_geometry) TRY(copy_batch_to_host(ctx, *out, hd, n_rays, prev_r0, prev_nr, prev_done, 0));
            TRY(copy_batch_to_host(ctx, *out, hd, n_rays, prev_r0, prev_nr, prev_done, 1));
        }
        return CPOL_OK;
    };
    rc = batches();
    if (!ctx->cfg.doppler_scheme) out->rvel = nullptr;
    if (host && ctx->copy_stream && !(async_delivery && rc == CPOL_OK)) {
        // also on failure: copies already queued write into the caller's buffers, which must not outlive this call
        const hipError_t e = hipStreamSynchronize(ctx->copy_stream);
        if (rc == CPOL_OK) HIPCHK(ctx, e);
    }
    return rc;
}

extern "C" int cpol_run_rays_dev(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                                 cpol_outputs *out)
{
    TRY(check_ready(ctx, true, true));
    if (n_rays == 0) return CPOL_OK;              // an empty scan is not an error
    if (n_rays < 0 || !azimuth || !elevation || !out) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return run_rays_impl(ctx, n_rays, azimuth, elevation, out, false);
}

extern "C" int cpol_run_rays_static(cpol_ctx *ctx, int n_rays, const double *azimuth_dev, const double *elevation_dev,
                                    cpol_outputs *out)
{
    TRY(check_ready(ctx, true, true));
    if (n_rays == 0) return CPOL_OK;
    if (n_rays < 0 || !azimuth_dev || !elevation_dev || !out) return CPOL_ERR_ARG;
    if (ctx->cfg.doppler_scheme == 3)
        return fail(ctx, CPOL_ERR_ARG, "cpol_run_rays_static does not serve doppler_scheme 3 (its tables are built on first use)");
    const long positions = (long)n_rays * ctx->n_sub * ctx->cfg.n_gates;
    if (positions > CPOL_STATIC_MAX_POSITIONS)
        return fail(ctx, CPOL_ERR_ARG, "cpol_run_rays_static: n_rays x sub-beams x gates exceeds CPOL_STATIC_MAX_POSITIONS");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void *base = g_scratch_ptr[ctx->device];      // set by cpol_create on this device
    if (!base) return fail(ctx, CPOL_ERR_STATE, "the static scratch array has no address on this device");
    // every workspace of this call is a piece of the static array; the heap buffers are left as they are
    ctx->arena_work = cpol_ctx::Work();
    ctx->w = &ctx->arena_work;
    ctx->arena = (char *)base; ctx->arena_cap = CPOL_STATIC_SCRATCH_BYTES; ctx->arena_off = 0;
    const int rc = run_rays_impl(ctx, n_rays, azimuth_dev, elevation_dev, out, true);
    ctx->arena = nullptr; ctx->arena_cap = 0;
    ctx->w = &ctx->heap;
    return rc;
}

extern "C" int cpol_run_rays(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                             cpol_outputs *host)
{
    if (!host) return CPOL_ERR_ARG;
    TRY(check_ready(ctx, true, true));
    if (n_rays == 0) return CPOL_OK;
    if (n_rays < 0 || !azimuth || !elevation) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the outputs are produced in ctx-owned device buffers and leave for the host batch by batch
    cpol_outputs d;
    memset(&d, 0, sizeof(d));
    TRY(run_rays_impl(ctx, n_rays, azimuth, elevation, &d, false, host));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    timing_collect(ctx);
    return CPOL_OK;
}

extern "C" int cpol_run_rays_async(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                                   cpol_outputs *host)
{
    if (!host) return CPOL_ERR_ARG;
    TRY(check_ready(ctx, true, true));
    if (n_rays == 0) return CPOL_OK;
    if (n_rays < 0 || !azimuth || !elevation) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // the device outputs of this call: the set that the call before the last one used
    const int slot = ctx->async_slot;
    ctx->async_slot ^= 1;
    cpol_ctx::Buf *b = ctx->async_out[slot];
    const size_t RG = (size_t)n_rays * ctx->cfg.n_gates;
    cpol_outputs d;
    memset(&d, 0, sizeof(d));
    TRY(ensure(ctx, b[0], CPOL_N_OBS * RG * sizeof(float)));  d.obs = (float *)b[0].p;
    TRY(ensure(ctx, b[1], RG * sizeof(double)));              d.rvel = (double *)b[1].p;
    TRY(ensure(ctx, b[2], RG * sizeof(double)));              d.mask = (double *)b[2].p;
    TRY(ensure(ctx, b[3], RG * sizeof(double)));              d.lat = (double *)b[3].p;
    TRY(ensure(ctx, b[4], RG * sizeof(double)));              d.lon = (double *)b[4].p;
    TRY(ensure(ctx, b[5], RG * sizeof(float)));               d.dist = (float *)b[5].p;
    TRY(ensure(ctx, b[6], RG * sizeof(float)));               d.height = (float *)b[6].p;
    TRY(ensure(ctx, b[7], (size_t)n_rays * sizeof(int)));     d.ray_status = (int *)b[7].p;
    if (ctx->cfg.doppler_scheme == 3 && ctx->n_vel >= 2) {
        TRY(ensure(ctx, b[8], RG * ctx->n_vel * sizeof(double)));
        d.dspectrum = (double *)b[8].p;
    }
    if (host->diag) { TRY(ensure(ctx, b[9], (size_t)ctx->nv * RG * sizeof(double))); d.diag = (double *)b[9].p; }
    // what the set held before must have left for the host before this call's kernels write into it
    if (ctx->async_done[slot]) HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, ctx->async_done[slot], 0));
    TRY(run_rays_impl(ctx, n_rays, azimuth, elevation, &d, false, host, true));
    if (!ctx->async_done[slot]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->async_done[slot], hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->async_done[slot], ctx->copy_stream));
    return CPOL_OK;
}

extern "C" int cpol_wait_async(cpol_ctx *ctx, int calls_back)
{
    if (!ctx || calls_back < 0 || calls_back > 1) return CPOL_ERR_ARG;
    // the most recent call used the set that is not next in turn
    hipEvent_t e = ctx->async_done[(ctx->async_slot ^ 1) ^ calls_back];
    if (e) HIPCHK(ctx, hipEventSynchronize(e));
    return CPOL_OK;
}

extern "C" int cpol_geolocate(cpol_ctx *ctx, int n_rays, const double *azimuth, const double *elevation,
                              float *dist, float *height, float *elev, double *lat, double *lon,
                              float *coords_rot, int *ray_status)
{
    TRY(check_ready(ctx, true, true));
    if (n_rays < 1 || !azimuth || !elevation) return CPOL_ERR_ARG;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int B = ctx->n_sub, G = ctx->cfg.n_gates;
    const long nu = (long)n_rays * B * G;
    TRY(ensure(ctx, ctx->w->b_az, n_rays * sizeof(double)));
    TRY(ensure(ctx, ctx->w->b_el, n_rays * sizeof(double)));
    TRY(ensure(ctx, ctx->w->b_status, n_rays * sizeof(int)));
    TRY(ensure_geom(ctx, (long)n_rays * B, nu));
    hipStream_t s = ctx->stream;
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->b_az.p, azimuth, n_rays * sizeof(double), hipMemcpyDefault, s));
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->b_el.p, elevation, n_rays * sizeof(double), hipMemcpyDefault, s));
    HIPCHK(ctx, hipMemsetAsync(ctx->w->b_status.p, 0, n_rays * sizeof(int), s));
    GeomParams gp;
    geom_params(ctx, gp);
    ModelView mv;
    build_model_view(ctx, mv);
    launch_geom(s, gp, n_rays, (const double *)ctx->w->b_az.p, (const double *)ctx->w->b_el.p, ctx->quad_dev,
                (double *)ctx->w->b_line.p, (float *)ctx->w->b_S.p, (float *)ctx->w->b_H.p, (float *)ctx->w->b_E.p,
                (double *)ctx->w->b_lat.p, (double *)ctx->w->b_lon.p, (float *)ctx->w->b_coords.p, (double *)ctx->w->b_phi.p, mv,
                (int *)ctx->w->b_status.p);
    HIPCHK(ctx, hipGetLastError());
    if (dist) HIPCHK(ctx, hipMemcpyAsync(dist, ctx->w->b_S.p, nu * sizeof(float), hipMemcpyDeviceToHost, s));
    if (height) HIPCHK(ctx, hipMemcpyAsync(height, ctx->w->b_H.p, nu * sizeof(float), hipMemcpyDeviceToHost, s));
    if (elev) HIPCHK(ctx, hipMemcpyAsync(elev, ctx->w->b_E.p, nu * sizeof(float), hipMemcpyDeviceToHost, s));
    if (lat) HIPCHK(ctx, hipMemcpyAsync(lat, ctx->w->b_lat.p, nu * sizeof(double), hipMemcpyDeviceToHost, s));
    if (lon) HIPCHK(ctx, hipMemcpyAsync(lon, ctx->w->b_lon.p, nu * sizeof(double), hipMemcpyDeviceToHost, s));
    if (coords_rot) HIPCHK(ctx, hipMemcpyAsync(coords_rot, ctx->w->b_coords.p, nu * 2 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (ray_status) HIPCHK(ctx, hipMemcpyAsync(ray_status, ctx->w->b_status.p, n_rays * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return CPOL_OK;
}

extern "C" int cpol_interp_points(cpol_ctx *ctx, int n, const float *coords_rot, const float *heights, float *out)
{
    if (!ctx || n < 1 || !coords_rot || !heights || !out) return CPOL_ERR_ARG;
    if (!ctx->have_model) return fail(ctx, CPOL_ERR_STATE, "no model uploaded");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    TRY(ensure(ctx, ctx->w->t_a, (size_t)n * 2 * sizeof(float)));
    TRY(ensure(ctx, ctx->w->t_b, (size_t)n * sizeof(float)));
    TRY(ensure(ctx, ctx->w->t_c, (size_t)n * ctx->nv * sizeof(float)));
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->t_a.p, coords_rot, (size_t)n * 2 * sizeof(float), hipMemcpyDefault, s));
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->t_b.p, heights, (size_t)n * sizeof(float), hipMemcpyDefault, s));
    ModelView mv;
    build_model_view(ctx, mv);
    launch_interp(s, mv, n, (const float *)ctx->w->t_a.p, (const float *)ctx->w->t_b.p, nullptr, 0, nullptr, (float *)ctx->w->t_c.p);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(out, ctx->w->t_c.p, (size_t)n * ctx->nv * sizeof(float), hipMemcpyDefault, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return CPOL_OK;
}

extern "C" int cpol_observables_from_beams(cpol_ctx *ctx, int n_sub, int n_gates, const float *values,
                                           const float *elev, const double *mask, const double *weight,
                                           const double *az_sub, float *obs, double *mask_out, double *rvel,
                                           double *dspectrum)
{
    TRY(check_ready(ctx, false, false));
    if (n_sub < 1 || n_gates < 1 || !values || !elev || !mask || !weight) return CPOL_ERR_ARG;
    if (ctx->cfg.doppler_scheme && !az_sub) return fail(ctx, CPOL_ERR_ARG, "az_sub is required with a Doppler scheme");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    ClassTable ct;
    TRY(build_class_table(ctx, ct, true));
    hipStream_t s = ctx->stream;
    const long nu = (long)n_sub * n_gates;
    TRY(ensure(ctx, ctx->w->b_vals, (size_t)CPOL_V_COUNT * nu * sizeof(float)));
    TRY(ensure(ctx, ctx->w->b_E, nu * sizeof(float)));
    TRY(ensure(ctx, ctx->w->b_mask, nu));
    TRY(ensure(ctx, ctx->w->t_a, n_sub * sizeof(double)));
    TRY(ensure(ctx, ctx->w->t_b, n_sub * sizeof(double)));
    TRY(ensure(ctx, ctx->w->t_c, (size_t)CPOL_N_OBS * n_gates * sizeof(float)));
    TRY(ensure(ctx, ctx->w->t_d, n_gates * sizeof(double)));
    TRY(ensure(ctx, ctx->w->t_e, n_gates * sizeof(double)));
    if (ctx->cfg.doppler_scheme == 3) TRY(ensure(ctx, ctx->w->o_spec, (size_t)n_gates * (ctx->n_vel > 0 ? ctx->n_vel : 1) * sizeof(double)));
    std::vector<signed char> m8(nu);
    for (long i = 0; i < nu; i++) m8[i] = (signed char)(mask[i] > 0 ? 1 : (mask[i] < 0 ? -1 : 0));
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->b_vals.p, values, (size_t)CPOL_V_COUNT * nu * sizeof(float), hipMemcpyDefault, s));
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->b_E.p, elev, nu * sizeof(float), hipMemcpyDefault, s));
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->b_mask.p, m8.data(), nu, hipMemcpyHostToDevice, s));
    HIPCHK(ctx, hipMemcpyAsync(ctx->w->t_a.p, weight, n_sub * sizeof(double), hipMemcpyDefault, s));
    TRY(ensure(ctx, ctx->w->b_phi, (size_t)n_sub * 2 * sizeof(double)));
    if (az_sub) {
        HIPCHK(ctx, hipMemcpyAsync(ctx->w->t_b.p, az_sub, n_sub * sizeof(double), hipMemcpyDefault, s));
        launch_sincos_deg(s, n_sub, (const double *)ctx->w->t_b.p, (double *)ctx->w->b_phi.p);
    }
    HIPCHK(ctx, hipStreamSynchronize(s));      // m8 is a local buffer
    int save_sens = ctx->cfg.sensitivity_mode;
    ctx->cfg.sensitivity_mode = 0;             // get_radar_observables itself applies no cut
    int rc = run_physics(ctx, ct, 1, n_sub, n_gates, (const float *)ctx->w->b_vals.p, nu, (const float *)ctx->w->b_E.p,
                         (const signed char *)ctx->w->b_mask.p, (const double *)ctx->w->t_a.p, (const double *)ctx->w->b_phi.p,
                         (float *)ctx->w->t_c.p, n_gates, 0, (double *)ctx->w->t_d.p, (double *)ctx->w->t_e.p,
                         ctx->cfg.doppler_scheme == 3 ? (double *)ctx->w->o_spec.p : nullptr);
    ctx->cfg.sensitivity_mode = save_sens;
    TRY(rc);
    if (obs) HIPCHK(ctx, hipMemcpyAsync(obs, ctx->w->t_c.p, (size_t)CPOL_N_OBS * n_gates * sizeof(float), hipMemcpyDefault, s));
    if (mask_out) HIPCHK(ctx, hipMemcpyAsync(mask_out, ctx->w->t_d.p, n_gates * sizeof(double), hipMemcpyDefault, s));
    if (rvel && ctx->cfg.doppler_scheme) HIPCHK(ctx, hipMemcpyAsync(rvel, ctx->w->t_e.p, n_gates * sizeof(double), hipMemcpyDefault, s));
    if (dspectrum && ctx->cfg.doppler_scheme == 3)
        HIPCHK(ctx, hipMemcpyAsync(dspectrum, ctx->w->o_spec.p, (size_t)n_gates * ctx->n_vel * sizeof(double), hipMemcpyDefault, s));
    HIPCHK(ctx, hipStreamSynchronize(s));
    return CPOL_OK;
}

// ------------------------------------------------------------------------------------
// Drop-in symbols of the reference's two C files.
// ------------------------------------------------------------------------------------
static std::mutex g_mutex;
static cpol_ctx *g_ctx = nullptr;

static cpol_ctx *global_ctx()
{
    if (!g_ctx) cpol_create(0, &g_ctx);
    return g_ctx;
}

extern "C" float *get_all_radar_pts(float *output, int len, float *coords_rad_pts, int crp_x, int crp_y,
                                    float *radar_heights, int rh_x, float *model_data, int md_x, int md_y, int md_z,
                                    float *model_heights, int mh_x, int mh_y, int mh_z,
                                    float *llc_cosmo, int llc_cosmo_x, float *res_cosmo, int res_cosmo_x)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    cpol_ctx *ctx = global_ctx();
    (void)len; (void)rh_x; (void)llc_cosmo_x; (void)res_cosmo_x;
    if (!ctx || crp_y != 2 || md_x != mh_x || md_y != mh_y || md_z != mh_z) return nullptr;
    int vid = 0;
    const float *data[1] = {model_data};
    float urc[2] = {llc_cosmo[0] + res_cosmo[0] * (md_z - 1), llc_cosmo[1] + res_cosmo[1] * (md_y - 1)};
    double pole[2] = {0, 0};
    if (cpol_upload_model(ctx, 1, &vid, data, model_heights, md_x, md_y, md_z, llc_cosmo, urc, res_cosmo, pole) != CPOL_OK)
        return nullptr;
    if (cpol_interp_points(ctx, crp_x, coords_rad_pts, radar_heights, output) != CPOL_OK) return nullptr;
    return output;
}

extern "C" float *get_refl(float *refl, int len, float *Da, int len_D_bins_x, int o2, float *Db, int o3, int o4,
                           float *D, int o5, int o6, float *rcs, int o8, int o9, float *N0, int o10,
                           float *mu, int o11, float *lambda, int o12, float *nu, int o13,
                           float *step_D, int o14, float *Dmin, int o15)
{
    std::lock_guard<std::mutex> lock(g_mutex);
    cpol_ctx *ctx = global_ctx();
    (void)o2; (void)o3; (void)o4; (void)o6; (void)o8; (void)o9; (void)o10; (void)o11; (void)o12; (void)o13; (void)o14; (void)o15;
    if (!ctx || len < len_D_bins_x) return nullptr;
    hipError_t err = hipSetDevice(ctx->device);
    auto keep = [&err](hipError_t r) { if (err == hipSuccess) err = r; };      // first failure wins
    hipStream_t s = ctx->stream;
    const int nb = len_D_bins_x, nd = o5;
    size_t fb = sizeof(float);
    size_t total = (size_t)len + 2 * (size_t)nb * 3 + 2 * (size_t)nd * 3 + 18;
    if (ensure(ctx, ctx->w->t_a, total * fb) != CPOL_OK) return nullptr;
    float *d_refl = (float *)ctx->w->t_a.p, *d_Da = d_refl + len, *d_Db = d_Da + nb * 3, *d_D = d_Db + nb * 3,
          *d_rcs = d_D + nd * 3, *d_par = d_rcs + nd * 3;
    keep(hipMemsetAsync(d_refl, 0, len * fb, s));
    keep(hipMemcpyAsync(d_Da, Da, nb * 3 * fb, hipMemcpyDefault, s));
    keep(hipMemcpyAsync(d_Db, Db, nb * 3 * fb, hipMemcpyDefault, s));
    keep(hipMemcpyAsync(d_D, D, nd * 3 * fb, hipMemcpyDefault, s));
    keep(hipMemcpyAsync(d_rcs, rcs, nd * 3 * fb, hipMemcpyDefault, s));
    const float *par[6] = {N0, mu, lambda, nu, step_D, Dmin};
    for (int i = 0; i < 6; i++) keep(hipMemcpyAsync(d_par + 3 * i, par[i], 3 * fb, hipMemcpyDefault, s));
    if (nb > 0)
        launch_get_refl(s, d_refl, nb, d_Da, d_Db, d_D, d_rcs, d_par, d_par + 3, d_par + 6, d_par + 9, d_par + 12, d_par + 15);
    keep(hipMemcpyAsync(refl, d_refl, len * fb, hipMemcpyDefault, s));
    keep(hipStreamSynchronize(s));
    return err == hipSuccess ? refl : nullptr;
}
