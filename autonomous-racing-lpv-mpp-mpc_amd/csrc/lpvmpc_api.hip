// lpvmpc_api.hip -- the C ABI of liblpvmpc.so (declared in include/lpvmpc.h): handle management,
// workspace, host<->device staging and kernel launches.  No CPU compute path exists here: every
// compute entry point fails with LPVMPC_E_NODEVICE when no HIP device is usable.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lpvmpc_handle.hpp"

static thread_local std::string g_last_error;
static thread_local int g_last_code = 0;

int lpvmpc_fail(lpvmpc_handle *h, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    g_last_error = buf;
    g_last_code = code;
    return code;
}

extern "C" int lpvmpc_version(void) { return LPVMPC_VERSION; }

extern "C" void lpvmpc_default_config(int32_t kind, lpvmpc_config *c) {
    std::memset(c, 0, sizeof(*c));
    c->kind = kind; c->device = 0;
    c->lf = 0.125; c->lr = 0.125; c->m = 1.98; c->Iz = 0.03; c->Cf = 60.0; c->Cr = 60.0; c->mu = 0.05;
    c->max_vel = 5.0; c->min_vel = 0.9;
    if (kind == LPVMPC_KIND_CONTROLLER) {
        c->N = 20; c->dt = 1.0 / 30.0;
        const double q[6] = {400.0, 1.0, 1.0, 20.0, 0.0, 1100.0};    // racing tuning, controllerMain.py:146-148
        for (int i = 0; i < 6; ++i) c->Q[i * 6 + i] = q[i];
        c->dR[0] = 100.0; c->dR[1] = 45.0;
    } else {
        c->N = 30; c->dt = 0.05;
        const double q[5] = {0.000000000000088, 9.703658572659423, 0.5, -0.000000000213635, 0.153591566469547};
        for (int i = 0; i < 5; ++i) c->Q[i * 5 + i] = q[i];           // plannerMain.py:96
        const double l[5] = {-1.00702414775175, -0.187661946033823, 0.0, -0.0, 0.0329493219494661};
        for (int i = 0; i < 5; ++i) c->L_cf[i] = l[i];                // plannerMain.py:97
        c->R[0] = 0.8; c->dR[0] = 6.0; c->dR[1] = 6.0;               // plannerMain.py:98-99
    }
    c->ctrl_vx_min = 0.01; c->ctrl_delta_max = 0.249; c->ctrl_a_max = 4.0; c->ctrl_a_min_abs = 1.0;
    const double xmin[5] = {0.9, -1, -2, -0.2, -0.8}, xmax[5] = {5.0, 1, 2, 0.2, 0.8};
    for (int i = 0; i < 5; ++i) { c->plan_xmin[i] = xmin[i]; c->plan_xmax[i] = xmax[i]; }
    c->plan_umin[0] = -0.249; c->plan_umin[1] = -0.7; c->plan_umax[0] = 0.249; c->plan_umax[1] = 2.0;
    c->rho = 0.1; c->sigma = 1e-6; c->alpha = 1.6; c->eps_abs = 1e-3; c->eps_rel = 1e-3;
    c->eps_prim_inf = 1e-4; c->eps_dual_inf = 1e-4; c->polish_delta = 1e-6; c->adaptive_rho_tolerance = 5.0;
    c->max_iter = 4000; c->check_termination = 25; c->scaling = 10; c->adaptive_rho = 1;
    c->adaptive_rho_interval = 25; c->polish = 1; c->polish_refine_iter = 3;
    c->track_rows = 0;
}

static void free_ws(lpvmpc_handle *h) {
    void *ptrs[] = {h->d_x0, h->d_uprev, h->d_vel, h->d_curv, h->d_uold, h->d_maxey, h->d_AB, h->d_states,
                    h->d_xPred, h->d_uPred, h->d_resid, h->d_xlast, h->d_delta, h->d_status, h->d_iters, h->d_polish, h->d_state, h->d_scal};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    h->d_x0 = h->d_uprev = h->d_vel = h->d_curv = h->d_uold = h->d_maxey = h->d_AB = h->d_states = nullptr;
    h->d_xPred = h->d_uPred = h->d_resid = h->d_xlast = h->d_delta = nullptr;
    h->d_status = h->d_iters = h->d_polish = nullptr; h->d_state = nullptr; h->d_scal = nullptr; h->state_valid_B = 0;
    h->cap = 0;
}

static const int kEventRing = 1024;      // event pairs kept by lpvmpc_set_timing (main launches, and separately resume passes)
extern "C" int lpvmpc_join(lpvmpc_handle *h, void *stream);
// ---- straggler deferral: the two pools ---------------------------------------------------------------------------------
static void free_defer(lpvmpc_handle *h) {
    for (int i = 0; i < 2; ++i) {
        if (h->dpool[i]) (void)hipFree(h->dpool[i]);
        if (h->dcount[i]) (void)hipFree(h->dcount[i]);
        h->dpool[i] = nullptr; h->dcount[i] = nullptr;
    }
    h->defer_cur_cap = 0;                   // (the counters of lpvmpc_defer_stats live as long as the handle: see ensure_defer / lpvmpc_destroy)
}
// an entry holds the LDS image of whichever kernel variant runs (the run-time-horizon kernel keeps its factor tiles in LDS:
// the largest image) plus the loop scalars and output pointers
static int entry_stride_for(int N) { return (N + 1) * (3 * lpvmpc::kTS + 19 * 8 + 8) + 16 + 64 + 8 + 80 + 64 + lpvmpc::kParkScalars; }
static int defer_entry_stride(const lpvmpc_handle *h) { return entry_stride_for(h->cfg.N); }
static int ensure_defer(lpvmpc_handle *h, int B, hipStream_t st) {
    const int cap = h->defer_cap > 0 ? h->defer_cap : (B / 8 > 64 ? B / 8 : 64);
    const int stride = defer_entry_stride(h);
    if (h->dpool[0] && h->defer_cur_cap >= cap && h->defer_stride == stride) return LPVMPC_OK;
    if (h->dpool[0]) {      // growing: finish what is parked in the old pools first
        int rc = lpvmpc_join(h, (void *)(h->defer_stream_set ? h->defer_stream : st)); if (rc) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->defer_stream_set ? h->defer_stream : st));
    }
    free_defer(h);
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(h, hipMalloc((void **)&h->dpool[i], (size_t)cap * stride * 8));
        HIP_TRY(h, hipMalloc((void **)&h->dcount[i], 16));
        HIP_TRY(h, hipMemsetAsync(h->dcount[i], 0, 16, st));
    }
    if (!h->defer_event) HIP_TRY(h, hipEventCreateWithFlags(&h->defer_event, hipEventDisableTiming));
    if (!h->dstats) { HIP_TRY(h, hipMalloc((void **)&h->dstats, 16)); HIP_TRY(h, hipMemsetAsync(h->dstats, 0, 16, st)); }
    h->defer_cur_cap = cap; h->defer_stride = stride; h->dcur = 0;
    return LPVMPC_OK;
}
// one resume pass on `st`: continues the entries of pool[dcur] for `budget` more iterations (0 = to completion), parks the
// unfinished ones in the other pool, which becomes the current one
static int resume_pass(lpvmpc_handle *h, int budget, hipStream_t st) {
    const int A = h->dcur, Bp = 1 - A;
    SolveArgs a{};
    a.B = h->defer_cur_cap; a.resume = 1; a.defer_after = budget; a.tail = h->defer_tail;
    a.pool_in = h->dpool[A]; a.pool_in_count = h->dcount[A];
    a.pool = h->dpool[Bp]; a.pool_count = h->dcount[Bp];
    a.pool_cap = h->defer_cur_cap; a.pool_stride = h->defer_stride; a.x0_stride = h->nx; a.defer_stats = h->dstats;
    const int slot = h->rv_count % kEventRing;
    if (h->timing) HIP_TRY(h, hipEventRecord(h->rv0[slot], st));
    HIP_TRY(h, lpvmpc::launch_solve(h->dev, h->d_cfg, a, st, h->force_generic));
    if (h->timing) { HIP_TRY(h, hipEventRecord(h->rv1[slot], st)); h->rv_count++; }
    h->dcur = Bp;
    return LPVMPC_OK;
}

static int ensure_ws(lpvmpc_handle *h, int B) {
    if (B <= h->cap) return LPVMPC_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    free_ws(h);
    const size_t N = h->cfg.N, nx = h->nx, nb = h->nb, b = B;
#define ALLOC(p, n) HIP_TRY(h, hipMalloc((void **)&(p), (n)))
    ALLOC(h->d_x0, b * nx * 8); ALLOC(h->d_uprev, b * N * 2 * 8); ALLOC(h->d_vel, b * (N + 1) * 8);
    ALLOC(h->d_curv, b * (N + 1) * 8); ALLOC(h->d_uold, b * (2 + h->cfg.steering_delay) * 8); ALLOC(h->d_maxey, b * 8);
    ALLOC(h->d_AB, b * N * nx * nb * 8); ALLOC(h->d_states, b * N * nx * 8);
    ALLOC(h->d_xPred, b * (N + 1) * nx * 8); ALLOC(h->d_uPred, b * N * 2 * 8); ALLOC(h->d_resid, b * 4 * 8);
    ALLOC(h->d_xlast, b * N * 6 * 8); ALLOC(h->d_delta, b * N * 8);
    ALLOC(h->d_state, b * 3 * (N + 1) * 8 * 8);
    if (h->cfg.kind == LPVMPC_KIND_PLANNER && N == 30) ALLOC(h->d_scal, b * 3 * (N + 1) * 8 * 8);      // see SolveArgs::scal
    ALLOC(h->d_status, b * 4); ALLOC(h->d_iters, b * 4); ALLOC(h->d_polish, b * 4);
#undef ALLOC
    h->cap = B;
    return LPVMPC_OK;
}

extern "C" lpvmpc_handle *lpvmpc_create(const lpvmpc_config *cfg) {
    if (!cfg) { fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: cfg is NULL"); return nullptr; }
    if (cfg->kind != LPVMPC_KIND_CONTROLLER && cfg->kind != LPVMPC_KIND_PLANNER) { fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: bad kind %d", cfg->kind); return nullptr; }
    if (cfg->N < 8 || cfg->N > LPVMPC_MAX_N) { fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: N=%d outside [8,%d]", cfg->N, LPVMPC_MAX_N); return nullptr; }
    if (cfg->steering_delay < 0 || cfg->steering_delay > 8 || cfg->steering_delay >= cfg->N || (cfg->kind == LPVMPC_KIND_PLANNER && cfg->steering_delay != 0)) {
        fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: steering_delay=%d (controller: 0..8 and < N; planner: 0)", cfg->steering_delay); return nullptr; }
    if (cfg->track_rows < 0 || cfg->track_rows > LPVMPC_MAX_TRACK_ROWS) { fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: track_rows=%d outside [0,%d]", cfg->track_rows, LPVMPC_MAX_TRACK_ROWS); return nullptr; }
    if (!(cfg->dt > 0) || !(cfg->rho > 0) || !(cfg->sigma > 0) || !(cfg->alpha > 0 && cfg->alpha < 2) || !(cfg->polish_delta > 0)) {
        fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: dt, rho, sigma, polish_delta must be > 0 and 0 < alpha < 2"); return nullptr; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { fail(nullptr, LPVMPC_E_NODEVICE, "lpvmpc_create: no HIP device available (there is no CPU fallback)"); return nullptr; }
    if (cfg->device < 0 || cfg->device >= ndev) { fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: device %d of %d", cfg->device, ndev); return nullptr; }
    lpvmpc_handle *h = new (std::nothrow) lpvmpc_handle();
    if (!h) { fail(nullptr, LPVMPC_E_NOMEM, "out of host memory"); return nullptr; }
    h->cfg = *cfg;
    h->nx = cfg->kind == LPVMPC_KIND_CONTROLLER ? 6 : 5; h->nb = h->nx + 2;
    h->d_cfg = nullptr; h->cap = 0; h->force_generic = 0; h->timing = false; h->last_ms = -1.0; h->stream = nullptr; h->ev_count = 0;
    h->d_x0 = h->d_uprev = h->d_vel = h->d_curv = h->d_uold = h->d_maxey = h->d_AB = h->d_states = nullptr;
    h->d_xPred = h->d_uPred = h->d_resid = h->d_xlast = h->d_delta = nullptr;
    h->d_status = h->d_iters = h->d_polish = nullptr; h->d_state = nullptr; h->d_scal = nullptr; h->warm_mode = 0; h->state_valid_B = 0;
    h->cl_plant = h->cl_local = h->cl_cmd = nullptr; h->cl_local_next = nullptr; h->cl_next_valid = 0; h->cl_B = 0; h->cl_first_it = 1; h->cl_q9 = 1; h->cl_ticks = 0;
    h->d_Wop = h->d_FWop = nullptr; h->ho_M = 0; h->cascade = nullptr; h->cascade_owner = nullptr; h->cascade_prefetch = 1;
    h->defer_after = 0; h->defer_budget = 200; h->defer_cap = 0; h->defer_cur_cap = 0; h->defer_stride = 0; h->rv_count = 0;
    h->dpool[0] = h->dpool[1] = nullptr; h->dcount[0] = h->dcount[1] = nullptr; h->dstats = nullptr; h->dcur = 0; h->defer_stream = nullptr; h->defer_event = nullptr;
    h->defer_stream_set = false; h->defer_tail = 1; h->defer_skip_pass = false;
    h->h_pack_in = h->h_pack_out = h->d_pack_in = h->d_pack_out = nullptr;
    DevCfg &d = h->dev;
    std::memset(&d, 0, sizeof(d));
    d.kind = cfg->kind; d.N = cfg->N; d.track_rows = cfg->track_rows; d.max_iter = cfg->max_iter;
    d.check_termination = cfg->check_termination; d.scaling = cfg->scaling; d.adaptive_rho = cfg->adaptive_rho;
    d.adaptive_rho_interval = cfg->adaptive_rho_interval; d.polish = cfg->polish; d.polish_refine_iter = cfg->polish_refine_iter;
    d.steering_delay = cfg->steering_delay;
    d.dt = cfg->dt; d.lf = cfg->lf; d.lr = cfg->lr; d.m = cfg->m; d.Iz = cfg->Iz; d.Cf = cfg->Cf; d.Cr = cfg->Cr; d.mu = cfg->mu;
    d.max_vel = cfg->max_vel; d.min_vel = cfg->min_vel;
    std::memcpy(d.Q, cfg->Q, sizeof(d.Q)); std::memcpy(d.R, cfg->R, sizeof(d.R));
    std::memcpy(d.dR, cfg->dR, sizeof(d.dR)); std::memcpy(d.Lcf, cfg->L_cf, sizeof(d.Lcf));
    const double inf = INFINITY;
    if (cfg->kind == LPVMPC_KIND_CONTROLLER) {
        // rows: -vx <= -vx_min, vx <= max_vel, d <= dmax, -d <= dmax, a <= amax, -a <= amin   (CTRL:334-348)
        const double hi[6] = {-cfg->ctrl_vx_min, cfg->max_vel, cfg->ctrl_delta_max, cfg->ctrl_delta_max, cfg->ctrl_a_max, cfg->ctrl_a_min_abs};
        for (int r = 0; r < 6; ++r) { d.box_lo[r] = -inf; d.box_hi[r] = hi[r]; }
    } else {
        for (int r = 0; r < 5; ++r) { d.box_lo[r] = cfg->plan_xmin[r]; d.box_hi[r] = cfg->plan_xmax[r]; }
        d.box_lo[0] = cfg->min_vel; d.box_hi[0] = cfg->max_vel;                 // PLAN:176-177
        for (int r = 0; r < 2; ++r) { d.box_lo[5 + r] = cfg->plan_umin[r]; d.box_hi[5 + r] = cfg->plan_umax[r]; }
    }
    d.rho = cfg->rho; d.sigma = cfg->sigma; d.alpha = cfg->alpha; d.eps_abs = cfg->eps_abs; d.eps_rel = cfg->eps_rel;
    d.eps_prim_inf = cfg->eps_prim_inf; d.eps_dual_inf = cfg->eps_dual_inf; d.delta = cfg->polish_delta;
    d.rho_tol = cfg->adaptive_rho_tolerance;
    std::memcpy(d.track, cfg->track, sizeof(double) * 6 * cfg->track_rows);
    if (hipSetDevice(cfg->device) != hipSuccess) { fail(nullptr, LPVMPC_E_HIP, "hipSetDevice(%d) failed", cfg->device); delete h; return nullptr; }
    if (hipMalloc((void **)&h->d_cfg, sizeof(DevCfg)) != hipSuccess ||
        hipMemcpy(h->d_cfg, &h->dev, sizeof(DevCfg), hipMemcpyHostToDevice) != hipSuccess) {
        fail(nullptr, LPVMPC_E_HIP, "uploading the configuration failed"); lpvmpc_destroy(h); return nullptr; }
    const size_t lds = lpvmpc::solve_lds_bytes(cfg->kind, cfg->N);
    if (lds > 160 * 1024) { fail(nullptr, LPVMPC_E_ARG, "lpvmpc_create: N=%d needs %zu B of LDS per instance (> 160 KiB)", cfg->N, lds); lpvmpc_destroy(h); return nullptr; }
    return h;
}

extern "C" void lpvmpc_destroy(lpvmpc_handle *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    (void)hipDeviceSynchronize();          // launches of this handle may still be running on the caller's streams (deferred calls: their resume passes)
    free_ws(h);
    if (h->cl_plant) (void)hipFree(h->cl_plant);
    if (h->cl_local_next) (void)hipFree(h->cl_local_next);
    if (h->cl_local) (void)hipFree(h->cl_local);
    if (h->cl_cmd) (void)hipFree(h->cl_cmd);
    if (h->cascade) lpvmpc_cascade_free(h);
    // a planner handle that a controller's cascade still drives: end that cascade first (it holds a pointer to this handle)
    if (h->cascade_owner && h->cascade_owner->cascade) lpvmpc_cascade_free(h->cascade_owner);
    if (h->h_pack_in) (void)hipHostFree(h->h_pack_in);
    if (h->h_pack_out) (void)hipHostFree(h->h_pack_out);
    if (h->d_pack_in) (void)hipFree(h->d_pack_in);
    if (h->d_pack_out) (void)hipFree(h->d_pack_out);
    if (h->d_Wop) (void)hipFree(h->d_Wop);
    if (h->d_FWop) (void)hipFree(h->d_FWop);
    if (h->d_cfg) (void)hipFree(h->d_cfg);
    free_defer(h);
    if (h->dstats) (void)hipFree(h->dstats);
    if (h->defer_event) (void)hipEventDestroy(h->defer_event);
    for (hipEvent_t e : h->rv0) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->rv1) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev0) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev1) (void)hipEventDestroy(e);
    delete h;
}

extern "C" int lpvmpc_last_error_code(void) { return g_last_code; }
extern "C" const char *lpvmpc_last_error(const lpvmpc_handle *h) { return h ? h->err.c_str() : g_last_error.c_str(); }

extern "C" int lpvmpc_set_option(lpvmpc_handle *h, const char *name, int32_t value) {
    if (!h || !name) return fail(h, LPVMPC_E_ARG, "lpvmpc_set_option: bad arguments");
    if ((std::strcmp(name, "force_generic_kernel") == 0 || std::strcmp(name, "kernel_variant") == 0) && h->dpool[0]) {
        // parked instances hold the LDS image of the kernel variant that parked them: finish them before the variant changes
        int rc = lpvmpc_join(h, (void *)h->defer_stream); if (rc) return rc;
        HIP_TRY(h, hipStreamSynchronize(h->defer_stream));
    }
    if (std::strcmp(name, "force_generic_kernel") == 0) { h->force_generic = value != 0 ? 1 : 0; return LPVMPC_OK; }
    if (std::strcmp(name, "warm_start") == 0) {
        if (value < 0 || value > 2) return fail(h, LPVMPC_E_ARG, "warm_start must be 0 (off), 1 (previous solution) or 2 (shifted by one stage)");
        h->warm_mode = value; h->state_valid_B = 0; return LPVMPC_OK;
    }
    if (std::strcmp(name, "kernel_variant") == 0) { if (value < 0 || value > 9) return fail(h, LPVMPC_E_ARG, "kernel_variant must be 0 .. 9"); h->force_generic = value; return LPVMPC_OK; }
    if (std::strcmp(name, "defer_after") == 0) {
        if (value < 0) return fail(h, LPVMPC_E_ARG, "defer_after must be >= 0 (iterations; 0 = off)");
        if (value == 0 && h->defer_after > 0 && h->dpool[0]) { int rc = lpvmpc_join(h, (void *)h->defer_stream); if (rc) return rc; }   // nothing stays parked
        h->defer_after = value; return LPVMPC_OK;
    }
    if (std::strcmp(name, "defer_budget") == 0) {
        if (value < -1) return fail(h, LPVMPC_E_ARG, "defer_budget must be >= -1 (iterations per resume pass; 0 = every pass runs to completion; -1 = no pass behind a call: parked instances wait for lpvmpc_join)");
        h->defer_budget = value; return LPVMPC_OK;
    }
    if (std::strcmp(name, "defer_pool") == 0) {
        if (value < 0) return fail(h, LPVMPC_E_ARG, "defer_pool must be >= 0 (pool entries; 0 = max(64, B / 8))");
        HIP_TRY(h, hipSetDevice(h->cfg.device));
        if (h->dpool[0]) { int rc = lpvmpc_join(h, (void *)h->defer_stream); if (rc) return rc; HIP_TRY(h, hipStreamSynchronize(h->defer_stream)); }
        free_defer(h); h->defer_cap = value; return LPVMPC_OK;
    }
    if (std::strcmp(name, "defer_tail") == 0) { h->defer_tail = value != 0 ? 1 : 0; return LPVMPC_OK; }      // both kernels continue the same pool entries
    if (std::strcmp(name, "cascade_prefetch") == 0) { h->cascade_prefetch = value != 0 ? 1 : 0; return LPVMPC_OK; }   // read by lpvmpc_cascade_init
    return fail(h, LPVMPC_E_ARG, "lpvmpc_set_option: unknown option '%s'", name);
}

extern "C" int lpvmpc_reserve(lpvmpc_handle *h, int32_t B) {
    if (!h || B <= 0) return fail(h, LPVMPC_E_ARG, "lpvmpc_reserve: bad arguments");
    return ensure_ws(h, B);
}


extern "C" int lpvmpc_set_timing(lpvmpc_handle *h, int32_t on) {
    if (!h) return LPVMPC_E_ARG;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (on && h->ev0.empty()) {
        h->ev0.resize(kEventRing); h->ev1.resize(kEventRing);
        for (int i = 0; i < kEventRing; ++i) { HIP_TRY(h, hipEventCreate(&h->ev0[i])); HIP_TRY(h, hipEventCreate(&h->ev1[i])); }
    }
    if (on && h->rv0.empty()) {
        h->rv0.resize(kEventRing); h->rv1.resize(kEventRing);
        for (int i = 0; i < kEventRing; ++i) { HIP_TRY(h, hipEventCreate(&h->rv0[i])); HIP_TRY(h, hipEventCreate(&h->rv1[i])); }
    }
    h->timing = on != 0; h->ev_count = 0; h->rv_count = 0; h->last_ms = -1.0;
    return LPVMPC_OK;
}

extern "C" int lpvmpc_kernel_time_stats(lpvmpc_handle *h, double *total_ms, int32_t *count) {
    if (!h || !total_ms || !count) return LPVMPC_E_ARG;
    const int n = h->ev_count < kEventRing ? h->ev_count : kEventRing;
    double tot = 0.0;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_TRY(h, hipEventSynchronize(h->ev1[i]));
        HIP_TRY(h, hipEventElapsedTime(&ms, h->ev0[i], h->ev1[i]));
        tot += ms; h->last_ms = ms;
    }
    *total_ms = tot; *count = n;
    return LPVMPC_OK;
}

extern "C" double lpvmpc_last_kernel_ms(lpvmpc_handle *h) {
    if (!h) return -1.0;
    double tot; int32_t n;
    if (lpvmpc_kernel_time_stats(h, &tot, &n) != LPVMPC_OK || n == 0) return -1.0;
    return h->last_ms;
}

// ------------------------------------------------------------------------------------------------
int lpvmpc_need_track(lpvmpc_handle *h, const char *who) {
    if (h->cfg.track_rows < 1) return fail(h, LPVMPC_E_ARG, "%s: the handle was created without a track table", who);
    return LPVMPC_OK;
}

static int launch_lpv(lpvmpc_handle *h, int B, const double *x0, const double *u_prev, const double *vel_ref,
                      const double *curv_s, double cf_new, int lap, double *states, double *AB, hipStream_t st) {
    HIP_TRY(h, lpvmpc::launch_lpv(h->dev, h->d_cfg, B, x0, u_prev, vel_ref, curv_s, cf_new, lap, states, AB, st));
    return LPVMPC_OK;
}

int lpvmpc_launch_solve_timed(lpvmpc_handle *h, const SolveArgs &a, hipStream_t st) {
    const int slot = h->ev_count % kEventRing;
    if (h->timing) HIP_TRY(h, hipEventRecord(h->ev0[slot], st));
    SolveArgs b = a;
    if (h->d_scal && a.B <= h->cap) b.scal = h->d_scal;          // (the launcher ignores it for deferred / resumed launches)
    HIP_TRY(h, lpvmpc::launch_solve(h->dev, h->d_cfg, b, st, h->force_generic));
    if (h->timing) { HIP_TRY(h, hipEventRecord(h->ev1[slot], st)); h->ev_count++; }
    return LPVMPC_OK;
}

// The closed-loop fleet (lpvmpc_cl_*) and the cascade keep their state between ticks in the handle's workspace (reference
// windows, receding-horizon inputs, last commands, statuses): a stand-alone batch call on the same handle would overwrite it
// -- or, with a larger B, reallocate it -- without any error.  Such calls are refused; use a second handle.
int lpvmpc_check_batch(lpvmpc_handle *h, int B, const char *who) {
    if (h && (h->cl_plant || h->cascade || h->cascade_owner))
        return fail(h, LPVMPC_E_ARG, "%s: this handle runs a %s whose state lives in its workspace; use another handle for batch calls "
                    "(lpvmpc_cl_release ends the fleet)", who, h->cl_plant ? "closed-loop fleet" : "planner + controller cascade");
    return lpvmpc_check_common(h, B, who);
}

int lpvmpc_check_common(lpvmpc_handle *h, int B, const char *who) {
    if (!h) return fail(nullptr, LPVMPC_E_ARG, "%s: handle is NULL", who);
    if (B <= 0) return fail(h, LPVMPC_E_ARG, "%s: B=%d", who, B);
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    return ensure_ws(h, B);
}



// ------------------------------------------------------------------------------------------------
// Small batches (the 30 Hz control loop calls with B = 1): every host array of a call travels through ONE pinned
// staging buffer and ONE copy per direction instead of one pageable copy per array (11 copies per fused tick).
// Large batches keep the direct copies.  in() / out() return the device pointer the kernels must use.
// ------------------------------------------------------------------------------------------------
static const size_t kPackBytes = 512 * 1024;
struct IoPack {
    lpvmpc_handle *h; hipStream_t st; bool on; size_t in_off, out_off;
    struct Out { void *user, *dev; size_t off, bytes; } outs[8];
    int n_out;
    static size_t al(size_t n) { return (n + 255) & ~(size_t)255; }
    int begin(lpvmpc_handle *h_, hipStream_t st_, size_t in_bytes, size_t out_bytes, int n_arrays) {
        h = h_; st = st_; in_off = out_off = 0; n_out = 0;
        on = in_bytes + 256 * (size_t)n_arrays <= kPackBytes && out_bytes + 256 * (size_t)n_arrays <= kPackBytes;
        if (on && !h->h_pack_in) {
            HIP_TRY(h, hipHostMalloc((void **)&h->h_pack_in, kPackBytes)); HIP_TRY(h, hipHostMalloc((void **)&h->h_pack_out, kPackBytes));
            HIP_TRY(h, hipMalloc((void **)&h->d_pack_in, kPackBytes)); HIP_TRY(h, hipMalloc((void **)&h->d_pack_out, kPackBytes));
        }
        return LPVMPC_OK;
    }
    // host staging area of an input (packed mode) -- lets a caller write a re-laid-out array directly
    void *in_slot(size_t bytes, void **dev) { void *p = h->h_pack_in + in_off; *dev = h->d_pack_in + in_off; in_off += al(bytes); return p; }
    int in(const void *src, void *dev_default, size_t bytes, void **dev) {
        if (!on) { *dev = dev_default; HIP_TRY(h, hipMemcpyAsync(dev_default, src, bytes, hipMemcpyHostToDevice, st)); return LPVMPC_OK; }
        std::memcpy(in_slot(bytes, dev), src, bytes);
        return LPVMPC_OK;
    }
    int flush_in() {
        if (on && in_off) HIP_TRY(h, hipMemcpyAsync(h->d_pack_in, h->h_pack_in, in_off, hipMemcpyHostToDevice, st));
        return LPVMPC_OK;
    }
    void *out(void *user, void *dev_default, size_t bytes) {          // user may be NULL (result not wanted)
        void *dev = on ? (void *)(h->d_pack_out + out_off) : dev_default;
        outs[n_out++] = Out{user, dev, out_off, bytes};
        if (on) out_off += al(bytes);
        return dev;
    }
    const void *host_of(int i) const { return h->h_pack_out + outs[i].off; }           // valid after flush_out in packed mode
    int flush_out() {
        if (on) {
            if (out_off) HIP_TRY(h, hipMemcpyAsync(h->h_pack_out, h->d_pack_out, out_off, hipMemcpyDeviceToHost, st));
            HIP_TRY(h, hipStreamSynchronize(st));
            for (int i = 0; i < n_out; ++i) if (outs[i].user) std::memcpy(outs[i].user, h->h_pack_out + outs[i].off, outs[i].bytes);
        } else {
            for (int i = 0; i < n_out; ++i) if (outs[i].user) HIP_TRY(h, hipMemcpyAsync(outs[i].user, outs[i].dev, outs[i].bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(h, hipStreamSynchronize(st));
        }
        return LPVMPC_OK;
    }
};
#define IO_TRY(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

extern "C" int lpvmpc_lpv_batch(lpvmpc_handle *h, int32_t B, const double *x0, const double *u_prev,
                                const double *vel_ref, const double *curv_s, double cf_new, int32_t lap,
                                double *states, double *A, double *Bm) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_lpv_batch"); if (rc) return rc;
    const bool ctrl = h->cfg.kind == LPVMPC_KIND_CONTROLLER;
    if (!x0 || !u_prev) return fail(h, LPVMPC_E_ARG, "lpvmpc_lpv_batch: x0 / u_prev is NULL");
    if (ctrl && !vel_ref) return fail(h, LPVMPC_E_ARG, "lpvmpc_lpv_batch: controller needs vel_ref");
    if (ctrl && lap != 0 && !curv_s) return fail(h, LPVMPC_E_ARG, "lpvmpc_lpv_batch: controller with lap != 0 needs curv_ref");
    if (!ctrl && !curv_s) return fail(h, LPVMPC_E_ARG, "lpvmpc_lpv_batch: planner needs SS");
    if ((ctrl && lap == 0) || !ctrl) { rc = lpvmpc_need_track(h, "lpvmpc_lpv_batch"); if (rc) return rc; }
    const size_t N = h->cfg.N, nx = h->nx, nb = h->nb, b = B;
    hipStream_t st = h->stream;
    const size_t n_ab = b * N * nx * nb, n_st = b * N * nx;
    IoPack io;
    IO_TRY(io.begin(h, st, (b * nx + b * N * 2 + b * (N + 1) * 2) * 8, (n_ab + n_st) * 8, 6));
    void *p_x0, *p_up, *p_vel = nullptr, *p_curv = nullptr;
    IO_TRY(io.in(x0, h->d_x0, b * nx * 8, &p_x0)); IO_TRY(io.in(u_prev, h->d_uprev, b * N * 2 * 8, &p_up));
    if (ctrl) IO_TRY(io.in(vel_ref, h->d_vel, b * (N + 1) * 8, &p_vel));
    if (curv_s) IO_TRY(io.in(curv_s, h->d_curv, b * (ctrl ? N : N + 1) * 8, &p_curv));
    IO_TRY(io.flush_in());
    std::vector<double> ab;
    if (!io.on && (A || Bm)) ab.resize(n_ab);
    // packed mode: [A|B] is de-interleaved straight from the staging buffer (no user copy of the raw block)
    double *p_ab = (double *)io.out(io.on ? nullptr : (void *)ab.data(), h->d_AB, n_ab * 8);
    double *p_states = (double *)io.out(states, h->d_states, n_st * 8);
    rc = launch_lpv(h, B, (const double *)p_x0, (const double *)p_up, (const double *)p_vel, (const double *)p_curv, cf_new, lap, p_states, p_ab, st);
    if (rc) return rc;
    IO_TRY(io.flush_out());
    if (A || Bm) {
        const double *src = io.on ? (const double *)io.host_of(0) : ab.data();
        for (size_t t = 0; t < b * N; ++t)
            for (size_t r = 0; r < nx; ++r) {
                const double *row = src + (t * nx + r) * nb;
                if (A) for (size_t a = 0; a < nx; ++a) A[(t * nx + r) * nx + a] = row[a];
                if (Bm) { Bm[(t * nx + r) * 2 + 0] = row[nx]; Bm[(t * nx + r) * 2 + 1] = row[nx + 1]; }
            }
    }
    return LPVMPC_OK;
}

extern "C" int lpvmpc_estimate_abc_batch(lpvmpc_handle *h, int32_t B, const double *xlast, const double *delta,
                                         double *A, double *Bm) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_estimate_abc_batch"); if (rc) return rc;
    if (!xlast || !delta) return fail(h, LPVMPC_E_ARG, "lpvmpc_estimate_abc_batch: NULL input");
    rc = lpvmpc_need_track(h, "lpvmpc_estimate_abc_batch"); if (rc) return rc;
    const size_t N = h->cfg.N, nx = h->nx, nb = h->nb, b = B;
    hipStream_t st = h->stream;
    H2D(h->d_xlast, xlast, b * N * 6 * 8); H2D(h->d_delta, delta, b * N * 8);
    HIP_TRY(h, lpvmpc::launch_abc(h->dev, h->d_cfg, B, h->d_xlast, h->d_delta, h->d_AB, st));
    std::vector<double> ab(b * N * nx * nb);
    D2H(ab.data(), h->d_AB, ab.size() * 8);
    HIP_TRY(h, hipStreamSynchronize(st));
    for (size_t t = 0; t < b * N; ++t)
        for (size_t r = 0; r < nx; ++r) {
            const double *row = ab.data() + (t * nx + r) * nb;
            if (A) for (size_t a = 0; a < nx; ++a) A[(t * nx + r) * nx + a] = row[a];
            if (Bm) { Bm[(t * nx + r) * 2 + 0] = row[nx]; Bm[(t * nx + r) * 2 + 1] = row[nx + 1]; }
        }
    return LPVMPC_OK;
}

extern "C" int lpvmpc_solve_batch_AB(lpvmpc_handle *h, int32_t B, const double *x0, const double *A, const double *Bm,
                                     const double *vel_ref, const double *u_old, const double *max_ey,
                                     double *xPred, double *uPred, int32_t *status, int32_t *iters, double *resid,
                                     int32_t *polish) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_solve_batch_AB"); if (rc) return rc;
    const bool ctrl = h->cfg.kind == LPVMPC_KIND_CONTROLLER;
    if (!x0 || !A || !Bm) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch_AB: x0 / A / B is NULL");
    if (ctrl && !vel_ref) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch_AB: controller needs vel_ref");
    if (!ctrl && !max_ey) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch_AB: planner needs max_ey");
    const size_t N = h->cfg.N, nx = h->nx, nb = h->nb, b = B;
    hipStream_t st = h->stream;
    const size_t n_ab = b * N * nx * nb, n_x = b * (N + 1) * nx, n_u = b * N * 2;
    IoPack io;
    IO_TRY(io.begin(h, st, (b * nx + n_ab + b * (N + 1) + b * (3 + h->cfg.steering_delay)) * 8, (n_x + n_u + b * 4) * 8 + b * 12, 12));
    void *p_x0, *p_ab, *p_vel = nullptr, *p_uold = nullptr, *p_mey = nullptr;
    IO_TRY(io.in(x0, h->d_x0, b * nx * 8, &p_x0));
    std::vector<double> ab;
    double *abh;
    if (io.on) abh = (double *)io.in_slot(n_ab * 8, &p_ab);                 // interleave [A|B] straight into the staging buffer
    else { ab.resize(n_ab); abh = ab.data(); }
    for (size_t t = 0; t < b * N; ++t)
        for (size_t r = 0; r < nx; ++r) {
            double *row = abh + (t * nx + r) * nb;
            for (size_t a = 0; a < nx; ++a) row[a] = A[(t * nx + r) * nx + a];
            row[nx] = Bm[(t * nx + r) * 2 + 0]; row[nx + 1] = Bm[(t * nx + r) * 2 + 1];
        }
    if (!io.on) { p_ab = h->d_AB; H2D(h->d_AB, ab.data(), n_ab * 8); }
    if (ctrl) IO_TRY(io.in(vel_ref, h->d_vel, b * (N + 1) * 8, &p_vel));
    if (u_old) IO_TRY(io.in(u_old, h->d_uold, b * (2 + h->cfg.steering_delay) * 8, &p_uold));
    if (!ctrl) IO_TRY(io.in(max_ey, h->d_maxey, b * 8, &p_mey));
    IO_TRY(io.flush_in());
    double *o_x = (double *)io.out(xPred, h->d_xPred, n_x * 8), *o_u = (double *)io.out(uPred, h->d_uPred, n_u * 8);
    int32_t *o_st = (int32_t *)io.out(status, h->d_status, b * 4), *o_it = (int32_t *)io.out(iters, h->d_iters, b * 4);
    double *o_res = (double *)io.out(resid, h->d_resid, b * 4 * 8);
    int32_t *o_pol = (int32_t *)io.out(polish, h->d_polish, b * 4);
    SolveArgs a{B, (const double *)p_x0, (const double *)p_ab, (const double *)p_vel, (const double *)p_uold, (const double *)p_mey,
                o_x, o_u, o_st, o_it, o_pol, o_res,
                h->warm_mode ? h->d_state : nullptr, (h->warm_mode && h->state_valid_B == B) ? h->warm_mode : 0, h->nx};
    rc = lpvmpc_launch_solve_timed(h, a, st); if (rc) return rc;
    if (h->warm_mode) h->state_valid_B = B;
    return io.flush_out();
}

extern "C" int lpvmpc_solve_batch_dev(lpvmpc_handle *h, int32_t B, const double *x0, const double *u_prev,
                                      const double *vel_ref, const double *curv_s, const double *u_old,
                                      const double *max_ey, double cf_new, int32_t lap, double *xPred, double *uPred,
                                      int32_t *status, int32_t *iters, double *resid, int32_t *polish, void *stream) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_solve_batch_dev"); if (rc) return rc;
    const bool ctrl = h->cfg.kind == LPVMPC_KIND_CONTROLLER;
    if (!x0 || !u_prev || !xPred || !uPred) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch_dev: NULL x0 / u_prev / xPred / uPred");
    if (ctrl && !vel_ref) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch_dev: controller needs vel_ref");
    if (ctrl && lap != 0 && !curv_s) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch_dev: controller with lap != 0 needs curv_ref");
    if (!ctrl && (!curv_s || !max_ey)) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch_dev: planner needs SS and max_ey");
    if ((ctrl && lap == 0) || !ctrl) { rc = lpvmpc_need_track(h, "lpvmpc_solve_batch_dev"); if (rc) return rc; }
    hipStream_t st = (hipStream_t)stream;
    // a warm start reads the previous call's final (x, y): with straggler deferral, finish what that call left parked first
    if (h->warm_mode && h->defer_after > 0 && h->dpool[0]) { rc = lpvmpc_join(h, stream); if (rc) return rc; }
    rc = launch_lpv(h, B, x0, u_prev, vel_ref, curv_s, cf_new, lap, nullptr, h->d_AB, st); if (rc) return rc;
    SolveArgs a{B, x0, h->d_AB, ctrl ? vel_ref : nullptr, u_old, ctrl ? nullptr : max_ey, xPred, uPred, status, iters, polish, resid,
                h->warm_mode ? h->d_state : nullptr, (h->warm_mode && h->state_valid_B == B) ? h->warm_mode : 0, h->nx};
    if (h->defer_after > 0) {
        // Straggler deferral: this launch parks what is still unsolved after defer_after iterations; the resume pass behind it
        // (same stream) continues everything that is parked on this handle -- from this call and from earlier ones -- for
        // defer_budget more iterations.  No launch lasts much longer than its budget, so the stream is never held by one slow
        // instance; lpvmpc_join runs the pass that finishes whatever is still parked.
        rc = ensure_defer(h, B, st); if (rc) return rc;
        if (h->defer_stream_set && h->defer_stream != st) {        // the pools are ordered by stream: hand them over
            HIP_TRY(h, hipEventRecord(h->defer_event, h->defer_stream));
            HIP_TRY(h, hipStreamWaitEvent(st, h->defer_event, 0));
        }
        h->defer_stream = st; h->defer_stream_set = true;
        a.defer_after = h->defer_after; a.resume = 0; a.pool = h->dpool[h->dcur]; a.pool_count = h->dcount[h->dcur];
        a.pool_cap = h->defer_cur_cap; a.pool_stride = h->defer_stride; a.defer_stats = h->dstats;
        rc = lpvmpc_launch_solve_timed(h, a, st); if (rc) return rc;
        // the bounded pass behind the call -- unless the caller joins right away (budget -1, and the synchronous host-array entry
        // point): then the closing pass (the tail kernel) takes the parked instances straight from this launch
        if (h->defer_budget >= 0 && !h->defer_skip_pass) { rc = resume_pass(h, h->defer_budget, st); if (rc) return rc; }
    } else {
        rc = lpvmpc_launch_solve_timed(h, a, st); if (rc) return rc;
    }
    if (h->warm_mode) h->state_valid_B = B;
    return LPVMPC_OK;
}

extern "C" int lpvmpc_join(lpvmpc_handle *h, void *stream) {
    if (!h) return fail(nullptr, LPVMPC_E_ARG, "lpvmpc_join: handle is NULL");
    if (!h->dpool[0]) return LPVMPC_OK;
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    if (h->defer_stream_set && h->defer_stream != st) {
        HIP_TRY(h, hipEventRecord(h->defer_event, h->defer_stream));
        HIP_TRY(h, hipStreamWaitEvent(st, h->defer_event, 0));
    }
    h->defer_stream = st; h->defer_stream_set = true;
    return resume_pass(h, 0, st);                                  // to completion: nothing stays parked
}

extern "C" int lpvmpc_resume_time_stats(lpvmpc_handle *h, double *total_ms, int32_t *count) {
    if (!h || !total_ms || !count) return LPVMPC_E_ARG;
    const int n = h->rv_count < kEventRing ? h->rv_count : kEventRing;
    double tot = 0.0;
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
        HIP_TRY(h, hipEventSynchronize(h->rv1[i]));
        HIP_TRY(h, hipEventElapsedTime(&ms, h->rv0[i], h->rv1[i]));
        tot += ms;
    }
    *total_ms = tot; *count = n;
    return LPVMPC_OK;
}

// instances parked / parking requests refused by the launches of this handle that have COMPLETED on the stream of its last deferred
// call (waits for that stream); counted since the handle's first deferred call
extern "C" int lpvmpc_defer_stats(lpvmpc_handle *h, int64_t *parked, int64_t *refused) {
    if (!h) return LPVMPC_E_ARG;
    unsigned long long v[2] = {0, 0};
    if (h->dstats) {
        HIP_TRY(h, hipSetDevice(h->cfg.device));
        if (h->defer_stream_set) HIP_TRY(h, hipStreamSynchronize(h->defer_stream));
        HIP_TRY(h, hipMemcpy(v, h->dstats, 16, hipMemcpyDeviceToHost));
    }
    if (parked) *parked = (int64_t)v[0];
    if (refused) *refused = (int64_t)v[1];
    return LPVMPC_OK;
}

extern "C" int lpvmpc_solve_batch(lpvmpc_handle *h, int32_t B, const double *x0, const double *u_prev,
                                  const double *vel_ref, const double *curv_s, const double *u_old,
                                  const double *max_ey, double cf_new, int32_t lap, double *xPred, double *uPred,
                                  int32_t *status, int32_t *iters, double *resid, int32_t *polish) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_solve_batch"); if (rc) return rc;
    const bool ctrl = h->cfg.kind == LPVMPC_KIND_CONTROLLER;
    if (!x0 || !u_prev) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch: x0 / u_prev is NULL");
    if (ctrl && !vel_ref) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch: controller needs vel_ref");
    if (ctrl && lap != 0 && !curv_s) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch: controller with lap != 0 needs curv_ref");
    if (!ctrl && (!curv_s || !max_ey)) return fail(h, LPVMPC_E_ARG, "lpvmpc_solve_batch: planner needs SS and max_ey");
    const size_t N = h->cfg.N, nx = h->nx, b = B;
    hipStream_t st = h->stream;
    const size_t n_x = b * (N + 1) * nx, n_u = b * N * 2;
    IoPack io;
    IO_TRY(io.begin(h, st, (b * nx + n_u + b * (N + 1) * 2 + b * (3 + h->cfg.steering_delay)) * 8, (n_x + n_u + b * 4) * 8 + b * 12, 12));
    void *p_x0, *p_up, *p_vel = nullptr, *p_curv = nullptr, *p_uold = nullptr, *p_mey = nullptr;
    IO_TRY(io.in(x0, h->d_x0, b * nx * 8, &p_x0)); IO_TRY(io.in(u_prev, h->d_uprev, n_u * 8, &p_up));
    if (ctrl) IO_TRY(io.in(vel_ref, h->d_vel, b * (N + 1) * 8, &p_vel));
    if (curv_s) IO_TRY(io.in(curv_s, h->d_curv, b * (ctrl ? N : N + 1) * 8, &p_curv));
    if (u_old) IO_TRY(io.in(u_old, h->d_uold, b * (2 + h->cfg.steering_delay) * 8, &p_uold));
    if (!ctrl) IO_TRY(io.in(max_ey, h->d_maxey, b * 8, &p_mey));
    IO_TRY(io.flush_in());
    double *o_x = (double *)io.out(xPred, h->d_xPred, n_x * 8), *o_u = (double *)io.out(uPred, h->d_uPred, n_u * 8);
    int32_t *o_st = (int32_t *)io.out(status, h->d_status, b * 4), *o_it = (int32_t *)io.out(iters, h->d_iters, b * 4);
    double *o_res = (double *)io.out(resid, h->d_resid, b * 4 * 8);
    int32_t *o_pol = (int32_t *)io.out(polish, h->d_polish, b * 4);
    h->defer_skip_pass = true;
    rc = lpvmpc_solve_batch_dev(h, B, (const double *)p_x0, (const double *)p_up, (const double *)p_vel, (const double *)p_curv,
                                (const double *)p_uold, (const double *)p_mey, cf_new, lap, o_x, o_u, o_st, o_it, o_res, o_pol, (void *)st);
    h->defer_skip_pass = false;
    if (rc) return rc;
    if (h->defer_after > 0) { rc = lpvmpc_join(h, (void *)st); if (rc) return rc; }      // a synchronous call returns finished instances only
    return io.flush_out();
}

// ------------------------------------------------------------------------------------------------
// caller-side helpers of the reference, batched on the device
// ------------------------------------------------------------------------------------------------
extern "C" int lpvmpc_local_position_batch(lpvmpc_handle *h, int32_t B, const double *xy_psi, double half_width, double slack, double *out) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_local_position_batch"); if (rc) return rc;
    if (!xy_psi || !out) return fail(h, LPVMPC_E_ARG, "lpvmpc_local_position_batch: NULL argument");
    rc = lpvmpc_need_track(h, "lpvmpc_local_position_batch"); if (rc) return rc;
    hipStream_t st = h->stream;
    // workspace reuse: inputs in d_xlast ([cap][N][6] >= [B][3]), outputs in d_states ([cap][N][nx] >= [B][4])
    H2D(h->d_xlast, xy_psi, (size_t)B * 3 * 8);
    HIP_TRY(h, lpvmpc::launch_local_position(h->d_cfg, B, h->d_xlast, half_width, slack, h->d_states, st));
    D2H(out, h->d_states, (size_t)B * 4 * 8);
    HIP_TRY(h, hipStreamSynchronize(st));
    return LPVMPC_OK;
}

extern "C" int lpvmpc_global_position_batch(lpvmpc_handle *h, int32_t B, const double *s_ey, double *out) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_global_position_batch"); if (rc) return rc;
    if (!s_ey || !out) return fail(h, LPVMPC_E_ARG, "lpvmpc_global_position_batch: NULL argument");
    rc = lpvmpc_need_track(h, "lpvmpc_global_position_batch"); if (rc) return rc;
    hipStream_t st = h->stream;
    H2D(h->d_xlast, s_ey, (size_t)B * 2 * 8);
    HIP_TRY(h, lpvmpc::launch_global_position(h->d_cfg, B, h->d_xlast, h->d_states, st));
    D2H(out, h->d_states, (size_t)B * 3 * 8);
    HIP_TRY(h, hipStreamSynchronize(st));
    return LPVMPC_OK;
}

lpvmpc::PlantCfg lpvmpc_plant_cfg(const lpvmpc_handle *h, int n_sub, double dt_sim, double mu_sim) {
    lpvmpc::PlantCfg pc; pc.lf = h->cfg.lf; pc.lr = h->cfg.lr; pc.m = h->cfg.m; pc.Iz = h->cfg.Iz; pc.mu = mu_sim; pc.dt = dt_sim; pc.n_sub = n_sub;
    return pc;
}

extern "C" int lpvmpc_plant_step_batch(lpvmpc_handle *h, int32_t B, double *state, const double *u, int32_t n_sub, double dt_sim, double mu_sim) {
    if (h && B == 0) return LPVMPC_OK;                                   // an empty batch is a no-op
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_plant_step_batch"); if (rc) return rc;
    if (!state || !u || n_sub < 1 || !(dt_sim > 0)) return fail(h, LPVMPC_E_ARG, "lpvmpc_plant_step_batch: bad argument");
    hipStream_t st = h->stream;
    H2D(h->d_xlast, state, (size_t)B * 8 * 8); H2D(h->d_states, u, (size_t)B * 2 * 8);
    HIP_TRY(h, lpvmpc::launch_plant(B, h->d_xlast, h->d_states, lpvmpc_plant_cfg(h, n_sub, dt_sim, mu_sim), st));
    D2H(state, h->d_xlast, (size_t)B * 8 * 8);
    HIP_TRY(h, hipStreamSynchronize(st));
    return LPVMPC_OK;
}

// ---- closed-loop fleet: controller in the lap-0 path-tracking branch of controllerMain.py ----------------
extern "C" int lpvmpc_cl_release(lpvmpc_handle *h) {
    if (!h) return fail(nullptr, LPVMPC_E_ARG, "lpvmpc_cl_release: handle is NULL");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->cascade) { HIP_TRY(h, hipDeviceSynchronize()); lpvmpc_cascade_free(h); }
    if (h->cl_plant) { (void)hipFree(h->cl_plant); (void)hipFree(h->cl_local); (void)hipFree(h->cl_cmd); (void)hipFree(h->cl_local_next); h->cl_plant = h->cl_local = h->cl_cmd = h->cl_local_next = nullptr; }
    h->cl_B = 0; h->cl_ticks = 0; h->cl_first_it = 1; h->cl_next_valid = 0;
    return LPVMPC_OK;
}

extern "C" int lpvmpc_cl_init(lpvmpc_handle *h, int32_t B, const double *plant0, double half_width, double slack, int32_t q9_swap,
                              int32_t n_sub, double dt_sim, double mu_sim) {
    int rc = lpvmpc_check_common(h, B, "lpvmpc_cl_init"); if (rc) return rc;
    if (h->cfg.kind != LPVMPC_KIND_CONTROLLER) return fail(h, LPVMPC_E_ARG, "lpvmpc_cl_init: controller handles only");
    if (h->cfg.N > 20) return fail(h, LPVMPC_E_ARG, "lpvmpc_cl_init: the reference's seed trajectories have 20 rows (N <= 20)");
    if (h->cfg.steering_delay != 0) return fail(h, LPVMPC_E_ARG, "lpvmpc_cl_init: the fleet engines run the reference's steeringDelay = 0 (CMAIN:49)");
    if (!plant0 || n_sub < 1 || !(dt_sim > 0)) return fail(h, LPVMPC_E_ARG, "lpvmpc_cl_init: bad argument");
    rc = lpvmpc_need_track(h, "lpvmpc_cl_init"); if (rc) return rc;
    if (h->cl_plant) { (void)hipFree(h->cl_plant); (void)hipFree(h->cl_local); (void)hipFree(h->cl_cmd); (void)hipFree(h->cl_local_next); h->cl_plant = h->cl_local = h->cl_cmd = h->cl_local_next = nullptr; }
    h->cl_next_valid = 0;
    HIP_TRY(h, hipMalloc((void **)&h->cl_local_next, (size_t)B * 6 * 8));
    HIP_TRY(h, hipMalloc((void **)&h->cl_plant, (size_t)B * 8 * 8));
    HIP_TRY(h, hipMalloc((void **)&h->cl_local, (size_t)B * 6 * 8));
    HIP_TRY(h, hipMalloc((void **)&h->cl_cmd, (size_t)B * 2 * 8));
    hipStream_t st = h->stream;
    H2D(h->cl_plant, plant0, (size_t)B * 8 * 8);
    HIP_TRY(h, hipMemsetAsync(h->cl_cmd, 0, (size_t)B * 2 * 8, st));
    std::vector<double> ones((size_t)B * (h->cfg.N + 1), 1.0);                  // vel_ref = 1 m/s on lap 0 (CMAIN:311,326)
    H2D(h->d_vel, ones.data(), ones.size() * 8);
    HIP_TRY(h, hipStreamSynchronize(st));
    h->cl_B = B; h->cl_first_it = 1; h->cl_q9 = q9_swap != 0; h->cl_ticks = 0; h->cl_hw = half_width; h->cl_slack = slack;
    h->cl_pc = lpvmpc_plant_cfg(h, n_sub, dt_sim, mu_sim);
    h->state_valid_B = 0;
    return LPVMPC_OK;
}

extern "C" int lpvmpc_cl_tick(lpvmpc_handle *h, int32_t n_ticks) {
    if (!h || !h->cl_plant || n_ticks < 1) return fail(h, LPVMPC_E_ARG, "lpvmpc_cl_tick: call lpvmpc_cl_init first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const int B = h->cl_B, N = h->cfg.N;
    hipStream_t st = h->stream;
    for (int t = 0; t < n_ticks; ++t) {
        // the measurement of this tick: made by the launch that advanced the plant at the end of the previous tick, or here
        if (h->cl_next_valid) { double *t_ = h->cl_local; h->cl_local = h->cl_local_next; h->cl_local_next = t_; }
        else HIP_TRY(h, lpvmpc::launch_cl_measure(h->d_cfg, B, h->cl_plant, h->cl_cmd, h->cl_hw, h->cl_slack, h->cl_q9, h->cl_local, h->d_uold, st));
        const double *x0 = h->cl_local; int x0_stride = 6;
        if (h->cl_first_it < 10) {                                           // CMAIN:310-315: seed mode
            HIP_TRY(h, lpvmpc::launch_cl_seed(B, N, h->cl_local, h->d_xlast, h->d_delta, st));
            HIP_TRY(h, lpvmpc::launch_abc(h->dev, h->d_cfg, B, h->d_xlast, h->d_delta, h->d_AB, st));
            h->cl_first_it++;
        } else {                                                             // CMAIN:325-331: LPV prediction, x0 = first rolled-out state
            HIP_TRY(h, lpvmpc::launch_lpv(h->dev, h->d_cfg, B, h->cl_local, h->d_uPred, h->d_vel, nullptr, 60.0, 0, h->d_states, h->d_AB, st));
            x0 = h->d_states; x0_stride = N * 6;
        }
        SolveArgs a{B, x0, h->d_AB, h->d_vel, h->d_uold, nullptr, h->d_xPred, h->d_uPred, h->d_status, h->d_iters, h->d_polish, h->d_resid,
                    h->warm_mode ? h->d_state : nullptr, (h->warm_mode && h->state_valid_B == B) ? h->warm_mode : 0, x0_stride};
        int rc = lpvmpc_launch_solve_timed(h, a, st); if (rc) return rc;
        if (h->warm_mode) h->state_valid_B = B;
        HIP_TRY(h, lpvmpc::launch_cl_command_plant_measure(h->d_cfg, B, N, h->d_uPred, h->cl_cmd, h->cl_plant, h->cl_pc, h->cl_hw, h->cl_slack, h->cl_q9,
                                                            h->cl_local_next, h->d_uold, st));
        h->cl_next_valid = 1;
        h->cl_ticks++;
    }
    return LPVMPC_OK;
}

extern "C" int lpvmpc_cl_read(lpvmpc_handle *h, double *plant, double *local_state, double *cmd, int32_t *iters, int32_t *status) {
    if (!h || !h->cl_plant) return fail(h, LPVMPC_E_ARG, "lpvmpc_cl_read: call lpvmpc_cl_init first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    const size_t B = h->cl_B;
    hipStream_t st = h->stream;
    if (plant) D2H(plant, h->cl_plant, B * 8 * 8);
    if (local_state) D2H(local_state, h->cl_local, B * 6 * 8);
    if (cmd) D2H(cmd, h->cl_cmd, B * 2 * 8);
    if (iters) D2H(iters, h->d_iters, B * 4);
    if (status) D2H(status, h->d_status, B * 4);
    HIP_TRY(h, hipStreamSynchronize(st));
    return LPVMPC_OK;
}
