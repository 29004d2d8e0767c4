// cascade_api.hip -- C ABI of the planner -> controller hand-off and of the
// planner + controller + plant cascade of BASELINE.json configs[4], everything resident on the device.
#include <cmath>
#include <cstring>
#include <new>

#include "lpvmpc_handle.hpp"

// fleet state of a cascade, owned by the controller handle
struct lpvmpc_cascade {
    lpvmpc_handle *plan;
    int B, M, q9;
    int ctrl_ticks, plan_ticks, index;       // `index` of CMAIN:219-235 (0: re-read the windows on this tick)
    int prefetch;                            // 1: planner tick j+1 is enqueued as soon as message j has its first reader
    int n_sub[3];
    double hw, slack, dt_sim, mu_sim;
    double *plant, *cmd, *local, *ref0, *SSc;       // [B][8], [B][2], [B][6], [B][3], [B]
    int32_t *lap, *lap_tick;                        // [B]
    int32_t *alive_ticks;                           // [B] controller ticks entered with a finite plant state
    double *refs, *sig, *SSp, *pose, *px0;          // planner side: [2][B][5][M] (message j in buffer j % 2), [B][5][Np], [B][Np+1], [B][3], [B][5]
    hipStream_t s_ctrl, s_plan;                     // the two nodes run on their own streams
    hipEvent_t ev_plan;                             // planner tick done (planner stream) -> controller stream waits
    hipEvent_t ev_ctrl;                             // readers of the other message buffer done -> planner stream waits
};

extern "C" void lpvmpc_handoff_default_config(lpvmpc_handoff_config *c) {
    if (!c) return;
    std::memset(c, 0, sizeof(*c));
    c->interp_dt = 0.033;                                       // PMAIN:257
    c->padlen = 50;                                             // PMAIN:280
    c->order = 4;                                               // PMAIN:112  signal.ellip(4, 0.01, 120, 0.125)
    const double b[5] = {0x1.36cacfc84c09bp-9, 0x1.2978f4cf2cd5cp-7, 0x1.b7d732d9307bap-7, 0x1.2978f4cf2cd5cp-7, 0x1.36cacfc84c09cp-9};
    const double a[5] = {1.0, -0x1.6596f5e4bd392p+1, 0x1.9153055f04f89p+1, -0x1.a3dfadcb54790p+0, 0x1.56daa75fe13a8p-2};
    for (int i = 0; i < 5; ++i) { c->b[i] = b[i]; c->a[i] = a[i]; }
}

static int handoff_check(lpvmpc_handle *h, int N, double dt, const lpvmpc_handoff_config *c, const char *who) {
    if (!c) return fail(h, LPVMPC_E_ARG, "%s: configuration is NULL", who);
    if (N < 4 || N > LPVMPC_HANDOFF_MAX_N || !(dt > 0) || !(c->interp_dt > 0)) return fail(h, LPVMPC_E_ARG, "%s: N=%d outside [4,%d], or dt / interp_dt <= 0", who, N, LPVMPC_HANDOFF_MAX_N);
    if (c->order < 1 || c->order > LPVMPC_MAX_FILTER_ORDER || c->padlen < 0 || c->a[0] == 0.0) return fail(h, LPVMPC_E_ARG, "%s: bad filter (order %d, padlen %d)", who, c->order, c->padlen);
    const int M = lpvmpc::handoff_length(N, dt, c->interp_dt);
    // scipy.signal.filtfilt raises ValueError here, and with it the reference's planner node (N < 34 at its rates)
    if (M <= c->padlen) return fail(h, LPVMPC_E_ARG, "%s: the resampled length %d must be greater than padlen %d", who, M, c->padlen);
    return M;
}

extern "C" int lpvmpc_handoff_length(int32_t N, double dt, const lpvmpc_handoff_config *c) {
    return handoff_check(nullptr, N, dt, c, "lpvmpc_handoff_length");
}

extern "C" int lpvmpc_handoff_operators(int32_t N, double dt, const lpvmpc_handoff_config *c, double *W, double *FW) {
    const int M = handoff_check(nullptr, N, dt, c, "lpvmpc_handoff_operators"); if (M < 0) return M;
    if (!W || !FW) return fail(nullptr, LPVMPC_E_ARG, "lpvmpc_handoff_operators: NULL output");
    std::vector<double> w, fw;
    if (!lpvmpc::handoff_operators(N, dt, c->interp_dt, c->padlen, c->order, c->b, c->a, w, fw))
        return fail(nullptr, LPVMPC_E_ARG, "lpvmpc_handoff_operators: singular spline or filter system");
    std::memcpy(W, w.data(), w.size() * 8); std::memcpy(FW, fw.data(), fw.size() * 8);
    return M;
}

extern "C" int lpvmpc_handoff_setup(lpvmpc_handle *h, const lpvmpc_handoff_config *c) {
    if (!h) return fail(nullptr, LPVMPC_E_ARG, "lpvmpc_handoff_setup: handle is NULL");
    if (h->cfg.kind != LPVMPC_KIND_PLANNER) return fail(h, LPVMPC_E_ARG, "lpvmpc_handoff_setup: planner handles only");
    const int N = h->cfg.N;
    const int M = handoff_check(h, N, h->cfg.dt, c, "lpvmpc_handoff_setup"); if (M < 0) return M;
    int rc = lpvmpc_need_track(h, "lpvmpc_handoff_setup"); if (rc) return rc;
    std::vector<double> w, fw;
    if (!lpvmpc::handoff_operators(N, h->cfg.dt, c->interp_dt, c->padlen, c->order, c->b, c->a, w, fw))
        return fail(h, LPVMPC_E_ARG, "lpvmpc_handoff_setup: singular spline or filter system");
    std::vector<double> wt((size_t)N * M), fwt((size_t)N * M);                 // device copies are transposed: [N][M]
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) { wt[(size_t)n * M + m] = w[(size_t)m * N + n]; fwt[(size_t)n * M + m] = fw[(size_t)m * N + n]; }
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    if (h->d_Wop) { (void)hipFree(h->d_Wop); (void)hipFree(h->d_FWop); h->d_Wop = h->d_FWop = nullptr; h->ho_M = 0; }
    HIP_TRY(h, hipMalloc((void **)&h->d_Wop, wt.size() * 8));
    HIP_TRY(h, hipMalloc((void **)&h->d_FWop, fwt.size() * 8));
    HIP_TRY(h, hipMemcpy(h->d_Wop, wt.data(), wt.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(h->d_FWop, fwt.data(), fwt.size() * 8, hipMemcpyHostToDevice));
    h->ho_M = M;
    return M;
}

extern "C" int lpvmpc_handoff_batch(lpvmpc_handle *h, int32_t B, const double *xPred, double *SS, double *pose, double *sig, double *refs) {
    if (h && B == 0) return LPVMPC_OK;
    int rc = lpvmpc_check_batch(h, B, "lpvmpc_handoff_batch"); if (rc) return rc;
    if (!h->d_Wop) return fail(h, LPVMPC_E_ARG, "lpvmpc_handoff_batch: call lpvmpc_handoff_setup first");
    if (!xPred || !SS || !pose || !refs) return fail(h, LPVMPC_E_ARG, "lpvmpc_handoff_batch: NULL argument");
    const size_t N = h->cfg.N, M = h->ho_M, b = B;
    hipStream_t st = h->stream;
    // workspace reuse: xPred -> d_xPred, SS -> d_curv ([cap][N+1]), pose -> d_uold ([cap][2] is too small: use d_resid [cap][4]),
    // sig -> d_states ([cap][N][5] = [cap][5][N]); refs need [B][5][M] words: a temporary
    double *d_refs = nullptr;
    HIP_TRY(h, hipMalloc((void **)&d_refs, b * 5 * M * 8));
    auto body = [&]() -> int {
        H2D(h->d_xPred, xPred, b * (N + 1) * 5 * 8); H2D(h->d_curv, SS, b * (N + 1) * 8); H2D(h->d_resid, pose, b * 3 * 8);
        HIP_TRY(h, lpvmpc::launch_plan_pose(h->d_cfg, B, h->d_xPred, h->d_curv, h->d_resid, h->d_states, st));
        HIP_TRY(h, lpvmpc::launch_resample(B, (int)N, (int)M, h->d_Wop, h->d_FWop, h->d_states, d_refs, st));
        D2H(SS, h->d_curv, b * (N + 1) * 8); D2H(pose, h->d_resid, b * 3 * 8); D2H(refs, d_refs, b * 5 * M * 8);
        if (sig) D2H(sig, h->d_states, b * 5 * N * 8);
        HIP_TRY(h, hipStreamSynchronize(st));
        return LPVMPC_OK;
    };
    rc = body();
    (void)hipFree(d_refs);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// cascade: planner (20 Hz) + trajectory-tracking controller (30 Hz) + plant, one fleet per controller handle
// ------------------------------------------------------------------------------------------------
void lpvmpc_cascade_free(lpvmpc_handle *h) {
    lpvmpc_cascade *c = h->cascade;
    if (!c) return;
    void *ptrs[] = {c->plant, c->cmd, c->local, c->ref0, c->SSc, c->lap, c->lap_tick, c->alive_ticks, c->refs, c->sig, c->SSp, c->pose, c->px0};
    for (void *p : ptrs) if (p) (void)hipFree(p);
    if (c->ev_plan) (void)hipEventDestroy(c->ev_plan);
    if (c->ev_ctrl) (void)hipEventDestroy(c->ev_ctrl);
    if (c->s_ctrl) (void)hipStreamDestroy(c->s_ctrl);
    if (c->s_plan) (void)hipStreamDestroy(c->s_plan);
    if (c->plan && c->plan->cascade_owner == h) c->plan->cascade_owner = nullptr;
    delete c;
    h->cascade = nullptr;
}

extern "C" int lpvmpc_cascade_init(lpvmpc_handle *h, lpvmpc_handle *plan, int32_t B, const double *plant0, const double *cmd0,
                                   const double *uPred0, int32_t lap0, double half_width, double slack, double plan_max_ey,
                                   int32_t q9_swap, const int32_t *n_sub, double dt_sim, double mu_sim) {
    int rc = lpvmpc_check_common(h, B, "lpvmpc_cascade_init"); if (rc) return rc;
    if (!plan) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_init: planner handle is NULL");
    if (h->cfg.kind != LPVMPC_KIND_CONTROLLER || plan->cfg.kind != LPVMPC_KIND_PLANNER)
        return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_init: needs a controller handle and a planner handle");
    if (h->cfg.device != plan->cfg.device) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_init: the two handles live on different devices");
    // the cascade's measurement kernel keeps u_old as [B][2]; a handle with steering delay reads it as [B][2 + delay]
    if (h->cfg.steering_delay != 0) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_init: the fleet engines run the reference's steeringDelay = 0 (CMAIN:49)");
    if (!plan->d_Wop) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_init: call lpvmpc_handoff_setup on the planner handle first");
    if (plan->ho_M < h->cfg.N) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_init: the planner message (%d samples) is shorter than the controller horizon", plan->ho_M);
    if (!plant0 || !cmd0 || !uPred0 || !n_sub || lap0 < 1 || !(dt_sim > 0) || n_sub[0] < 1 || n_sub[1] < 1 || n_sub[2] < 1)
        return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_init: bad argument (lap0 must be >= 1: the cascade is the racing phase)");
    rc = lpvmpc_need_track(h, "lpvmpc_cascade_init"); if (rc) return rc;
    rc = lpvmpc_check_common(plan, B, "lpvmpc_cascade_init(planner)"); if (rc) return fail(h, rc, "%s", lpvmpc_last_error(plan));
    lpvmpc_cascade_free(h);
    lpvmpc_cascade *c = new (std::nothrow) lpvmpc_cascade();
    if (!c) return fail(h, LPVMPC_E_NOMEM, "out of host memory");
    std::memset(c, 0, sizeof(*c));
    h->cascade = c;
    c->prefetch = h->cascade_prefetch;
    plan->cascade_owner = h;
    c->plan = plan; c->B = B; c->M = plan->ho_M; c->q9 = q9_swap != 0; c->hw = half_width; c->slack = slack; c->dt_sim = dt_sim; c->mu_sim = mu_sim;
    for (int i = 0; i < 3; ++i) c->n_sub[i] = n_sub[i];
    const size_t b = B, N = h->cfg.N, Np = plan->cfg.N, M = c->M;
#define ALLOC(p, n) HIP_TRY(h, hipMalloc((void **)&(p), (n)))
    ALLOC(c->plant, b * 8 * 8); ALLOC(c->cmd, b * 2 * 8); ALLOC(c->local, b * 6 * 8); ALLOC(c->ref0, b * 3 * 8); ALLOC(c->SSc, b * 8);
    ALLOC(c->lap, b * 4); ALLOC(c->lap_tick, b * 4); ALLOC(c->alive_ticks, b * 4);
    ALLOC(c->refs, 2 * b * 5 * M * 8); ALLOC(c->sig, b * 5 * Np * 8); ALLOC(c->SSp, b * (Np + 1) * 8); ALLOC(c->pose, b * 3 * 8); ALLOC(c->px0, b * 5 * 8);
#undef ALLOC
    HIP_TRY(h, hipEventCreateWithFlags(&c->ev_plan, hipEventDisableTiming));
    HIP_TRY(h, hipEventCreateWithFlags(&c->ev_ctrl, hipEventDisableTiming));
    HIP_TRY(h, hipStreamCreateWithFlags(&c->s_ctrl, hipStreamNonBlocking));
    HIP_TRY(h, hipStreamCreateWithFlags(&c->s_plan, hipStreamNonBlocking));
    hipStream_t st = h->stream;
    H2D(c->plant, plant0, b * 8 * 8); H2D(c->cmd, cmd0, b * 2 * 8); H2D(h->d_uPred, uPred0, b * N * 2 * 8);
    std::vector<int32_t> laps(b, lap0);
    H2D(c->lap, laps.data(), b * 4);
    HIP_TRY(h, hipMemsetAsync(c->lap_tick, 0, b * 4, st));
    HIP_TRY(h, hipMemsetAsync(c->alive_ticks, 0, b * 4, st));
    HIP_TRY(h, hipMemsetAsync(c->SSc, 0, b * 8, st));                            // SS = 0 at the lap event (CMAIN:258,271)
    HIP_TRY(h, hipMemsetAsync(c->SSp, 0, b * (Np + 1) * 8, st));                 // planner: SS = 0, Xlast = Ylast = Thetalast = 0 (PMAIN:72-74,124)
    HIP_TRY(h, hipMemsetAsync(c->pose, 0, b * 3 * 8, st));
    std::vector<double> hw(b, plan_max_ey);                                      // Planner.solve(..., HW)  (PMAIN:162,176)
    H2D(plan->d_maxey, hw.data(), b * 8);
    HIP_TRY(h, hipStreamSynchronize(st));
    h->state_valid_B = 0; plan->state_valid_B = 0;
    return LPVMPC_OK;
}

// one planner tick on the planner's stream (PMAIN:126-224,257-308)
static int planner_tick(lpvmpc_handle *h, lpvmpc_cascade *c) {
    lpvmpc_handle *p = c->plan;
    const int B = c->B, Np = p->cfg.N;
    hipStream_t st = c->s_plan;
    double *refs = c->refs + (size_t)(c->plan_ticks % 2) * B * 5 * c->M;
    const double *x0;
    if (c->plan_ticks == 0) {                                                    // first_it == 1: seed trajectory, measured x0
        HIP_TRY(h, lpvmpc::launch_plan_first(p->d_cfg, B, c->plant, c->hw, c->slack, c->q9, 0.2, c->px0, p->d_xlast, p->d_delta, st));
        HIP_TRY(h, lpvmpc::launch_abc(p->dev, p->d_cfg, B, p->d_xlast, p->d_delta, p->d_AB, st));
        x0 = c->px0;
    } else {                                                                     // x0 = Planner.xPred[1,:], LPVPrediction(x0, SS, uPred)
        HIP_TRY(h, hipMemcpy2DAsync(p->d_x0, 5 * 8, p->d_xPred + 5, (size_t)(Np + 1) * 5 * 8, 5 * 8, B, hipMemcpyDeviceToDevice, st));
        HIP_TRY(h, lpvmpc::launch_lpv(p->dev, p->d_cfg, B, p->d_x0, p->d_uPred, nullptr, c->SSp, 60.0, 0, p->d_states, p->d_AB, st));
        x0 = p->d_x0;
    }
    // uOld of the planner is always (0, 0): its node only appends to OldSteering / OldAccelera
    SolveArgs a{B, x0, p->d_AB, nullptr, nullptr, p->d_maxey, p->d_xPred, p->d_uPred, p->d_status, p->d_iters, p->d_polish, p->d_resid,
                p->warm_mode ? p->d_state : nullptr, (p->warm_mode && p->state_valid_B == B) ? p->warm_mode : 0, 5};
    int rc = lpvmpc_launch_solve_timed(p, a, st); if (rc) return fail(h, rc, "%s", lpvmpc_last_error(p));
    if (p->warm_mode) p->state_valid_B = B;
    HIP_TRY(h, lpvmpc::launch_plan_pose(p->d_cfg, B, p->d_xPred, c->SSp, c->pose, c->sig, st));
    HIP_TRY(h, lpvmpc::launch_resample(B, Np, c->M, p->d_Wop, p->d_FWop, c->sig, refs, st));
    c->plan_ticks++;
    return LPVMPC_OK;
}

extern "C" int lpvmpc_cascade_tick(lpvmpc_handle *h, int32_t n_ticks) {
    if (!h || !h->cascade || n_ticks < 1) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_tick: call lpvmpc_cascade_init first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    lpvmpc_cascade *c = h->cascade;
    const int B = c->B, N = h->cfg.N;
    hipStream_t st = c->s_ctrl, sp = c->s_plan;
    for (int t = 0; t < n_ticks; ++t) {
        const int k = c->ctrl_ticks;
        // 20 Hz against 30 Hz: controller tick k sees the message of planner tick j = floor(2k/3)
        const int j = (2 * k) / 3;
        const bool first_reader = k == 0 || (2 * (k - 1)) / 3 < j;
        if (c->plan_ticks <= j) {                                                // not prefetched: planner tick j right before its first reader
            if (k > 0) HIP_TRY(h, hipStreamWaitEvent(sp, c->ev_ctrl, 0));        // readers of message j - 2 (same buffer) are done
            int rc = planner_tick(h, c); if (rc) return rc;
            HIP_TRY(h, hipEventRecord(c->ev_plan, sp));
        }
        if (first_reader) HIP_TRY(h, hipStreamWaitEvent(st, c->ev_plan, 0));
        const double *refs = c->refs + (size_t)(j % 2) * B * 5 * c->M;
        const int latch = c->index == 0;
        c->index = c->index == 0 ? 1 : 0;
        HIP_TRY(h, lpvmpc::launch_tt_measure(h->d_cfg, B, c->M, k, c->plant, c->cmd, refs, latch, h->d_vel, h->d_curv, c->ref0, c->lap,
                                             c->lap_tick, c->SSc, c->local, h->d_uold, c->alive_ticks, st));
        HIP_TRY(h, hipEventRecord(c->ev_ctrl, st));
        if (c->prefetch && first_reader) {                                       // message j has its first reader: every reader of j - 1 is enqueued,
            HIP_TRY(h, hipStreamWaitEvent(sp, c->ev_ctrl, 0));                   // so planner tick j + 1 may overwrite that buffer and run beside
            int rc = planner_tick(h, c); if (rc) return rc;                      // the controller ticks that still use message j
            HIP_TRY(h, hipEventRecord(c->ev_plan, sp));
        }
        // Controller_TT.LPVPrediction(LocalState, uPred, vel_ref, curv_ref, Cf_new, LapNumber >= 1); solve(LocalState, ...)  (CMAIN:361-363)
        HIP_TRY(h, lpvmpc::launch_lpv(h->dev, h->d_cfg, B, c->local, h->d_uPred, h->d_vel, h->d_curv, 60.0, 1, h->d_states, h->d_AB, st));
        SolveArgs a{B, c->local, h->d_AB, h->d_vel, h->d_uold, nullptr, h->d_xPred, h->d_uPred, h->d_status, h->d_iters, h->d_polish, h->d_resid,
                    h->warm_mode ? h->d_state : nullptr, (h->warm_mode && h->state_valid_B == B) ? h->warm_mode : 0, 6};
        int rc = lpvmpc_launch_solve_timed(h, a, st); if (rc) return rc;
        if (h->warm_mode) h->state_valid_B = B;
        HIP_TRY(h, lpvmpc::launch_cl_command_plant(B, N, h->d_uPred, c->cmd, c->plant, lpvmpc_plant_cfg(h, c->n_sub[k % 3], c->dt_sim, c->mu_sim), st));
        c->ctrl_ticks++;
    }
    return LPVMPC_OK;
}

extern "C" int lpvmpc_cascade_read(lpvmpc_handle *h, double *plant, double *local_state, double *cmd, int32_t *ctrl_iters, int32_t *ctrl_status,
                                   int32_t *lap, int32_t *lap_tick, double *refs, double *plan_xPred, int32_t *plan_iters, int32_t *plan_status,
                                   int32_t *ticks) {
    if (!h || !h->cascade) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_read: call lpvmpc_cascade_init first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    lpvmpc_cascade *c = h->cascade;
    lpvmpc_handle *p = c->plan;
    const size_t B = c->B, Np = p->cfg.N, M = c->M;
    HIP_TRY(h, hipStreamSynchronize(c->s_plan));
    HIP_TRY(h, hipStreamSynchronize(c->s_ctrl));
    hipStream_t st = h->stream;
    if (plant) D2H(plant, c->plant, B * 8 * 8);
    if (local_state) D2H(local_state, c->local, B * 6 * 8);
    if (cmd) D2H(cmd, c->cmd, B * 2 * 8);
    if (ctrl_iters) D2H(ctrl_iters, h->d_iters, B * 4);
    if (ctrl_status) D2H(ctrl_status, h->d_status, B * 4);
    if (lap) D2H(lap, c->lap, B * 4);
    if (lap_tick) D2H(lap_tick, c->lap_tick, B * 4);
    if (refs) D2H(refs, c->refs + (size_t)((c->plan_ticks + 1) % 2) * B * 5 * M, B * 5 * M * 8);   // message of the last planner tick
    if (plan_xPred) D2H(plan_xPred, p->d_xPred, B * (Np + 1) * 5 * 8);
    if (plan_iters) D2H(plan_iters, p->d_iters, B * 4);
    if (plan_status) D2H(plan_status, p->d_status, B * 4);
    HIP_TRY(h, hipStreamSynchronize(st));
    if (ticks) { ticks[0] = c->ctrl_ticks; ticks[1] = c->plan_ticks; }
    return LPVMPC_OK;
}

extern "C" int lpvmpc_cascade_alive_ticks(lpvmpc_handle *h, int32_t *alive_ticks) {
    if (!h || !h->cascade || !alive_ticks) return fail(h, LPVMPC_E_ARG, "lpvmpc_cascade_alive_ticks: call lpvmpc_cascade_init first");
    HIP_TRY(h, hipSetDevice(h->cfg.device));
    lpvmpc_cascade *c = h->cascade;
    HIP_TRY(h, hipStreamSynchronize(c->s_plan));
    HIP_TRY(h, hipStreamSynchronize(c->s_ctrl));
    HIP_TRY(h, hipMemcpy(alive_ticks, c->alive_ticks, (size_t)c->B * 4, hipMemcpyDeviceToHost));
    return LPVMPC_OK;
}
