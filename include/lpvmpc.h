/*
 * lpvmpc.h -- C ABI of liblpvmpc.so: batched LPV-MPC / LPV-MPP solve path on MI355X (gfx950).
 *
 * The reference (euge2838/Autonomous-Racing-LPV-MPP-MPC) has no FFI for this path: its boundary is
 * two Python classes that call numpy/scipy and the third-party OSQP wheel.  Each entry point below
 * names the reference interface it replaces (paths relative to workspace/src/barc/src):
 *   CTRL = ControllerObject/PathFollowingLPVMPC.py     PLAN = PlannerObject/LPV_MPC_Planner.py
 *   UTIL = Utilities/utilities.py                      TRACK = Utilities/trackInitialization.py
 *
 * Conventions
 *   - plain C, no exceptions across the boundary; every int-returning call gives 0 on success or a
 *     negative LPVMPC_E_* code, and lpvmpc_last_error() returns a message for the last failure;
 *   - all arrays are float64, row-major ("C order"), instance-major: [B][...]; a batch call with B = 0 is a no-op
 *     that returns LPVMPC_OK (fleet engines need B >= 1);
 *   - the caller owns every buffer; nothing passed in is retained after the call returns;
 *   - one handle per (device, stream); a handle is not thread-safe, different handles are;
 *   - a process that drives SEVERAL streams should export GPU_MAX_HW_QUEUES (the HIP runtime's number of hardware queues, default 4, read
 *     once when the runtime initialises) before anything touches HIP: streams that share a hardware queue run in order, so a launch
 *     holding a slow instance blocks its queue's other streams (configs[1]: 1.3 M solves/s with 8 queues, 2.4 M with 16; with straggler
 *     deferral four streams on four queues reach 4.1 M; beyond ~20 queues the hardware time-slices).  The library does not set it: it is
 *     the process's to choose (bench.py sets 16);
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     LPVMPC_E_NODEVICE.
 *
 * Decision vector per instance: z = [x_0 .. x_N, u_0 .. u_{N-1}]  (CTRL:479-492, PLAN:434-445)
 *   controller nx = 6  [vx vy wz epsi s ey]   (CTRL:712-718)
 *   planner    nx = 5  [vx vy wz ey epsi]     (PLAN:288-292)
 *   inputs     nu = 2  [delta a]
 */
#ifndef LPVMPC_H
#define LPVMPC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LPVMPC_VERSION 200            /* 0.2.0: round 5 removed the lpvmpc_lane_* exports (an ABI break), round 6 adds lpvmpc_defer_stats and kernel_variant 9 */

#define LPVMPC_KIND_CONTROLLER 0      /* PathFollowingLPV_MPC  (CTRL:30-258) */
#define LPVMPC_KIND_PLANNER    1      /* LPV_MPC_Planner       (PLAN:29-320) */

#define LPVMPC_MAX_TRACK_ROWS 16
#define LPVMPC_MAX_N          52      /* (N+1) stages x 3008 B of LDS per instance must fit 160 KiB (run-time-horizon kernel) */

/* error codes */
#define LPVMPC_OK            0
#define LPVMPC_E_ARG        -1
#define LPVMPC_E_NODEVICE   -2
#define LPVMPC_E_HIP        -3
#define LPVMPC_E_NOMEM      -4

/* per-instance solver status, numerically equal to OSQP's status_val (CTRL:320-324, PLAN:214-216) */
#define LPVMPC_SOLVED                        1
#define LPVMPC_SOLVED_INACCURATE             2
#define LPVMPC_PRIMAL_INFEASIBLE_INACCURATE  3
#define LPVMPC_DUAL_INFEASIBLE_INACCURATE    4
#define LPVMPC_MAX_ITER_REACHED             -2
#define LPVMPC_PRIMAL_INFEASIBLE            -3
#define LPVMPC_DUAL_INFEASIBLE              -4
#define LPVMPC_NON_CVX                      -7
#define LPVMPC_UNSOLVED                    -10      /* also: non-finite input data (NaN / Inf in x0, A, B, vel_ref, uOld,
                                                      max_ey) -- no iteration is run, xPred / uPred are NaN, iters = 0 */
#define LPVMPC_PENDING                     -11      /* straggler deferral only (option "defer_after"): the instance was parked by the
                                                      * main launch; its resume launch has not written the final status yet */

typedef struct lpvmpc_config {
    int32_t kind;            /* LPVMPC_KIND_* */
    int32_t N;               /* horizon (ctor arg N, CTRL:35 / PLAN:34) */
    int32_t device;          /* HIP device ordinal */
    int32_t steering_delay;  /* controller: ctor arg steeringDelay (CTRL:35,68-71): the first steering_delay stages get an equality
                              * row delta_i = OldSteering[i+1] (CTRL:518-527); 0 (the reference's mains, CMAIN:49) .. 8.  Planner: 0 */
    double  dt;              /* sample time (ctor arg dt) */
    /* vehicle parameters the reference reads from ROS (CTRL:38-48, PLAN:70-82) */
    double  lf, lr, m, Iz, Cf, Cr, mu;
    double  max_vel, min_vel;
    /* weights (ctor args Q, R, dR, L_cf); Q is nx*nx row-major in the first nx*nx slots */
    double  Q[36];
    double  R[4];
    double  dR[2];
    double  L_cf[6];         /* planner only (PLAN:163) */
    /* hard-coded limits of the reference, exposed: controller CTRL:334-348, planner PLAN:173-177 */
    double  ctrl_vx_min;     /* 0.01  */
    double  ctrl_delta_max;  /* 0.249 */
    double  ctrl_a_max;      /* 4.0   */
    double  ctrl_a_min_abs;  /* 1.0   (a >= -1.0) */
    double  plan_xmin[5];    /* [min_vel -1 -2 -max_ey -0.8]; slots 0 and 3 are overwritten from min_vel / max_ey */
    double  plan_xmax[5];
    double  plan_umin[2];    /* [-0.249 -0.7] */
    double  plan_umax[2];    /* [ 0.249  2.0] */
    /* OSQP settings (defaults of the 0.6.x series; the reference passes only polish=True) */
    double  rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
    double  polish_delta, adaptive_rho_tolerance;
    int32_t max_iter, check_termination, scaling, adaptive_rho, adaptive_rho_interval;
    int32_t polish, polish_refine_iter, reserved1;
    /* track table = Map.PointAndTangent (TRACK:88-202): rows [x y psi cum_s seg_len curvature] */
    int32_t track_rows;
    int32_t reserved2;
    double  track[LPVMPC_MAX_TRACK_ROWS * 6];
} lpvmpc_config;

typedef struct lpvmpc_handle lpvmpc_handle;

/* library / ABI version (LPVMPC_VERSION). */
int lpvmpc_version(void);

/* Fill *cfg with the reference's launch-file and hard-coded defaults for `kind`
 * (MAIN_LAUNCH.launch:5-11,40-44; controllerMain.py:139-150; plannerMain.py:96-99). */
void lpvmpc_default_config(int32_t kind, lpvmpc_config *cfg);

/* Replaces the constructors PathFollowingLPV_MPC.__init__ (CTRL:35-84) and
 * LPV_MPC_Planner.__init__ (PLAN:34-82).  Returns NULL on failure (see lpvmpc_last_error(NULL)). */
lpvmpc_handle *lpvmpc_create(const lpvmpc_config *cfg);
void lpvmpc_destroy(lpvmpc_handle *h);
const char *lpvmpc_last_error(const lpvmpc_handle *h);
/* LPVMPC_E_* code of the last failure on this thread (0 when none); tells why lpvmpc_create returned NULL. */
int lpvmpc_last_error_code(void);

/* Runtime options.  "kernel_variant": 0 = fastest instantiation for (kind, N) (default), 1 = run-time-horizon
 * kernel (factor tiles in LDS, any N), 2 = compile-time horizon with ONE wavefront per instance (where it
 * exists; the default for N = 20 / 30 / 40 uses two wavefronts and a two-sided elimination).
 * "force_generic_kernel" (0/1) is shorthand for variants 0 / 1.  Used by the tests to cross-check the kernels.
 * "warm_start": 0 = every solve starts from x = z = y = 0 like the reference (fresh OSQP object per call,
 * CTRL:302,316 / PLAN:204-208; default); 1 = start from the previous solve's (x, y) of the same handle and
 * batch size; 2 = the same shifted by one stage (receding horizon).  Opt-in, changes iteration counts, not optima.
 * "cascade_prefetch" (0/1, default 1): read by lpvmpc_cascade_init on the controller handle, see there.
 * "kernel_variant" 3 = the DPP two-wavefront kernels of round 1 for the controller at N = 20 and the planner at N = 20 / 30 / 40 (their
 * defaults run the KKT sweeps and the factorisation on the matrix cores; the planner at N = 30 / 40 with FOUR wavefronts per instance, the
 * two elimination chains relayed over two wavefronts each -- at N = 30 for launches that leave compute units free and for handles with
 * straggler deferral, while plain launches of 512 instances or more take the two-wavefront form of the same arithmetic: every output word
 * is the same either way); 4 = the planner N = 30 kernel with two wavefronts and MFMA sweeps whatever the batch; 5 = the planner N = 30 DPP kernel
 * with every vector in LDS (two instances per CU); 6 = the planner N = 40 kernel with two wavefronts and MFMA sweeps whatever the batch;
 * 7 = round 3's default at N = 30: the DPP kernel with its three equilibration vectors in global memory (three instances per CU) for
 * batches beyond 512 instances without deferral, variant 5 otherwise; 8 = the four-wavefront planner kernels whatever the batch.  One
 * arithmetic per (kind, N) by default: an instance's result does not depend on the batch it is solved in.
 * 9 = the LATENCY form of the N = 20 kernels, controller (steeringDelay 0) and planner, on handles without straggler deferral (every
 * other handle and shape: as 0):
 * the two elimination chains relayed over FOUR wavefronts, the arithmetic of the default kernel step for step.  An instance that has its
 * compute unit to itself (batches up to 256) finishes 6 % sooner, a full chip gains nothing.  Opt-in: the block-wide sums of the N = 20
 * kernels (the cost normalisation's mean, the infeasibility certificates' sums, the objective) associate by the number of wavefronts,
 * so between variants 0 and 9 a word may differ in its last bits where such a sum decides (statuses and iteration counts are equal,
 * solutions to 5e-6: tests/test_gpu_parity.py; every word equal on the batches tried, where the cost normalisation is decided by the
 * linear term's maximum; planner N = 20: -7 % for a lone instance, -8 % at 64 ... 256).  The drop-in classes (one vehicle per handle)
 * select it.
 * "defer_after" (iterations, 0 = off, default): STRAGGLER DEFERRAL for lpvmpc_solve_batch_dev.  One OSQP solve in a thousand
 * needs thousands of ADMM iterations where the typical one needs 50; a launch lasts as long as its slowest instance, so those
 * few hold the caller's stream for milliseconds.  With defer_after = K an instance that is still unsolved at a termination
 * check with iter >= K is parked (status LPVMPC_PENDING, whole solver state saved) and the launch ends.  Every deferred call
 * is followed, on the same stream, by a resume pass of the same kernel that continues everything parked on the handle -- from
 * this call and from earlier ones -- for "defer_budget" more iterations (default 200; 0 = to completion) and parks again what
 * is still unsolved.  With "defer_tail" 0 the results are bit-identical to the plain call (a restored instance re-factors K
 * from its saved state); with the default "defer_tail" 1 the closing passes use the tail kernel: equal to round-off, see there.
 * Completion contract: an instance's outputs are final when its status is no longer LPVMPC_PENDING; lpvmpc_join(h, stream)
 * enqueues the pass that finishes whatever is still parked, so work behind it in `stream` sees complete outputs.  Until then
 * the output buffers of the deferred calls must stay valid and must not be reused for other data.  The synchronous host-array
 * call lpvmpc_solve_batch joins by itself before it copies the outputs back (it never returns LPVMPC_PENDING).
 * "defer_budget" -1: no pass behind a deferred call at all; the parked instances wait for lpvmpc_join (what a caller that joins
 * after every call wants: one batch, then the tail kernel; lpvmpc_solve_batch does this by itself).
 * "defer_pool" (entries, 0 = max(64, B / 8), default): capacity of each of the two pools.  ADMISSION IS ORDERED BY AGE (K = defer_after):
 * instances with fewer than 2 K iterations may take three quarters of a pool, those between 2 K and 4 K an eighth of their own, and an
 * instance beyond 4 K takes any free entry (at least the last eighth).  An instance that finds its class's share full is not parked at
 * that check; it goes on iterating inside the launch that holds it and asks again at its next check (25 iterations older), so the few
 * many-thousand-iteration instances of a batch always get parked, however many nearly-done ones a small K sends to the pool first
 * (before round 6 admission was first come, first served: with K = 50 two fifths of a controller batch filled the pool at iteration 50,
 * the 4000-iteration instance stayed in the main launch to its end and a burst ran at half the rate of K = 75 .. 125; now K = 50 .. 125
 * are within 3 % of each other).  K is still a cost parameter: every parked instance is restored and re-factored by the pass that
 * continues it, so park what is rare -- choose K near the point where ~99 % of the instances are done (100 for the controller workloads
 * here; K = 25 parks most of a batch once and runs a burst at three quarters of the rate).  lpvmpc_defer_stats tells how many requests
 * were refused, i.e. whether "defer_pool" should grow.
 * "defer_tail" (0 | 1, default 1): the passes that run parked instances to completion (lpvmpc_join, the synchronous entry
 * points, "defer_budget" 0) use the whole-CU tail kernel where one exists for the handle (controller or planner, N = 20): a 512-thread
 * workgroup per instance that applies K^-1 as a dense matrix held in registers, runs two phases per ADMM iteration and evaluates the
 * termination checks beside the iterations (round 5: 1.0 us per iteration against 1.9 us for an instance that has the GPU to
 * itself).  Handles with steering_delay > 0 keep the two-wavefront kernel for these passes.  Statuses, iteration counts and polish
 * flags have been OBSERVED equal to the other kernel's on every instance compared so far (196 608 over four tracks,
 * profiles/r05_tail_parity_sweep.txt; the seeds of tests/test_gpu_deferral.py are regression fixtures for this build) -- an
 * observation, not a guarantee: the two kernels round differently, and a termination test that is decided by round-off can move by
 * one check (25 iterations) on another toolchain or device.  Solutions agree to round-off (1e-7 polished, 1e-6 for an un-polished
 * iterate; observed 3.1e-9 / 1.1e-7), so bit-identity with the plain call holds with 0 only.
 *
 * FLOAT TOLERANCE of the QP stage against the reference algorithm (the CPU oracle on identical data; tests/_tolerance.py,
 * DESIGN.md section 2): status and iteration count equal -- except a run that ends at max_iter, where OSQP's 10 eps "solved
 * inaccurate" test is decided by round-off (MAX_ITER_REACHED <-> SOLVED_INACCURATE, same iteration count) -- and xPred / uPred in one of
 * three classes: (A) polished: 1e-6; (B) un-polished, converged: 2e-4 (observed <= 1e-6); (C) ran to the max_iter cap (an unconverged
 * ADMM iterate, which OSQP guarantees nothing for and the reference uses as it comes): same iteration count, |du| <= 2e-2 (observed
 * <= 1.01e-2 on 42 of 110 202 instances, all of them planner QPs at 4000 iterations).  Class C carries NO objective bound: two of the 42
 * sit 1.8e-2 / 2.8e-4 (relative) from the oracle's objective -- MAX_ITER_REACHED points, for which OSQP promises nothing.
 * (D) no solution on either side (PRIMAL / DUAL INFEASIBLE: NaN outputs, equal statuses): the certificate of a diverging iterate may fire
 * one termination check (25 iterations) apart -- the oracle itself does under its two KKT elimination orders (1 of 110 202: Euge_Track,
 * planner N = 40: device 50, batch oracle 75, oracle with the other order 50).  The reference discards such a tick either way. */
int lpvmpc_set_option(lpvmpc_handle *h, const char *name, int32_t value);
/* Straggler deferral (see "defer_after"): enqueues on `stream` (a hipStream_t; ordered behind the stream of the handle's last
 * deferred call if it is another one) the resume pass that runs every parked instance to completion.  No-op without deferral. */
int lpvmpc_join(lpvmpc_handle *h, void *stream);
/* Straggler deferral counters of the handle since its first deferred call: *parked = instances parked (every parking counts, also a
 * re-parking by a bounded resume pass), *refused = parking requests turned down because the pool share of the instance's age class was
 * full (see "defer_pool": the instance went on in its launch and asked again later).  Waits for the stream of the handle's last
 * deferred call, so every launch enqueued so far is counted.  Either pointer may be NULL. */
int lpvmpc_defer_stats(lpvmpc_handle *h, int64_t *parked, int64_t *refused);
/* Like lpvmpc_kernel_time_stats (below) for the resume launches of the straggler deferral. */
int lpvmpc_resume_time_stats(lpvmpc_handle *h, double *total_ms, int32_t *count);

/* (Round 4 shipped an experimental "long-runner lane" here -- lpvmpc_lane_*: reserved compute units for the whole-CU tail kernel.
 * It lost on every configuration measured (profiles/r04_lane_ab.txt: the driver's burst 13.3 -> 14.1-16.6 ms, the default run
 * 3.17 -> 2.5-2.7 M solves/s) and was removed from the library in round 5; docs/HISTORY.md keeps the design notes.) */

/* Pre-size the device workspace for batches up to B (otherwise grown on demand). */
int lpvmpc_reserve(lpvmpc_handle *h, int32_t B);

/*
 * LPV evaluation + horizon roll-out.  Replaces LPVPrediction (CTRL:166-258 / PLAN:242-320).
 *   x0       [B][nx]
 *   u_prev   [B][N][2]       previous input prediction (uPred)
 *   vel_ref  [B][N+1]        controller: vx scheduling (CTRL:200); entry N unused here.  NULL for planner
 *   curv_s   controller: curv_ref [B][N] used when lap != 0 (CTRL:196-198), may be NULL when lap == 0;
 *            planner:    SS [B][N+1]  (PLAN:270-271)
 *   cf_new   controller: Cf = Cr = cf_new (CTRL:172-173); ignored for the planner
 *   lap      controller: LapNumber (0 -> curvature from the map at the rolled-out s)
 * outputs (any may be NULL):
 *   states   [B][N][nx]      STATES_vec
 *   A        [B][N][nx][nx]  Atv      Bm [B][N][nx][2]  Btv       (Ctv is identically zero, CTRL:236-241)
 */
int lpvmpc_lpv_batch(lpvmpc_handle *h, int32_t B, const double *x0, const double *u_prev,
                     const double *vel_ref, const double *curv_s, double cf_new, int32_t lap,
                     double *states, double *A, double *Bm);

/* Seed-mode linearisation along a given trajectory.  Replaces _EstimateABC (CTRL:732-809 / PLAN:519-591).
 *   xlast [B][N][6]  controller columns [vx vy wz epsi s ey]; planner columns [vx vy wz ey epsi s]
 *   delta [B][N]     steering angle per stage */
int lpvmpc_estimate_abc_batch(lpvmpc_handle *h, int32_t B, const double *xlast, const double *delta,
                              double *A, double *Bm);

/*
 * QP build + OSQP-ADMM solve with caller-supplied LPV matrices.  Replaces
 * PathFollowingLPV_MPC.solve(x0, ., uPred, ., vel_ref, A_L, B_L, C_L, .) (CTRL:89-162 incl. _buildMatEqConst,
 * _buildMatCost, _buildMatIneqConst, osqp_solve_qp) and LPV_MPC_Planner.solve (PLAN:86-236).
 *   x0      [B][nx]
 *   A       [B][N][nx][nx]   Bm [B][N][nx][2]
 *   vel_ref [B][N+1]   controller tracking reference: entries 0..N-1 = vel_ref[i], entry N = vel_ref[-1]
 *                      (CTRL:434-438); NULL for the planner
 *   u_old   [B][2 + steering_delay]  [OldSteering[0], OldAccelera[0], OldSteering[1 .. steering_delay]] (CTRL:395, 523;
 *                      PLAN:114); NULL = zeros
 *   max_ey  [B]        planner lateral bound (solve arg max_ey, PLAN:176-177); NULL for the controller
 * outputs:
 *   xPred [B][N+1][nx], uPred [B][N][2]  (NaN for instances without a solution, as OSQP returns)
 *   status [B], iters [B]  (may be NULL)
 *   resid  [B][4] = {pri_res, dua_res, obj_val, rho_final}  (may be NULL)
 *   polish [B]    = OSQP status_polish (1 accepted, -1 rejected, 0 not run)  (may be NULL)
 */
int lpvmpc_solve_batch_AB(lpvmpc_handle *h, int32_t B, const double *x0, const double *A, const double *Bm,
                          const double *vel_ref, const double *u_old, const double *max_ey,
                          double *xPred, double *uPred, int32_t *status, int32_t *iters, double *resid,
                          int32_t *polish);

/* Fused tick: lpvmpc_lpv_batch followed by lpvmpc_solve_batch_AB with x0 as the initial state
 * (the call pair controllerMain.py:361-363 / plannerMain.py:175-176), without the host round trip. */
int lpvmpc_solve_batch(lpvmpc_handle *h, int32_t B, const double *x0, const double *u_prev,
                       const double *vel_ref, const double *curv_s, const double *u_old, const double *max_ey,
                       double cf_new, int32_t lap,
                       double *xPred, double *uPred, int32_t *status, int32_t *iters, double *resid,
                       int32_t *polish);

/* Same as lpvmpc_solve_batch but every pointer is a DEVICE pointer and the work is enqueued on `stream`
 * (a hipStream_t; NULL = default stream) without synchronising.  Used by the closed-loop / bench paths. */
int lpvmpc_solve_batch_dev(lpvmpc_handle *h, int32_t B, const double *x0, const double *u_prev,
                           const double *vel_ref, const double *curv_s, const double *u_old, const double *max_ey,
                           double cf_new, int32_t lap,
                           double *xPred, double *uPred, int32_t *status, int32_t *iters, double *resid,
                           int32_t *polish, void *stream);

/* Enable (1) / disable (0) HIP-event timing of the solve kernel: while enabled every launch of the
 * ADMM solve kernel is bracketed by an event pair recorded on the launch stream (no host sync).
 * Enabling resets the statistics.  (bench.py roofline leg.) */
int lpvmpc_set_timing(lpvmpc_handle *h, int32_t on);

/* Sum of the solve-kernel durations (ms) and the number of launches recorded since timing was enabled
 * (at most the last 1024).  Synchronises with the recorded events. */
int lpvmpc_kernel_time_stats(lpvmpc_handle *h, double *total_ms, int32_t *count);

/* Duration (ms) of the most recent timed solve-kernel launch, negative if none. */
double lpvmpc_last_kernel_ms(lpvmpc_handle *h);

/* ---------------------------------------------------------------------------------------------------
 * "Next row" f1 of the hot-path scope table: the caller-side pieces a closed-loop run needs, batched.
 * ------------------------------------------------------------------------------------------------- */

/* Map.getLocalPosition (TRACK:283-383): xy_psi [B][3] -> out [B][4] = {s, ey, epsi, inside}; off the track the
 * reference's sentinels {10000, 10000, 10000, 0}.  half_width / slack = Map.halfWidth / Map.slack. */
int lpvmpc_local_position_batch(lpvmpc_handle *h, int32_t B, const double *xy_psi, double half_width, double slack,
                                double *out);

/* Map.getGlobalPosition (TRACK:205-262): s_ey [B][2] -> out [B][3] = {x, y, theta}. */
int lpvmpc_global_position_batch(lpvmpc_handle *h, int32_t B, const double *s_ey, double *out);

/* n_sub steps of Simulator.f (vehicleSimulator.py:164-199; linear tyres Fy = 60 alpha) on
 * state [B][8] = {x, y, vx, vy, ax, ay, yaw, psiDot} (in/out) under the constant input u [B][2] = {a, delta}.
 * lf, lr, m, Iz come from the handle's configuration; mu_sim = simulator/mu, dt_sim = simulator/dt. */
int lpvmpc_plant_step_batch(lpvmpc_handle *h, int32_t B, double *state, const double *u, int32_t n_sub, double dt_sim,
                            double mu_sim);

/* Closed-loop fleet of B vehicles under the LPV-MPC controller in the lap-0 path-tracking branch of
 * controllerMain.py (:179-190 measurement incl. quirk Q9 when q9_swap != 0, :289-298 last command as uOld,
 * :310-315 nine seed ticks, :325-331 LPV prediction with x0 = first rolled-out state, :381-386 command), with the
 * plant advanced n_sub simulator steps per control tick.  Everything stays on the device; lpvmpc_cl_tick enqueues
 * n_ticks control ticks without synchronising, lpvmpc_cl_read synchronises and copies the current fleet state
 * (any pointer may be NULL): plant [B][8], local_state [B][6], cmd [B][2] = {servo, motor}, iters / status [B]. */
int lpvmpc_cl_init(lpvmpc_handle *h, int32_t B, const double *plant0, double half_width, double slack, int32_t q9_swap,
                   int32_t n_sub, double dt_sim, double mu_sim);
int lpvmpc_cl_tick(lpvmpc_handle *h, int32_t n_ticks);
int lpvmpc_cl_read(lpvmpc_handle *h, double *plant, double *local_state, double *cmd, int32_t *iters, int32_t *status);
/* A handle that runs a fleet (lpvmpc_cl_init) or a cascade (lpvmpc_cascade_init, the controller handle and its planner
 * handle) keeps the fleet's receding-horizon state in its workspace between ticks: the stand-alone batch calls
 * (lpvmpc_solve_batch*, lpvmpc_lpv_batch, lpvmpc_estimate_abc_batch, lpvmpc_*_position_batch, lpvmpc_plant_step_batch,
 * lpvmpc_handoff_batch) on such a handle fail with LPVMPC_E_ARG instead of overwriting it -- use a second handle.
 * lpvmpc_cl_release ends the fleet / cascade of the handle (waits for its queued ticks, frees the fleet buffers); the handle
 * then accepts batch calls again. */
int lpvmpc_cl_release(lpvmpc_handle *h);

/* ---------------------------------------------------------------------------------------------------------------
 * Planner -> controller reference hand-off.  Replaces the post-processing in the planner node
 *   plannerMain.py:201-224   s integration along the planned states, centre-line pose (Map.getGlobalPosition),
 *                            xp / yp / yaw reconstruction, vel = vx, curv = wz / vx
 *   plannerMain.py:257-280   scipy interp1d(kind='cubic') from N samples at dt to round(N dt / interp_dt) samples,
 *                            scipy.signal.filtfilt(b, a, curvature, padlen)
 *   plannerMain.py:303-308   the five My_Planning arrays x_d, y_d, psi_d, vx_d, curv_d (barc/msg/My_Planning.msg:1-6)
 * Resampling and filtering are linear in the N samples, so they are applied as two dense operators built on the host:
 * W (cubic not-a-knot spline evaluation) and FW = filtfilt o W. */
#define LPVMPC_MAX_FILTER_ORDER 8
typedef struct lpvmpc_handoff_config {
    double  interp_dt;                          /* 0.033                                   PMAIN:257 */
    int32_t padlen;                             /* 50                                      PMAIN:280 */
    int32_t order;                              /* 4: signal.ellip(4, 0.01, 120, 0.125)    PMAIN:112 */
    double  b[LPVMPC_MAX_FILTER_ORDER + 1];     /* numerator, order + 1 entries used */
    double  a[LPVMPC_MAX_FILTER_ORDER + 1];     /* denominator */
} lpvmpc_handoff_config;

/* the reference's values, including the coefficients of its elliptic filter */
void lpvmpc_handoff_default_config(lpvmpc_handoff_config *cfg);

/* Number of resampled points M = round(N dt / interp_dt) (PMAIN:259), or LPVMPC_E_ARG when the configuration cannot
 * work: like scipy's filtfilt (and so the reference's node) this refuses M <= padlen, i.e. N < 34 at the reference's rates. */
int lpvmpc_handoff_length(int32_t N, double dt, const lpvmpc_handoff_config *cfg);

/* Host-only (no device needed): the operators W and FW, each [M][N] row-major; returns M. */
int lpvmpc_handoff_operators(int32_t N, double dt, const lpvmpc_handoff_config *cfg, double *W, double *FW);

/* Build the operators for a PLANNER handle's (N, dt) and keep them on its device; returns M. */
int lpvmpc_handoff_setup(lpvmpc_handle *planner, const lpvmpc_handoff_config *cfg);

/* One hand-off for B planner solutions: xPred [B][N+1][5]; SS [B][N+1] and pose [B][3] = {Xlast, Ylast, Thetalast}
 * are the node's carried state (in/out: SS is re-integrated and SS[0] = SS[1], pose = centre-line pose of stage 1);
 * sig [B][5][N] (may be NULL) = xp, yp, yaw, vel, curv at the planner's rate; refs [B][5][M] = the My_Planning arrays. */
int lpvmpc_handoff_batch(lpvmpc_handle *planner, int32_t B, const double *xPred, double *SS, double *pose, double *sig,
                         double *refs);

/* Planner + controller + plant cascade for a fleet of B vehicles in the racing phase (LapNumber >= 1), resident on the
 * device.  `ctrl` is a controller handle with the trajectory-tracking tuning (Controller_TT, CMAIN:142-150), `planner` a
 * planner handle on which lpvmpc_handoff_setup has been called.  Per controller tick (30 Hz):
 *   - planner ticks 0 .. floor(2k/3) have run before controller tick k (20 Hz node; PMAIN:126-224,257-308: first tick
 *     from the measured state with the seed trajectory of PMAIN:465-505, later ticks open loop from xPred[1]);
 *   - measurement of the LapNumber >= 1 branch (CMAIN:176-182,198-248): yaw - 2 pi LapNumber wrapped, reference windows
 *     [0:N] re-read from the latest message on every second tick only (`index` toggle), Body_Frame_Errors (CMAIN:495-506)
 *     with dead-reckoned s, racing lap counter (CMAIN:268-272); uOld = last command (CMAIN:289-298);
 *   - Controller_TT.LPVPrediction + solve from the measured state (CMAIN:361-363); command = uPred[0] (CMAIN:381-386);
 *   - the plant advances n_sub[k % 3] steps of Simulator.f (7, 7, 6 steps of 5 ms = 100 ms per 3 ticks).
 * plant0 [B][8], cmd0 [B][2] = {servo, motor} and uPred0 [B][N][2] (Controller_TT.uPred = Controller.uPred, CMAIN:336)
 * describe the fleet at the lap event; lap0 >= 1.  half_width / slack: the map's, used by Map.getLocalPosition;
 * plan_max_ey: the max_ey argument of the planner's solve (the ROS parameter /TrajectoryPlanner/halfWidth).  q9_swap: the planner's first x0 takes the map's (ey, epsi) in the
 * (epsi, ey) slots as PMAIN:141 assigns them.  The two nodes run on their own HIP streams; with option
 * "cascade_prefetch" = 1 (default; set on `ctrl` before init) a planner tick is enqueued as soon as its message buffer is
 * free, so that it overlaps the controller ticks that still use the previous message -- results do not change.
 * lpvmpc_cascade_tick enqueues n_ticks controller ticks without synchronising; lpvmpc_cascade_read synchronises and copies
 * (any pointer may be NULL): plant [B][8], local_state [B][6], cmd [B][2], ctrl_iters / ctrl_status [B], lap / lap_tick [B]
 * (lap counter and the controller tick of the last lap event), refs [B][5][M] and plan_xPred [B][Np+1][5] / plan_iters /
 * plan_status [B] of the most recent planner tick, ticks [2] = {controller ticks, planner ticks} enqueued so far.
 * The planner handle must outlive the cascade (it ends with the controller handle, or with the next lpvmpc_cascade_init). */
int lpvmpc_cascade_init(lpvmpc_handle *ctrl, lpvmpc_handle *planner, int32_t B, const double *plant0, const double *cmd0,
                        const double *uPred0, int32_t lap0, double half_width, double slack, double plan_max_ey,
                        int32_t q9_swap, const int32_t *n_sub, double dt_sim, double mu_sim);
int lpvmpc_cascade_tick(lpvmpc_handle *ctrl, int32_t n_ticks);
int lpvmpc_cascade_read(lpvmpc_handle *ctrl, double *plant, double *local_state, double *cmd, int32_t *ctrl_iters,
                        int32_t *ctrl_status, int32_t *lap, int32_t *lap_tick, double *refs, double *plan_xPred,
                        int32_t *plan_iters, int32_t *plan_status, int32_t *ticks);
/* alive_ticks [B]: controller ticks each vehicle has entered with a finite plant state since lpvmpc_cascade_init (a vehicle whose
 * planner QP went primal infeasible carries NaN from then on and costs no iterations: the sum over the fleet is the number of
 * vehicle-ticks that did work).  Synchronises like lpvmpc_cascade_read. */
int lpvmpc_cascade_alive_ticks(lpvmpc_handle *ctrl, int32_t *alive_ticks);

#ifdef __cplusplus
}
#endif
#endif /* LPVMPC_H */
