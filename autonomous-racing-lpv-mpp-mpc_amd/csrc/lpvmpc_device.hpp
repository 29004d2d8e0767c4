// lpvmpc_device.hpp -- device-side configuration shared by the kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lpvmpc {

constexpr int kMaxSeg = 16;   // rows of the track table kept in the kernel argument block
constexpr int kTS = 72;       // LDS stride (doubles) of one 8x8 tile: 64 + 8 pad => stage tiles start 16 banks apart

// OSQP constants (0.6.x), see oracle/osqp_ref.c header
constexpr double kInfty = 1e30;
constexpr double kRhoMin = 1e-6, kRhoMax = 1e6, kRhoEqOverIneq = 1e3, kRhoTol = 1e-4;
constexpr double kMinScaling = 1e-4, kMaxScaling = 1e4;

struct DevCfg {
    int32_t kind, N, track_rows, max_iter;
    int32_t check_termination, scaling, adaptive_rho, adaptive_rho_interval;
    int32_t polish, polish_refine_iter, steering_delay, pad1;
    double dt, lf, lr, m, Iz, Cf, Cr, mu, max_vel, min_vel;
    double Q[36], R[4], dR[2], Lcf[6];
    double box_lo[8], box_hi[8];   // per-stage box rows, unscaled (planner row 3 = ey is per instance)
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, delta, rho_tol;
    double track[kMaxSeg * 6];
};

// Piecewise-constant curvature lookup, reference UTIL:31-50 (Curvature).  Same wrap (repeated subtraction of the track
// length, UTIL:36-40) and the same comparisons (s >= start && s < start + len) in float64.  Where the reference raises
// (UTIL:44-48 finds no segment: s < 0, s exactly on the end of the closing segment) or would spin (s not finite, or more than
// kMaxWrapLaps track lengths ahead: a diverged roll-out) the lookup returns NaN: the NaN travels through the LPV blocks into
// the solve kernel, which runs no iteration on non-finite data and reports LPVMPC_UNSOLVED with NaN outputs -- the caller
// sees the failure instead of a silently substituted segment.
constexpr int kMaxWrapLaps = 4096;
__device__ inline double wrap_track_s(double s, double L) {
    if (!(s <= L * (double)kMaxWrapLaps)) return __builtin_nan("");     // also catches NaN and +Inf
    for (int it = 0; it < kMaxWrapLaps && s > L; ++it) s -= L;
    return s;
}
__device__ inline double track_curvature(const DevCfg &c, double s) {
    const int rows = c.track_rows;
    const double L = c.track[(rows - 1) * 6 + 3] + c.track[(rows - 1) * 6 + 4];
    s = wrap_track_s(s, L);
    for (int i = 0; i < rows; ++i) {
        const double st = c.track[i * 6 + 3], ln = c.track[i * 6 + 4];
        if (s >= st && s < st + ln) return c.track[i * 6 + 5];
    }
    return __builtin_nan("");
}

// arguments of the solve kernel (device pointers)
struct SolveArgs {
    int B;
    const double *x0;       // [B][x0_stride], first NX used
    const double *AB;       // [B][N][NX][NX+2]
    const double *vel_ref;  // [B][N+1]   controller
    const double *u_old;    // [B][2 + steering_delay] or null
    const double *max_ey;   // [B]        planner
    double *xPred;          // [B][N+1][NX]
    double *uPred;          // [B][N][2]
    int32_t *status, *iters, *polish;
    double *resid;          // [B][4]
    double *state;          // [B][3][8(N+1)] unscaled x, y(dynamics rows), y(box rows) of the previous solve, or null
    int warm;               // 0 cold start (reference behaviour), 1 warm start from state, 2 same shifted by one stage
    int x0_stride;          // doubles between consecutive instances' x0 (NX, or N*NX when x0 = first rolled-out state)
    // straggler deferral (lpvmpc_set_option "defer_after"): an instance that is still unsolved at a termination check after its
    // iteration budget is PARKED -- its whole LDS image, loop state and output pointers go to a pool entry -- and the workgroup
    // ends; a later launch of the same kernel with resume = 1 (one workgroup per entry of pool_in) restores it, re-factors K
    // with the saved rho (a pure function of the saved image, so the iterates continue bit for bit) and runs it for another
    // budget (parking it again in pool) or to completion (defer_after = 0).
    int defer_after;        // main launch: park at the first check with iter >= defer_after; resume launch: after that many more
                            // iterations; 0: run to completion
    int resume;             // 1: continue the parked instances of pool_in (blockIdx.x = entry)
    double *pool;           // [pool_cap][pool_stride] entries written by this launch
    int32_t *pool_count;    // [4]: slots requested so far in pool (may exceed pool_cap: the surplus instances were not parked), [1] see pool_in_count,
                            // [2] / [3] requests of the young / middle age class (admission by age: Solver::park_slot)
    unsigned long long *defer_stats;   // [2]: instances parked / parking attempts refused (pool share of the instance's age class full) by launches of this handle
    const double *pool_in;  // resume: entries to continue
    int32_t *pool_in_count; // [4]: entries, the number of resume workgroups that have read it, the class counters (the last reader clears all:
                            // the pool is empty again when the pass ends, without a separate memset launch)
    int pool_cap, pool_stride;
    // planner N = 30, no deferral: the three equilibration vectors (D, E of the dynamics rows, E of the box rows) of instance i
    // live at scal + i * 3 * 8 (N + 1) in global memory instead of LDS (they are read at set-up, at the termination checks and by
    // the factorisations, never inside an ADMM iteration), which brings the instance under a third of a CU's LDS; null: in LDS
    double *scal;
    // resume launches that run to completion (defer_after == 0): 1 = take the whole-CU tail kernel where one exists for the
    // handle's (kind, N) (lpvmpc_set_option "defer_tail", default on); it continues the same pool entries
    int tail;
};
constexpr int kParkScalars = 16;     // behind the LDS image of a pool entry: c, cinv, rho, iter, to_chk, to_adp, instance index and the
                                     // instance's output pointers (xPred, uPred, status, iters, polish, resid, state) as 64-bit words
constexpr int LPVMPC_PENDING_ = -11; // status of a parked instance until its resume launch has finished it (lpvmpc.h: LPVMPC_PENDING)

// host-side launchers (defined next to their kernels)
int solve_has_fast_path(int kind, int N);
size_t solve_lds_bytes(int kind, int N);
hipError_t launch_solve(const DevCfg &cfg, const DevCfg *dcfg, const SolveArgs &a, hipStream_t stream, int force_generic);
hipError_t launch_lpv(const DevCfg &cfg, const DevCfg *dcfg, int B, const double *x0, const double *u_prev, const double *vel_ref,
                      const double *curv_s, double cf_new, int lap, double *states, double *AB, hipStream_t stream);
hipError_t launch_abc(const DevCfg &cfg, const DevCfg *dcfg, int B, const double *xlast, const double *delta, double *AB, hipStream_t stream);

// closed-loop helpers (closed_loop.hip)
struct PlantCfg { double lf, lr, m, Iz, mu, dt; int n_sub; };
hipError_t launch_local_position(const DevCfg *dcfg, int B, const double *xypsi, double half_width, double slack, double *out, hipStream_t s);
hipError_t launch_global_position(const DevCfg *dcfg, int B, const double *sey, double *out, hipStream_t s);
hipError_t launch_plant(int B, double *plant, const double *u_a_delta, PlantCfg pc, hipStream_t s);
hipError_t launch_cl_measure(const DevCfg *dcfg, int B, const double *plant, const double *cmd, double half_width, double slack,
                             int q9_swap, double *local_state, double *u_old, hipStream_t s);
hipError_t launch_cl_seed(int B, int N, const double *local_state, double *xlast, double *delta, hipStream_t s);
hipError_t launch_cl_command_plant(int B, int N, const double *uPred, double *cmd, double *plant, PlantCfg pc, hipStream_t s);
hipError_t launch_cl_command_plant_measure(const DevCfg *dcfg, int B, int N, const double *uPred, double *cmd, double *plant, PlantCfg pc,
                                           double hw, double slack, int q9_swap, double *local_next, double *u_old, hipStream_t s);


// planner -> controller hand-off and trajectory-tracking measurement (handoff.hip)
#define LPVMPC_HANDOFF_MAX_N 64
int handoff_length(int N, double dt, double interp_dt);
hipError_t launch_plan_pose(const DevCfg *dcfg, int B, const double *xPred, double *SS, double *pose, double *sig, hipStream_t s);
hipError_t launch_resample(int B, int N, int M, const double *WT, const double *FWT, const double *sig, double *refs, hipStream_t s);
hipError_t launch_plan_first(const DevCfg *dcfg, int B, const double *plant, double hw, double slack, int q9_swap, double accel_rate,
                             double *x0, double *xlast, double *delta, hipStream_t s);
hipError_t launch_tt_measure(const DevCfg *dcfg, int B, int M, int tick, const double *plant, const double *cmd, const double *refs, int latch,
                             double *vel, double *curv, double *ref0, int32_t *lap, int32_t *lap_tick, double *SS, double *local_state,
                             double *u_old, int32_t *alive_ticks, hipStream_t s);

}  // namespace lpvmpc
