// lpvmpc_handle.hpp -- private host-side definitions shared by the translation units that implement the C ABI
// (lpvmpc_api.hip: solver entry points and the lap-0 fleet; cascade_api.hip: hand-off and planner + controller cascade).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "lpvmpc.h"
#include "lpvmpc_device.hpp"

#define LPVMPC_HIDDEN __attribute__((visibility("hidden")))

struct lpvmpc_cascade;       // cascade_api.hip

using lpvmpc::DevCfg;
using lpvmpc::SolveArgs;

struct lpvmpc_handle {
    lpvmpc_config cfg;
    DevCfg dev;
    DevCfg *d_cfg;           // device copy of dev (kernels read the configuration through this pointer)
    int nx, nb;
    int cap;                 // workspace capacity (instances)
    // device workspace
    double *d_x0, *d_uprev, *d_vel, *d_curv, *d_uold, *d_maxey, *d_AB, *d_states, *d_xPred, *d_uPred, *d_resid;
    double *d_xlast, *d_delta;
    double *d_scal;          // planner N = 30: the kernel's equilibration vectors [cap][3][8(N+1)] (SolveArgs::scal), else null
    double *d_state;         // warm-start state [cap][3][8(N+1)] (opt-in)
    int warm_mode, state_valid_B;   // 0 off (default); instances whose state is valid from the previous solve
    // closed-loop fleet (lpvmpc_cl_*): plant [B][8], local state [B][6], command [B][2] and scratch
    double *cl_plant, *cl_local, *cl_cmd;
    double *cl_local_next;   // measurement of the coming tick, made by the launch that advanced the plant (valid: cl_next_valid)
    int cl_next_valid;
    int cl_B, cl_first_it, cl_q9, cl_ticks;
    double cl_hw, cl_slack;
    lpvmpc::PlantCfg cl_pc;
    int32_t *d_status, *d_iters, *d_polish;
    hipStream_t stream;
    std::vector<hipEvent_t> ev0, ev1;   // ring of event pairs around the solve-kernel launches
    int ev_count;                       // pairs recorded since timing was (re)enabled
    bool timing;
    int force_generic;                  // 1: always use the run-time-horizon kernel (validation)
    double last_ms;
    std::string err;
    // small-batch I/O staging (lpvmpc_api.hip, IoPack): one pinned host buffer and one device buffer per direction
    char *h_pack_in, *h_pack_out, *d_pack_in, *d_pack_out;
    // planner -> controller hand-off operators (lpvmpc_handoff_setup, planner handles): [M][N] row-major each
    double *d_Wop, *d_FWop;
    int ho_M;
    // straggler deferral (options "defer_after" / "defer_budget" / "defer_pool", lpvmpc_solve_batch_dev only): two pools of
    // parked instances used alternately -- launches park into pool[dcur], the resume pass that follows continues the entries
    // of pool[dcur] and parks what is still unsolved after its budget into the other pool, which becomes dcur
    int defer_after, defer_budget, defer_cap;   // iterations before parking (0 = off); iterations per resume pass; pool entries (0 = default)
    int defer_tail;                     // option "defer_tail" (default 1): passes that run to completion take the whole-CU tail kernel
    bool defer_skip_pass;               // transient: the synchronous entry point joins at once, no bounded pass in between
    int defer_cur_cap, defer_stride;
    double *dpool[2];
    int32_t *dcount[2];
    unsigned long long *dstats;         // [2] device counters: instances parked / parking requests refused (lpvmpc_defer_stats)
    int dcur;
    hipStream_t defer_stream;           // stream of the last deferred call (lpvmpc_join orders against it); valid iff defer_stream_set
    bool defer_stream_set;              // (the null stream is a stream like any other: nullptr cannot mean "none yet")
    hipEvent_t defer_event;
    std::vector<hipEvent_t> rv0, rv1;   // event pairs around the resume launches (timing)
    int rv_count;
    int cascade_prefetch;               // option "cascade_prefetch" (default 1)
    lpvmpc_cascade *cascade;            // owned by the controller handle of a cascade (lpvmpc_cascade_init)
    lpvmpc_handle *cascade_owner;       // planner handle: the controller handle whose cascade drives it (its workspace carries the planner recursion)
};

LPVMPC_HIDDEN int lpvmpc_fail(lpvmpc_handle *h, int code, const char *fmt, ...);
#define fail lpvmpc_fail
#define HIP_TRY(h, expr)                                                                            \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(h, LPVMPC_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)


#define H2D(dst, src, n) HIP_TRY(h, hipMemcpyAsync(dst, src, (n), hipMemcpyHostToDevice, st))
#define D2H(dst, src, n) HIP_TRY(h, hipMemcpyAsync(dst, src, (n), hipMemcpyDeviceToHost, st))

// helpers defined in lpvmpc_api.hip
LPVMPC_HIDDEN int lpvmpc_need_track(lpvmpc_handle *h, const char *who);
LPVMPC_HIDDEN int lpvmpc_check_common(lpvmpc_handle *h, int B, const char *who);       // validates, selects the device, sizes the workspace
LPVMPC_HIDDEN int lpvmpc_check_batch(lpvmpc_handle *h, int B, const char *who);        // the same for the stand-alone batch calls: refused while the handle runs a fleet
LPVMPC_HIDDEN int lpvmpc_launch_solve_timed(lpvmpc_handle *h, const lpvmpc::SolveArgs &a, hipStream_t st);
LPVMPC_HIDDEN lpvmpc::PlantCfg lpvmpc_plant_cfg(const lpvmpc_handle *h, int n_sub, double dt_sim, double mu_sim);
LPVMPC_HIDDEN void lpvmpc_cascade_free(lpvmpc_handle *h);                               // cascade_api.hip
namespace lpvmpc {
// handoff.hip (host): interpolation operator W and interpolation + filtfilt operator FW, both [M][N] row-major
bool handoff_operators(int N, double dt, double interp_dt, int padlen, int ord, const double *b, const double *a,
                       std::vector<double> &W, std::vector<double> &FW);
}
