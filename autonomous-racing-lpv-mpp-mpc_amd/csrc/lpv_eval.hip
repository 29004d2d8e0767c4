// lpv_eval.hip -- batched LPV evaluation A(rho), B(rho) + horizon roll-out (gfx950).
//
// Replaces the per-stage Python loops of the reference:
//   controller  LPVPrediction  CTRL:166-258   _EstimateABC  CTRL:732-809
//   planner     LPVPrediction  PLAN:242-320   _EstimateABC  PLAN:519-591
// (CTRL = ControllerObject/PathFollowingLPVMPC.py, PLAN = PlannerObject/LPV_MPC_Planner.py)
//
// One lane per instance: the roll-out is a serial recursion over the horizon (the scheduling
// variables of stage i+1 are the rolled-out state of stage i), so the parallel axis is the batch.
// Output layout (HBM): AB[B][N][NX][NX+2] -- row r of stage k holds [A_k[r][0..NX) | B_k[r][0..2)] --
// which is what the solve kernel stages into its per-stage LDS tiles; states[B][N][NX].
#include "lpvmpc_device.hpp"

namespace lpvmpc {

// continuous-time entries shared by both models (CTRL:203-218 == PLAN:275-286)
struct Tyre {
    double a12, a13, a22, a23, a32, a33, b11, b21, b31;
};

__device__ inline Tyre tyre_terms(const DevCfg &c, double Cf, double Cr, double vx, double vy, double delta) {
    Tyre t;
    double sd, cd;
    sincos(delta, &sd, &cd);           // one argument reduction for both (the roll-out is a serial chain of these calls)
    const double m = c.m, I = c.Iz, lf = c.lf, lr = c.lr;
    t.a12 = (sd * Cf) / (m * vx);
    t.a13 = (sd * Cf * lf) / (m * vx) + vy;
    t.a22 = -(Cr + Cf * cd) / (m * vx);
    t.a23 = -(lf * Cf * cd - lr * Cr) / (m * vx) - vx;
    t.a32 = -(lf * Cf * cd - lr * Cr) / (I * vx);
    t.a33 = -(lf * lf * Cf * cd + lr * lr * Cr) / (I * vx);
    t.b11 = -(sd * Cf) / m;
    t.b21 = (cd * Cf) / m;
    t.b31 = (lf * Cf * cd) / I;
    return t;
}

// controller stage: fills ab[6][8] = [I + dt*Ac | dt*Bc]   (CTRL:220-246)
__device__ inline void ctrl_stage(const DevCfg &c, double Cf, double Cr, double vx, double vy, double epsi,
                                  double ey, double cur, double delta, double ab[6][8]) {
    const Tyre t = tyre_terms(c, Cf, Cr, vx, vy, delta);
    const double dt = c.dt;
    double se, ce;
    sincos(epsi, &se, &ce);
    const double den = 1.0 - ey * cur;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) ab[r][a] = 0.0;
    ab[0][0] = 1.0 + dt * (-c.mu); ab[0][1] = dt * t.a12; ab[0][2] = dt * t.a13;
    ab[1][1] = 1.0 + dt * t.a22;   ab[1][2] = dt * t.a23;
    ab[2][1] = dt * t.a32;         ab[2][2] = 1.0 + dt * t.a33;
    ab[3][0] = dt * ((1.0 / den) * (-ce * cur)); ab[3][1] = dt * ((1.0 / den) * (se * cur)); ab[3][2] = dt * 1.0; ab[3][3] = 1.0;
    ab[4][0] = dt * (ce / den);    ab[4][1] = dt * (se / den);  ab[4][4] = 1.0;
    ab[5][0] = dt * se;            ab[5][1] = dt * ce;          ab[5][5] = 1.0;
    ab[0][6] = dt * t.b11; ab[0][7] = dt * 1.0;
    ab[1][6] = dt * t.b21;
    ab[2][6] = dt * t.b31;
}

// planner stage: fills ab[5][7]   (PLAN:288-308), states [vx vy wz ey epsi]
__device__ inline void plan_stage(const DevCfg &c, double vx, double vy, double ey, double epsi, double cur,
                                  double delta, double ab[5][7]) {
    const Tyre t = tyre_terms(c, c.Cf, c.Cr, vx, vy, delta);
    const double dt = c.dt;
    const double A1 = 1.0 / (1.0 - ey * cur);
    const double A2 = sin(epsi);
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int a = 0; a < 7; ++a) ab[r][a] = 0.0;
    ab[0][0] = 1.0 + dt * (-c.mu); ab[0][1] = dt * t.a12; ab[0][2] = dt * t.a13;
    ab[1][1] = 1.0 + dt * t.a22;   ab[1][2] = dt * t.a23;
    ab[2][1] = dt * t.a32;         ab[2][2] = 1.0 + dt * t.a33;
    ab[3][1] = dt * 1.0;           ab[3][3] = 1.0;  ab[3][4] = dt * vx;
    ab[4][0] = dt * (-A1 * cur);   ab[4][1] = dt * (A1 * A2 * cur); ab[4][2] = dt * 1.0; ab[4][4] = 1.0;
    ab[0][5] = dt * t.b11; ab[0][6] = dt * 1.0;
    ab[1][5] = dt * t.b21;
    ab[2][5] = dt * t.b31;
}

// ---------------------------------------------------------------------------------------------
// controller roll-out, CTRL:166-258
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) ctrl_lpv_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ x0,
                                                      const double *__restrict__ u_prev,
                                                      const double *__restrict__ vel_ref,
                                                      const double *__restrict__ curv_ref, double cf_new, int lap,
                                                      double *__restrict__ states, double *__restrict__ AB) {
    const DevCfg &c = *cp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = c.N;
    double st[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) st[r] = x0[(size_t)b * 6 + r];
    for (int i = 0; i < N; ++i) {
        const double vy = st[1], epsi = st[3], s = st[4], ey = st[5];
        const double cur = (lap == 0) ? track_curvature(c, s) : curv_ref[(size_t)b * N + i];
        const double vx = vel_ref[(size_t)b * (N + 1) + i];                 // quirk Q5: vx from vel_ref
        const double u0 = u_prev[((size_t)b * N + i) * 2 + 0], u1 = u_prev[((size_t)b * N + i) * 2 + 1];
        double ab[6][8];
        ctrl_stage(c, cf_new, cf_new, vx, vy, epsi, ey, cur, u0, ab);
        double nx[6];
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < 6; ++a) acc += ab[r][a] * st[a];
            nx[r] = acc + (ab[r][6] * u0 + ab[r][7] * u1);
        }
        if (AB) {
            double *o = AB + ((size_t)b * N + i) * 48;
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int a = 0; a < 8; ++a) o[r * 8 + a] = ab[r][a];
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            st[r] = nx[r];
            if (states) states[((size_t)b * N + i) * 6 + r] = nx[r];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same roll-out in two launches (used whenever AB is wanted, i.e. by every caller of the library).
// Quirk Q5 makes most of a controller stage independent of the rolled-out state: vx comes from vel_ref, the steering angle from
// u_prev.  Only vy (one summand of A[0][2]), epsi and ey (rows 3-5) and -- on lap 0 -- the curvature at the rolled-out s are not
// known in advance.  The serial kernel above executes ~600 double-precision instructions per stage on one lane per instance
// (two sincos, nine divisions, 48 multiply-adds, 54 strided stores): 16 wavefronts busy for 42 us at B = 1024.  Here
//   ctrl_lpv_pre_kernel   one thread per (instance, stage): sincos(delta), the tyre terms with their six divisions, and the
//                         whole AB tile written (the seven state-dependent entries as place-holders; slot [0][2] carries the
//                         state-independent summand of a13);
//   ctrl_lpv_roll_kernel  one lane per instance, per stage: sincos(epsi), 1/den and two more divisions, the seven entries,
//                         the structurally non-zero products of the mat-vec and 13 stores.
// Every value is formed by the same operations in the same order as in ctrl_stage / the loop above (products with the
// structural zeros only ever added +-0), so the outputs are the same values (the sign of an exact zero aside).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) ctrl_lpv_pre_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ u_prev,
                                                          const double *__restrict__ vel_ref, double cf_new, double *__restrict__ AB) {
    const DevCfg &c = *cp;
    const int N = c.N;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * N) return;
    const int b = t / N, i = t - b * N;
    const double vx = vel_ref[(size_t)b * (N + 1) + i], delta = u_prev[(size_t)t * 2];
    const double Cf = cf_new, Cr = cf_new, m = c.m, I = c.Iz, lf = c.lf, lr = c.lr, dt = c.dt;
    double sd, cd;
    sincos(delta, &sd, &cd);
    const double a12 = (sd * Cf) / (m * vx);
    const double p13 = (sd * Cf * lf) / (m * vx);                  // a13 = p13 + vy
    const double a22 = -(Cr + Cf * cd) / (m * vx);
    const double a23 = -(lf * Cf * cd - lr * Cr) / (m * vx) - vx;
    const double a32 = -(lf * Cf * cd - lr * Cr) / (I * vx);
    const double a33 = -(lf * lf * Cf * cd + lr * lr * Cr) / (I * vx);
    const double b11 = -(sd * Cf) / m, b21 = (cd * Cf) / m, b31 = (lf * Cf * cd) / I;
    double ab[6][8];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) ab[r][a] = 0.0;
    ab[0][0] = 1.0 + dt * (-c.mu); ab[0][1] = dt * a12; ab[0][2] = p13;
    ab[1][1] = 1.0 + dt * a22;     ab[1][2] = dt * a23;
    ab[2][1] = dt * a32;           ab[2][2] = 1.0 + dt * a33;
    ab[3][2] = dt * 1.0; ab[3][3] = 1.0;
    ab[4][4] = 1.0;
    ab[5][5] = 1.0;
    ab[0][6] = dt * b11; ab[0][7] = dt * 1.0;
    ab[1][6] = dt * b21;
    ab[2][6] = dt * b31;
    double *o = AB + (size_t)t * 48;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) o[r * 8 + a] = ab[r][a];
}

__global__ void __launch_bounds__(64) ctrl_lpv_roll_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ x0,
                                                           const double *__restrict__ u_prev, const double *__restrict__ curv_ref,
                                                           int lap, double *__restrict__ states, double *__restrict__ AB) {
    const DevCfg &c = *cp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = c.N;
    const double dt = c.dt;
    double st[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) st[r] = x0[(size_t)b * 6 + r];
    for (int i = 0; i < N; ++i) {
        double *o = AB + ((size_t)b * N + i) * 48;
        // the state-independent entries the mat-vec needs (written by ctrl_lpv_pre_kernel)
        const double a00 = o[0], a01 = o[1], p13 = o[2], a11 = o[9], a12_ = o[10], a21 = o[17], a22_ = o[18];
        const double b06 = o[6], b07 = o[7], b16 = o[14], b26 = o[22];
        const double vy = st[1], epsi = st[3], s = st[4], ey = st[5];
        const double cur = (lap == 0) ? track_curvature(c, s) : curv_ref[(size_t)b * N + i];
        const double u0 = u_prev[((size_t)b * N + i) * 2 + 0], u1 = u_prev[((size_t)b * N + i) * 2 + 1];
        double se, ce;
        sincos(epsi, &se, &ce);
        const double den = 1.0 - ey * cur;
        const double a02 = dt * (p13 + vy);
        const double a30 = dt * ((1.0 / den) * (-ce * cur)), a31 = dt * ((1.0 / den) * (se * cur)), a32 = dt * 1.0;
        const double a40 = dt * (ce / den), a41 = dt * (se / den);
        const double a50 = dt * se, a51 = dt * ce;
        // rows of [A | B] times [st ; u]: the products with structural zeros are left out (they only ever add +-0), the rest in
        // the order of the dense loop: ((((((0 + p0) + p1) + p2) + p3) + p4) + p5) + (b6 u0 + b7 u1)
        double nx[6];
        nx[0] = ((a00 * st[0] + a01 * st[1]) + a02 * st[2]) + (b06 * u0 + b07 * u1);
        nx[1] = (a11 * st[1] + a12_ * st[2]) + (b16 * u0 + 0.0 * u1);
        nx[2] = (a21 * st[1] + a22_ * st[2]) + (b26 * u0 + 0.0 * u1);
        nx[3] = (((a30 * st[0] + a31 * st[1]) + a32 * st[2]) + 1.0 * st[3]) + (0.0 * u0 + 0.0 * u1);
        nx[4] = ((a40 * st[0] + a41 * st[1]) + 1.0 * st[4]) + (0.0 * u0 + 0.0 * u1);
        nx[5] = ((a50 * st[0] + a51 * st[1]) + 1.0 * st[5]) + (0.0 * u0 + 0.0 * u1);
        o[2] = a02;
        o[24] = a30; o[25] = a31;
        o[32] = a40; o[33] = a41;
        o[40] = a50; o[41] = a51;
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            st[r] = nx[r];
            if (states) states[((size_t)b * N + i) * 6 + r] = nx[r];
        }
    }
}

// controller seed-mode linearisation, CTRL:732-809 (vx from the trajectory, curvature from the map)
__global__ void __launch_bounds__(64) ctrl_abc_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ xlast,
                                                      const double *__restrict__ delta, double *__restrict__ AB) {
    const DevCfg &c = *cp;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = c.N;
    if (t >= B * N) return;
    const double *x = xlast + (size_t)t * 6;
    const double cur = track_curvature(c, x[4]);
    double ab[6][8];
    ctrl_stage(c, c.Cf, c.Cr, x[0], x[1], x[3], x[5], cur, delta[t], ab);
    double *o = AB + (size_t)t * 48;
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int a = 0; a < 8; ++a) o[r * 8 + a] = ab[r][a];
}

// ---------------------------------------------------------------------------------------------
// planner roll-out, PLAN:242-320
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) plan_lpv_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ x0,
                                                      const double *__restrict__ u_prev,
                                                      const double *__restrict__ SS, double *__restrict__ states,
                                                      double *__restrict__ AB) {
    const DevCfg &c = *cp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = c.N;
    double st[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) st[r] = x0[(size_t)b * 5 + r];
    for (int i = 0; i < N; ++i) {
        const double cur = track_curvature(c, SS[(size_t)b * (N + 1) + i]);
        const double u0 = u_prev[((size_t)b * N + i) * 2 + 0], u1 = u_prev[((size_t)b * N + i) * 2 + 1];
        double ab[5][7];
        plan_stage(c, st[0], st[1], st[3], st[4], cur, u0, ab);
        double nx[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            double acc = 0.0;
#pragma unroll
            for (int a = 0; a < 5; ++a) acc += ab[r][a] * st[a];
            nx[r] = acc + (ab[r][5] * u0 + ab[r][6] * u1);
        }
        if (AB) {
            double *o = AB + ((size_t)b * N + i) * 35;
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int a = 0; a < 7; ++a) o[r * 7 + a] = ab[r][a];
        }
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            st[r] = nx[r];
            if (states) states[((size_t)b * N + i) * 5 + r] = nx[r];
        }
    }
}

// planner seed-mode linearisation, PLAN:519-591; xlast columns [vx vy wz ey epsi s]
__global__ void __launch_bounds__(64) plan_abc_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ xlast,
                                                      const double *__restrict__ delta, double *__restrict__ AB) {
    const DevCfg &c = *cp;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = c.N;
    if (t >= B * N) return;
    const double *x = xlast + (size_t)t * 6;
    const double cur = track_curvature(c, x[5]);
    double ab[5][7];
    plan_stage(c, x[0], x[1], x[3], x[4], cur, delta[t], ab);
    double *o = AB + (size_t)t * 35;
#pragma unroll
    for (int r = 0; r < 5; ++r)
#pragma unroll
        for (int a = 0; a < 7; ++a) o[r * 7 + a] = ab[r][a];
}

hipError_t launch_lpv(const DevCfg &cfg, const DevCfg *dcfg, int B, const double *x0, const double *u_prev, const double *vel_ref,
                      const double *curv_s, double cf_new, int lap, double *states, double *AB, hipStream_t stream) {
    const int blocks = (B + 63) / 64;
    if (cfg.kind == 0 && AB) {
        hipLaunchKernelGGL(ctrl_lpv_pre_kernel, dim3((B * cfg.N + 63) / 64), dim3(64), 0, stream, dcfg, B, u_prev, vel_ref, cf_new, AB);
        hipLaunchKernelGGL(ctrl_lpv_roll_kernel, dim3(blocks), dim3(64), 0, stream, dcfg, B, x0, u_prev, curv_s, lap, states, AB);
    } else if (cfg.kind == 0)
        hipLaunchKernelGGL(ctrl_lpv_kernel, dim3(blocks), dim3(64), 0, stream, dcfg, B, x0, u_prev, vel_ref, curv_s, cf_new, lap, states, AB);
    else
        hipLaunchKernelGGL(plan_lpv_kernel, dim3(blocks), dim3(64), 0, stream, dcfg, B, x0, u_prev, curv_s, states, AB);
    return hipGetLastError();
}

hipError_t launch_abc(const DevCfg &cfg, const DevCfg *dcfg, int B, const double *xlast, const double *delta, double *AB, hipStream_t stream) {
    const int blocks = (B * cfg.N + 63) / 64;
    if (cfg.kind == 0)
        hipLaunchKernelGGL(ctrl_abc_kernel, dim3(blocks), dim3(64), 0, stream, dcfg, B, xlast, delta, AB);
    else
        hipLaunchKernelGGL(plan_abc_kernel, dim3(blocks), dim3(64), 0, stream, dcfg, B, xlast, delta, AB);
    return hipGetLastError();
}

}  // namespace lpvmpc
