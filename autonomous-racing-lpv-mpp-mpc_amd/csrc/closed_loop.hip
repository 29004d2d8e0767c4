// closed_loop.hip -- device-side plant model, track coordinate transforms and the per-tick glue of a
// batched closed-loop run.  One lane per vehicle: every piece is a short serial
// recipe per vehicle, the parallel axis is the fleet.
//
// Replaces, for a whole fleet at once, the caller-side code of the reference:
//   Simulator.f                         vehicleSimulator.py:164-199  (bicycle model, linear tyres Fy = 60 alpha)
//   Map.getLocalPosition                Utilities/trackInitialization.py:283-383 (+ computeAngle :393-411)
//   Map.getGlobalPosition               Utilities/trackInitialization.py:205-262
//   lap-0 measurement / command glue    controllerMain.py:179-190, 289-298, 381-386
//   predicted_vectors_generation        controllerMain.py:510-553 (seed trajectories of the first 9 ticks)
#include "lpvmpc_device.hpp"

namespace lpvmpc {

constexpr double kPi = 3.14159265358979323846;

__device__ inline double wrap_pi(double a) {          // TRACK:414-422
    if (a < -kPi) return 2 * kPi + a;
    if (a > kPi) return a - 2 * kPi;
    return a;
}
__device__ inline double sgn1(double a) { return a >= 0 ? 1.0 : -1.0; }   // TRACK:425-431 (sign(0) = +1)
// numpy.unwrap([a, b])[1]
__device__ inline double unwrap2(double a, double b) {
    const double dd = b - a;
    double ddmod = fmod(dd + kPi, 2 * kPi);
    if (ddmod < 0) ddmod += 2 * kPi;
    ddmod -= kPi;
    if (ddmod == -kPi && dd > 0) ddmod = kPi;
    double corr = ddmod - dd;
    if (fabs(dd) < kPi) corr = 0.0;
    return b + corr;
}
// computeAngle(point1, origin, point2), TRACK:393-411
__device__ inline double compute_angle(double p1x, double p1y, double ox, double oy, double p2x, double p2y) {
    const double v1x = p1x - ox, v1y = p1y - oy, v2x = p2x - ox, v2y = p2y - oy;
    return atan2(v1x * v2y - v1y * v2x, v1x * v2x + v1y * v2y);
}
__device__ inline double norm2(double x, double y) { return sqrt(x * x + y * y); }

// Map.getLocalPosition: (x, y, psi) -> (s, ey, epsi, inside); 10000 sentinels when off the track (TRACK:376-379)
__device__ inline void local_position(const DevCfg &c, double hw, double slack, double x, double y, double psi,
                                      double &s, double &ey, double &epsi, int &inside) {
    const double *T = c.track;
    const int rows = c.track_rows;
    int done = 0;
    s = ey = epsi = 0.0;
    for (int i = 0; i < rows && !done; ++i) {
        const int ip = i > 0 ? i - 1 : rows - 1;                       // PointAndTangent[i - 1] wraps to the last row
        const double xf = T[i * 6 + 0], yf = T[i * 6 + 1], xs = T[ip * 6 + 0], ys = T[ip * 6 + 1];
        if (T[i * 6 + 5] == 0.0) {                                      // straight segment
            epsi = unwrap2(T[ip * 6 + 2], psi) - T[ip * 6 + 2];
            if (norm2(xs - x, ys - y) == 0) { s = T[i * 6 + 3]; ey = 0; done = 1; }
            else if (norm2(xf - x, yf - y) == 0) { s = T[i * 6 + 3] + T[i * 6 + 4]; ey = 0; done = 1; }
            else if (fabs(compute_angle(x, y, xs, ys, xf, yf)) <= kPi / 2 && fabs(compute_angle(x, y, xf, yf, xs, ys)) <= kPi / 2) {
                const double n1 = norm2(x - xs, y - ys);
                const double ang = compute_angle(xf, yf, xs, ys, x, y);
                s = n1 * cos(ang) + T[i * 6 + 3];
                ey = n1 * sin(ang);
                if (fabs(ey) <= hw + slack) done = 1;
            }
        } else {
            const double r = 1 / T[i * 6 + 5];
            const double d = r >= 0 ? 1.0 : -1.0;
            const double ang = T[ip * 6 + 2];
            const double cx = xs + fabs(r) * cos(ang + d * kPi / 2), cy = ys + fabs(r) * sin(ang + d * kPi / 2);
            if (norm2(xs - x, ys - y) == 0) { ey = 0; epsi = unwrap2(ang, psi) - ang; s = T[i * 6 + 3]; done = 1; }
            else if (norm2(xf - x, yf - y) == 0) {
                s = T[i * 6 + 3] + T[i * 6 + 4]; ey = 0; epsi = unwrap2(T[i * 6 + 2], psi) - T[i * 6 + 2]; done = 1;
            } else {
                const double arc1 = T[i * 6 + 4] * T[i * 6 + 5];
                const double arc2 = compute_angle(xs, ys, cx, cy, x, y);
                const double s1 = arc1 > 0 ? 1.0 : (arc1 < 0 ? -1.0 : 0.0), s2 = arc2 > 0 ? 1.0 : (arc2 < 0 ? -1.0 : 0.0);
                if (s1 == s2 && fabs(arc1) >= fabs(arc2)) {
                    s = fabs(arc2) * fabs(r) + T[i * 6 + 3];
                    ey = -d * (norm2(x - cx, y - cy) - fabs(r));
                    epsi = unwrap2(ang + arc2, psi) - (ang + arc2);
                    if (fabs(ey) <= hw + slack) done = 1;
                }
            }
        }
    }
    inside = done;
    if (!done) { s = 10000; ey = 10000; epsi = 10000; }
}

// Map.getGlobalPosition: (s, ey) -> (x, y, theta).  Where the reference fails (no segment contains s) NaNs are returned.
__device__ inline void global_position(const DevCfg &c, double s, double ey, double &x, double &y, double &th) {
    const double *T = c.track;
    const int rows = c.track_rows;
    const double L = T[(rows - 1) * 6 + 3] + T[(rows - 1) * 6 + 4];
    for (int it = 0; it < 1000000 && s > L; ++it) s -= L;
    int i = -1;
    for (int k = 0; k < rows; ++k) if (s >= T[k * 6 + 3] && s < T[k * 6 + 3] + T[k * 6 + 4]) { i = k; break; }
    if (i < 0) { x = y = th = __builtin_nan(""); return; }
    const int ip = i > 0 ? i - 1 : rows - 1;
    if (T[i * 6 + 5] == 0.0) {
        const double xf = T[i * 6 + 0], yf = T[i * 6 + 1], xs = T[ip * 6 + 0], ys = T[ip * 6 + 1], psi = T[i * 6 + 2];
        const double dL = T[i * 6 + 4], rL = s - T[i * 6 + 3];
        x = (1 - rL / dL) * xs + rL / dL * xf + ey * cos(psi + kPi / 2);
        y = (1 - rL / dL) * ys + rL / dL * yf + ey * sin(psi + kPi / 2);
        th = psi;
    } else {
        const double r = 1 / T[i * 6 + 5], ang = T[ip * 6 + 2];
        const double d = r >= 0 ? 1.0 : -1.0;
        const double cx = T[ip * 6 + 0] + fabs(r) * cos(ang + d * kPi / 2), cy = T[ip * 6 + 1] + fabs(r) * sin(ang + d * kPi / 2);
        const double span = (s - T[i * 6 + 3]) / (kPi * fabs(r)) * kPi;
        const double an = wrap_pi(d * kPi / 2 + ang);
        const double a0 = -(kPi - fabs(an)) * sgn1(an);
        x = cx + (fabs(r) - d * ey) * cos(a0 + d * span);
        y = cy + (fabs(r) - d * ey) * sin(a0 + d * span);
        th = ang + d * span;
    }
}

// Simulator.f: st = [x y vx vy ax ay yaw psiDot], u = [a, delta]
__device__ inline void plant_step(const PlantCfg &p, double st[8], double ua, double ud) {
    const double x = st[0], y = st[1], vx = st[2], vy = st[3], ax = st[4], ay = st[5], yaw = st[6], w = st[7];
    double aF = 0.0, aR = 0.0;
    if (fabs(vx) > 0.2) {
        aF = ud - atan((vy + p.lf * w) / fabs(vx));
        aR = atan((-vy + p.lr * w) / fabs(vx));
    }
    const double FyF = 60 * aF, FyR = 60 * aR;
    st[0] = x + p.dt * (cos(yaw) * vx - sin(yaw) * vy);
    st[1] = y + p.dt * (sin(yaw) * vx + cos(yaw) * vy);
    st[2] = fabs(vx + p.dt * (ax + w * vy));
    st[3] = vy + p.dt * (ay - w * vx);
    st[4] = ua - p.mu * vx - FyF / p.m * sin(ud);
    st[5] = 1.0 / p.m * (FyF * cos(ud) + FyR);
    st[6] = yaw + p.dt * w;
    st[7] = w + p.dt * (1.0 / p.Iz * (p.lf * FyF * cos(ud) - p.lr * FyR));
}

__global__ void __launch_bounds__(64) local_position_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ in,
                                                            double hw, double slack, double *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double s, ey, epsi; int inside;
    local_position(*cp, hw, slack, in[b * 3 + 0], in[b * 3 + 1], in[b * 3 + 2], s, ey, epsi, inside);
    out[b * 4 + 0] = s; out[b * 4 + 1] = ey; out[b * 4 + 2] = epsi; out[b * 4 + 3] = inside;
}
__global__ void __launch_bounds__(64) global_position_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ in,
                                                             double *__restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x, y, th;
    global_position(*cp, in[b * 2 + 0], in[b * 2 + 1], x, y, th);
    out[b * 3 + 0] = x; out[b * 3 + 1] = y; out[b * 3 + 2] = th;
}
__global__ void __launch_bounds__(64) plant_kernel(int B, double *__restrict__ plant, const double *__restrict__ u, PlantCfg pc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = plant[(size_t)b * 8 + i];
    const double ua = u[b * 2 + 0], ud = u[b * 2 + 1];
    for (int k = 0; k < pc.n_sub; ++k) plant_step(pc, st, ua, ud);
#pragma unroll
    for (int i = 0; i < 8; ++i) plant[(size_t)b * 8 + i] = st[i];
}

// lap-0 measurement: GlobalState = [vx vy psiDot x y psi] from the plant (ground truth), vx clamped at 0.01
// (CMAIN:183-184), local coordinates from the map.  q9_swap reproduces CMAIN:188, which stores the returned
// (s, ey, epsi) as LocalState[4], LocalState[3], LocalState[5], i.e. ey lands in the epsi slot and vice versa
// (SURVEY quirk Q9); with q9_swap = 0 the slots are filled as the state definition says.
// u_old = last command [servo, motor] (CMAIN:289-298 leaves exactly that in OldSteering[0] / OldAccelera[0]).
__global__ void __launch_bounds__(64) cl_measure_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ plant,
                                                        const double *__restrict__ cmd, double hw, double slack, int q9_swap,
                                                        double *__restrict__ local_state, double *__restrict__ u_old) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double *p = plant + (size_t)b * 8;
    double s, ey, epsi; int inside;
    local_position(*cp, hw, slack, p[0], p[1], p[6], s, ey, epsi, inside);
    double *ls = local_state + (size_t)b * 6;
    ls[0] = p[2] < 0.01 ? 0.01 : p[2]; ls[1] = p[3]; ls[2] = p[7];
    ls[3] = q9_swap ? ey : epsi; ls[4] = s; ls[5] = q9_swap ? epsi : ey;
    u_old[b * 2 + 0] = cmd[b * 2 + 0]; u_old[b * 2 + 1] = cmd[b * 2 + 1];
}

// predicted_vectors_generation (CMAIN:510-553): 20 fixed rows built on the local state; delta seeds are zero
__constant__ double kSeedDvx[20] = {0.05, 0.2, 0.4, 0.6, 0.7, 0.8, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9, 0.9};
__constant__ double kSeedDs[20] = {0, 0.01, 0.02, 0.04, 0.07, 0.1, 0.14, 0.18, 0.23, 0.55, 0.66, 0.77, 0.89, 1.00, 1.19, 1.39, 1.59, 1.79, 1.89, 1.999};
__global__ void __launch_bounds__(64) cl_seed_kernel(int B, int N, const double *__restrict__ local_state, double *__restrict__ xlast,
                                                     double *__restrict__ delta) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= B * N) return;
    const int b = t / N, i = t - b * N;
    const double *ls = local_state + (size_t)b * 6;
    double *x = xlast + (size_t)t * 6;
    const int ii = i < 20 ? i : 19;
    x[0] = ls[0] + kSeedDvx[ii]; x[1] = ls[1]; x[2] = ls[2]; x[3] = 0.0001; x[4] = ls[4] + kSeedDs[ii]; x[5] = 0.0001;
    delta[t] = 0.0;
}

// command = first predicted input (CMAIN:381-386: servo = uPred[0,0], motor = uPred[0,1]); then the plant advances
// n_sub simulator steps under that command (u = [motor, servo], vehicleSimulator.py:330)
__global__ void __launch_bounds__(64) cl_command_plant_kernel(int B, int N, const double *__restrict__ uPred, double *__restrict__ cmd,
                                                              double *__restrict__ plant, PlantCfg pc) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const double servo = uPred[(size_t)b * N * 2 + 0], motor = uPred[(size_t)b * N * 2 + 1];
    cmd[b * 2 + 0] = servo; cmd[b * 2 + 1] = motor;
    double st[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) st[i] = plant[(size_t)b * 8 + i];
    for (int k = 0; k < pc.n_sub; ++k) plant_step(pc, st, motor, servo);
#pragma unroll
    for (int i = 0; i < 8; ++i) plant[(size_t)b * 8 + i] = st[i];
}

#define LPVMPC_GRID(n) dim3(((n) + 63) / 64), dim3(64)
hipError_t launch_local_position(const DevCfg *dcfg, int B, const double *xypsi, double hw, double slack, double *out, hipStream_t s) {
    hipLaunchKernelGGL(local_position_kernel, LPVMPC_GRID(B), 0, s, dcfg, B, xypsi, hw, slack, out);
    return hipGetLastError();
}
hipError_t launch_global_position(const DevCfg *dcfg, int B, const double *sey, double *out, hipStream_t s) {
    hipLaunchKernelGGL(global_position_kernel, LPVMPC_GRID(B), 0, s, dcfg, B, sey, out);
    return hipGetLastError();
}
hipError_t launch_plant(int B, double *plant, const double *u, PlantCfg pc, hipStream_t s) {
    hipLaunchKernelGGL(plant_kernel, LPVMPC_GRID(B), 0, s, B, plant, u, pc);
    return hipGetLastError();
}
hipError_t launch_cl_measure(const DevCfg *dcfg, int B, const double *plant, const double *cmd, double hw, double slack, int q9_swap,
                             double *local_state, double *u_old, hipStream_t s) {
    hipLaunchKernelGGL(cl_measure_kernel, LPVMPC_GRID(B), 0, s, dcfg, B, plant, cmd, hw, slack, q9_swap, local_state, u_old);
    return hipGetLastError();
}
hipError_t launch_cl_seed(int B, int N, const double *local_state, double *xlast, double *delta, hipStream_t s) {
    hipLaunchKernelGGL(cl_seed_kernel, LPVMPC_GRID(B * N), 0, s, B, N, local_state, xlast, delta);
    return hipGetLastError();
}
hipError_t launch_cl_command_plant(int B, int N, const double *uPred, double *cmd, double *plant, PlantCfg pc, hipStream_t s) {
    hipLaunchKernelGGL(cl_command_plant_kernel, LPVMPC_GRID(B), 0, s, B, N, uPred, cmd, plant, pc);
    return hipGetLastError();
}

}  // namespace lpvmpc
