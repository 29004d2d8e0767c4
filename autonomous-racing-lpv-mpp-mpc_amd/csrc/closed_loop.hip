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
#include "track_geometry.hpp"

namespace lpvmpc {

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

// the same followed by the NEXT tick's measurement (cl_measure_kernel on the state just advanced): one launch less per
// control tick; the measurement goes to its own buffer, the previous tick's local state stays readable
__global__ void __launch_bounds__(64) cl_command_plant_measure_kernel(const DevCfg *__restrict__ cp, int B, int N, const double *__restrict__ uPred,
                                                                      double *__restrict__ cmd, double *__restrict__ plant, PlantCfg pc,
                                                                      double hw, double slack, int q9_swap, double *__restrict__ local_next,
                                                                      double *__restrict__ u_old) {
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
    double s, ey, epsi; int inside;
    local_position(*cp, hw, slack, st[0], st[1], st[6], s, ey, epsi, inside);
    double *ls = local_next + (size_t)b * 6;
    ls[0] = st[2] < 0.01 ? 0.01 : st[2]; ls[1] = st[3]; ls[2] = st[7];
    ls[3] = q9_swap ? ey : epsi; ls[4] = s; ls[5] = q9_swap ? epsi : ey;
    u_old[b * 2 + 0] = servo; u_old[b * 2 + 1] = motor;
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

hipError_t launch_cl_command_plant_measure(const DevCfg *dcfg, int B, int N, const double *uPred, double *cmd, double *plant, PlantCfg pc,
                                           double hw, double slack, int q9_swap, double *local_next, double *u_old, hipStream_t s) {
    hipLaunchKernelGGL(cl_command_plant_measure_kernel, LPVMPC_GRID(B), 0, s, dcfg, B, N, uPred, cmd, plant, pc, hw, slack, q9_swap, local_next, u_old);
    return hipGetLastError();
}

}  // namespace lpvmpc
