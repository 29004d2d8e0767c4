// handoff.hip -- planner -> controller reference hand-off and the trajectory-tracking measurement of the
// controller, for a whole fleet on the device.
//
// Replaces the caller-side code of the reference's two nodes:
//   plannerMain.py:201-224     s integration along the planned states, centre-line pose from the map,
//                              XY / yaw reconstruction, vel = vx, curv = wz / vx
//   plannerMain.py:257-280     cubic interp1d resampling 50 ms -> 33 ms of the five signals, elliptic
//                              filtfilt(padlen = 50) on the curvature           (scipy in the reference)
//   plannerMain.py:465-505     predicted_vectors_generation (seed of the planner's first solve)
//   controllerMain.py:198-283  LapNumber >= 1 branch: yaw unwinding, reference windows [0:N] re-read on every
//                              second tick (`index` toggle, max_window = 0), Body_Frame_Errors (:495-506),
//                              dead-reckoned s, racing lap counter
//
// Resampling and zero-phase filtering are linear maps of the N planner samples that depend only on (N, dt,
// interp_dt, filter): they are built once on the host as dense operators W (interpolation) and F*W (interpolation
// followed by filtfilt) and applied on the device as small matrix products, one workgroup per vehicle.
#include <cmath>
#include <vector>

#include "lpvmpc_device.hpp"
#include "track_geometry.hpp"

namespace lpvmpc {

// ------------------------------------------------------------------------------------------------
// host: operator construction
// ------------------------------------------------------------------------------------------------
// dense Gaussian elimination with partial pivoting, A [n][n] row-major, X [n][k] right-hand sides (in place)
static bool gauss_solve(std::vector<double> &A, std::vector<double> &X, int n, int k) {
    for (int c = 0; c < n; ++c) {
        int p = c;
        for (int r = c + 1; r < n; ++r) if (std::fabs(A[r * n + c]) > std::fabs(A[p * n + c])) p = r;
        if (A[p * n + c] == 0.0) return false;
        if (p != c) {
            for (int j = 0; j < n; ++j) std::swap(A[p * n + j], A[c * n + j]);
            for (int j = 0; j < k; ++j) std::swap(X[p * k + j], X[c * k + j]);
        }
        for (int r = c + 1; r < n; ++r) {
            const double f = A[r * n + c] / A[c * n + c];
            if (f == 0.0) continue;
            for (int j = c; j < n; ++j) A[r * n + j] -= f * A[c * n + j];
            for (int j = 0; j < k; ++j) X[r * k + j] -= f * X[c * k + j];
        }
    }
    for (int c = n - 1; c >= 0; --c) {
        for (int j = 0; j < k; ++j) {
            double v = X[c * k + j];
            for (int r = c + 1; r < n; ++r) v -= A[c * n + r] * X[r * k + j];
            X[c * k + j] = v / A[c * n + c];
        }
    }
    return true;
}

// W [M][N]: values at tau_m = m T / (M - 1) of the not-a-knot cubic spline through (t_i = i T / (N - 1), e_n)
// -- what scipy's interp1d(kind='cubic') evaluates (PMAIN:262-276).
static bool spline_operator(int N, int M, double T, std::vector<double> &W) {
    if (N < 4 || M < 2) return false;
    const double h = T / (N - 1);
    std::vector<double> A((size_t)N * N, 0.0), S((size_t)N * N, 0.0);          // S[i][n] = second derivative at knot i for data e_n
    A[0] = 1; A[1] = -2; A[2] = 1;                                              // not-a-knot: third derivative continuous at t_1
    A[(size_t)(N - 1) * N + N - 3] = 1; A[(size_t)(N - 1) * N + N - 2] = -2; A[(size_t)(N - 1) * N + N - 1] = 1;
    for (int i = 1; i < N - 1; ++i) {
        A[(size_t)i * N + i - 1] = 1; A[(size_t)i * N + i] = 4; A[(size_t)i * N + i + 1] = 1;
        const double g = 6.0 / (h * h);
        S[(size_t)i * N + i - 1] += g; S[(size_t)i * N + i] -= 2 * g; S[(size_t)i * N + i + 1] += g;
    }
    if (!gauss_solve(A, S, N, N)) return false;
    W.assign((size_t)M * N, 0.0);
    for (int m = 0; m < M; ++m) {
        const double tau = m == M - 1 ? T : m * (T / (M - 1));
        int i = (int)std::floor(tau / h);
        if (i > N - 2) i = N - 2;
        if (i < 0) i = 0;
        const double u = tau - i * h;
        for (int n = 0; n < N; ++n) {
            const double yi = n == i ? 1.0 : 0.0, yj = n == i + 1 ? 1.0 : 0.0;
            const double si = S[(size_t)i * N + n], sj = S[(size_t)(i + 1) * N + n];
            W[(size_t)m * N + n] = yi + u * ((yj - yi) / h - h * (2 * si + sj) / 6.0) + u * u * si / 2.0 + u * u * u * (sj - si) / (6.0 * h);
        }
    }
    return true;
}

// scipy.signal.lfilter (transposed direct form II), a[0] = 1, state z [ord] in/out, x filtered in place
static void lfilter(const double *b, const double *a, int ord, std::vector<double> &x, std::vector<double> &z) {
    for (double &xk : x) {
        const double in = xk, y = b[0] * in + z[0];
        for (int i = 0; i < ord - 1; ++i) z[i] = b[i + 1] * in + z[i + 1] - a[i + 1] * y;
        z[ord - 1] = b[ord] * in - a[ord] * y;
        xk = y;
    }
}

// F [M][M]: scipy.signal.filtfilt(b, a, x, padlen = P) (padtype 'odd', method 'pad') as a matrix (PMAIN:280)
static bool filtfilt_operator(const double *b_in, const double *a_in, int ord, int P, int M, std::vector<double> &F) {
    if (ord < 1 || !(a_in[0] != 0.0) || P < 0 || M <= P) return false;
    std::vector<double> b(ord + 1), a(ord + 1);
    for (int i = 0; i <= ord; ++i) { b[i] = b_in[i] / a_in[0]; a[i] = a_in[i] / a_in[0]; }
    // lfilter_zi: (I - companion(a)^T) zi = b[1:] - a[1:] b[0]
    std::vector<double> IA((size_t)ord * ord, 0.0), zi(ord);
    for (int i = 0; i < ord; ++i) {
        IA[(size_t)i * ord + i] += 1.0;
        IA[(size_t)i * ord + 0] += a[i + 1];
        if (i + 1 < ord) IA[(size_t)i * ord + i + 1] -= 1.0;
        zi[i] = b[i + 1] - a[i + 1] * b[0];
    }
    if (!gauss_solve(IA, zi, ord, 1)) return false;
    F.assign((size_t)M * M, 0.0);
    const int E = M + 2 * P;
    std::vector<double> ext(E), z(ord);
    for (int n = 0; n < M; ++n) {
        auto x = [&](int i) { return i == n ? 1.0 : 0.0; };
        for (int i = 0; i < P; ++i) ext[i] = 2 * x(0) - x(P - i);                 // odd extension, left: 2 x0 - x[P .. 1]
        for (int i = 0; i < M; ++i) ext[P + i] = x(i);
        for (int i = 0; i < P; ++i) ext[P + M + i] = 2 * x(M - 1) - x(M - 2 - i); // right: 2 x_end - x[-2 .. -(P+1)]
        for (int i = 0; i < ord; ++i) z[i] = zi[i] * ext[0];
        lfilter(b.data(), a.data(), ord, ext, z);
        const double y0 = ext[E - 1];
        for (int i = 0; i < E / 2; ++i) std::swap(ext[i], ext[E - 1 - i]);
        for (int i = 0; i < ord; ++i) z[i] = zi[i] * y0;
        lfilter(b.data(), a.data(), ord, ext, z);
        for (int i = 0; i < M; ++i) F[(size_t)i * M + n] = ext[E - 1 - (P + i)];  // reverse back, drop the padding
    }
    return true;
}

int handoff_length(int N, double dt, double interp_dt) { return (int)std::nearbyint(N * dt / interp_dt); }   // np.around (PMAIN:259)

// W, FW: [M][N] row-major.  false when the spline / filter cannot be built (N < 4, M <= padlen, singular filter).
bool handoff_operators(int N, double dt, double interp_dt, int padlen, int ord, const double *b, const double *a,
                       std::vector<double> &W, std::vector<double> &FW) {
    const int M = handoff_length(N, dt, interp_dt);
    std::vector<double> F;
    if (!spline_operator(N, M, N * dt, W)) return false;
    if (!filtfilt_operator(b, a, ord, padlen, M, F)) return false;
    FW.assign((size_t)M * N, 0.0);
    for (int i = 0; i < M; ++i)
        for (int k = 0; k < M; ++k) {
            const double f = F[(size_t)i * M + k];
            for (int n = 0; n < N; ++n) FW[(size_t)i * N + n] += f * W[(size_t)k * N + n];
        }
    return true;
}

// ------------------------------------------------------------------------------------------------
// device
// ------------------------------------------------------------------------------------------------
// PMAIN:201-224, one lane per vehicle.  xPred [B][N+1][5]; SS [B][N+1] and pose [B][3] are carried between ticks
// (in/out); sig [B][5][N] = xp, yp, yaw, vel, curv.
__global__ void __launch_bounds__(64) plan_pose_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ xPred,
                                                       double *__restrict__ SS, double *__restrict__ pose, double *__restrict__ sig) {
    const DevCfg &c = *cp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = c.N;
    const double *xp = xPred + (size_t)b * (N + 1) * 5;
    double *ss = SS + (size_t)b * (N + 1), *sg = sig + (size_t)b * 5 * N;
    double X = pose[b * 3 + 0], Y = pose[b * 3 + 1], Th = pose[b * 3 + 2];
    double s = ss[0], s1 = 0.0;
    for (int j = 0; j < N; ++j) {
        const double vx = xp[j * 5 + 0], vy = xp[j * 5 + 1], wz = xp[j * 5 + 2], ey = xp[j * 5 + 3], epsi = xp[j * 5 + 4];
        const double yaw = Th + epsi;                                   // Xref[j], Yref[j], Thetaref[j] belong to stage j
        sg[0 * N + j] = X - ey * sin(yaw);
        sg[1 * N + j] = Y + ey * cos(yaw);
        sg[2 * N + j] = yaw;
        sg[3 * N + j] = vx;
        sg[4 * N + j] = wz / vx;
        const double cv = track_curvature(c, s);
        s = s + ((vx * cos(epsi) - vy * sin(epsi)) / (1 - ey * cv)) * c.dt;
        ss[j + 1] = s;
        global_position(c, s, 0.0, X, Y, Th);
        if (j == 0) { s1 = s; pose[b * 3 + 0] = X; pose[b * 3 + 1] = Y; pose[b * 3 + 2] = Th; }   // Xlast = Xref[1] ...
    }
    ss[0] = s1;                                                         // SS[0] = SS[1]  (PMAIN:216)
}

// refs [B][5][M] = operator x sig [B][5][N]; WT / FWT are the operators transposed ([N][M]) so that lanes (= output
// samples) read consecutive words.  One workgroup per vehicle, the five signals staged in LDS.
__global__ void __launch_bounds__(64) resample_kernel(int B, int N, int M, const double *__restrict__ WT, const double *__restrict__ FWT,
                                                      const double *__restrict__ sig, double *__restrict__ refs) {
    __shared__ double sh[5 * LPVMPC_HANDOFF_MAX_N];
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < 5 * N; i += 64) sh[i] = sig[(size_t)b * 5 * N + i];
    __syncthreads();
    for (int m = threadIdx.x; m < M; m += 64) {
        double acc[5] = {0, 0, 0, 0, 0};
        for (int n = 0; n < N; ++n) {
            const double w = WT[(size_t)n * M + m], fw = FWT[(size_t)n * M + m];
#pragma unroll
            for (int ch = 0; ch < 4; ++ch) acc[ch] = fma(w, sh[ch * N + n], acc[ch]);
            acc[4] = fma(fw, sh[4 * N + n], acc[4]);
        }
#pragma unroll
        for (int ch = 0; ch < 5; ++ch) refs[((size_t)b * 5 + ch) * M + m] = acc[ch];
    }
}

// planner's first tick (PMAIN:137-141 with Testing == 0, :152-162): x0 = [vx vy wz ey epsi] measured from the plant
// (with q9_swap the map's (ey, epsi) land in the (epsi, ey) slots, as PMAIN:141 assigns them: SURVEY quirk Q9),
// seed trajectory of predicted_vectors_generation (PMAIN:465-505): vx ramps with acc_i = 0.1 + 0.2 i, everything
// else frozen, s integrated with zero curvature; steering seeds are zero.
__global__ void __launch_bounds__(64) plan_first_kernel(const DevCfg *__restrict__ cp, int B, const double *__restrict__ plant, double hw,
                                                        double slack, int q9_swap, double accel_rate, double *__restrict__ x0, double *__restrict__ xlast,
                                                        double *__restrict__ delta) {
    const DevCfg &c = *cp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = c.N;
    const double *p = plant + (size_t)b * 8;
    double s, ey, epsi; int inside;
    local_position(c, hw, slack, p[0], p[1], p[6], s, ey, epsi, inside);
    double *x = x0 + (size_t)b * 5;
    x[0] = p[2]; x[1] = p[3]; x[2] = p[7]; x[3] = q9_swap ? epsi : ey; x[4] = q9_swap ? ey : epsi;
    double vx = x[0], S = 0.0;
    for (int i = 0; i < N; ++i) {
        double *r = xlast + ((size_t)b * N + i) * 6;
        r[0] = vx; r[1] = x[1]; r[2] = x[2]; r[3] = x[3]; r[4] = x[4]; r[5] = S;
        delta[(size_t)b * N + i] = 0.0;
        S = S + ((vx * cos(x[4]) - x[1] * sin(x[4])) / (1 - x[3] * 0)) * c.dt;
        vx = vx + (0.1 + accel_rate * i) * c.dt;
    }
}

// controller, LapNumber >= 1 (CMAIN:176-182,198-248,268-283,289-298), one lane per vehicle.  refs [B][5][M] is the
// latest My_Planning message; when `latch` != 0 the windows [0:N] are re-read from it (vel_ref -> vel [B][N+1] with
// the last entry repeated, CTRL:438; curv_ref -> curv [B][N]; x/y/yaw reference of the first sample -> ref0 [B][3]).
__global__ void __launch_bounds__(64) tt_measure_kernel(const DevCfg *__restrict__ cp, int B, int M, int tick, const double *__restrict__ plant,
                                                        const double *__restrict__ cmd, const double *__restrict__ refs, int latch,
                                                        double *__restrict__ vel, double *__restrict__ curv, double *__restrict__ ref0,
                                                        int32_t *__restrict__ lap, int32_t *__restrict__ lap_tick, double *__restrict__ SS,
                                                        double *__restrict__ local_state, double *__restrict__ u_old, int32_t *__restrict__ alive_ticks) {
    const DevCfg &c = *cp;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int N = c.N;
    const double *p = plant + (size_t)b * 8;
    if (alive_ticks) {      // controller ticks this vehicle has entered with a finite plant state (a lost vehicle carries NaN and costs nothing)
        bool fin = true;
        for (int i = 0; i < 8; ++i) fin = fin && __builtin_isfinite(p[i]);
        if (fin) alive_ticks[b] += 1;
    }
    if (latch) {
        const double *r = refs + (size_t)b * 5 * M;
        for (int i = 0; i < N; ++i) { vel[(size_t)b * (N + 1) + i] = r[3 * M + i]; curv[(size_t)b * N + i] = r[4 * M + i]; }
        vel[(size_t)b * (N + 1) + N] = r[3 * M + N - 1];
        ref0[b * 3 + 0] = r[0]; ref0[b * 3 + 1] = r[M]; ref0[b * 3 + 2] = r[2 * M];
    }
    const double vx = p[2] < 0.01 ? 0.01 : p[2], vy = p[3];
    const int lp = lap[b];
    const double psi = wrap_pi(p[6] - 2 * kPi * lp);
    const double xd = ref0[b * 3 + 0], yd = ref0[b * 3 + 1], psid = ref0[b * 3 + 2], cv0 = curv[(size_t)b * N];
    const double ey = -(p[0] - xd) * sin(psid) + (p[1] - yd) * cos(psid);
    const double epsi = wrap_pi(psi - psid);
    const double s = SS[b] + ((vx * cos(epsi) - vy * sin(epsi)) / (1 - ey * cv0)) * c.dt;
    double *ls = local_state + (size_t)b * 6;
    ls[0] = vx; ls[1] = vy; ls[2] = p[7]; ls[3] = epsi; ls[4] = s; ls[5] = ey;
    const double L = c.track[(c.track_rows - 1) * 6 + 3] + c.track[(c.track_rows - 1) * 6 + 4];
    if (lp >= 1 && fabs(p[0]) < 0.1 && s >= L - L / 10) { lap[b] = lp + 1; lap_tick[b] = tick; SS[b] = 0.0; }
    else SS[b] = s;
    u_old[b * 2 + 0] = cmd[b * 2 + 0]; u_old[b * 2 + 1] = cmd[b * 2 + 1];
}

#define LPVMPC_GRID(n) dim3(((n) + 63) / 64), dim3(64)
hipError_t launch_plan_pose(const DevCfg *dcfg, int B, const double *xPred, double *SS, double *pose, double *sig, hipStream_t s) {
    hipLaunchKernelGGL(plan_pose_kernel, LPVMPC_GRID(B), 0, s, dcfg, B, xPred, SS, pose, sig);
    return hipGetLastError();
}
hipError_t launch_resample(int B, int N, int M, const double *WT, const double *FWT, const double *sig, double *refs, hipStream_t s) {
    hipLaunchKernelGGL(resample_kernel, dim3(B), dim3(64), 0, s, B, N, M, WT, FWT, sig, refs);
    return hipGetLastError();
}
hipError_t launch_plan_first(const DevCfg *dcfg, int B, const double *plant, double hw, double slack, int q9_swap, double accel_rate,
                             double *x0, double *xlast, double *delta, hipStream_t s) {
    hipLaunchKernelGGL(plan_first_kernel, LPVMPC_GRID(B), 0, s, dcfg, B, plant, hw, slack, q9_swap, accel_rate, x0, xlast, delta);
    return hipGetLastError();
}
hipError_t launch_tt_measure(const DevCfg *dcfg, int B, int M, int tick, const double *plant, const double *cmd, const double *refs, int latch,
                             double *vel, double *curv, double *ref0, int32_t *lap, int32_t *lap_tick, double *SS, double *local_state,
                             double *u_old, int32_t *alive_ticks, hipStream_t s) {
    hipLaunchKernelGGL(tt_measure_kernel, LPVMPC_GRID(B), 0, s, dcfg, B, M, tick, plant, cmd, refs, latch, vel, curv, ref0, lap, lap_tick, SS,
                       local_state, u_old, alive_ticks);
    return hipGetLastError();
}

}  // namespace lpvmpc
