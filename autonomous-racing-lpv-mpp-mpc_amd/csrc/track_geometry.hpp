// track_geometry.hpp -- device functions shared by closed_loop.hip and handoff.hip: track coordinate transforms
// and the plant model, one call per vehicle.
//   Map.getLocalPosition     Utilities/trackInitialization.py:283-383 (+ computeAngle :393-411)
//   Map.getGlobalPosition    Utilities/trackInitialization.py:205-262
//   wrap / sign              Utilities/trackInitialization.py:413-431
//   Simulator.f              vehicleSimulator.py:164-199
#pragma once
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
    s = wrap_track_s(s, L);      // NaN beyond kMaxWrapLaps laps / non-finite s: no segment below, NaNs returned
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

}  // namespace lpvmpc
