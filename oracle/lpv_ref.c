/*
 * oracle/lpv_ref.c -- TEST INFRASTRUCTURE (CPU oracle / cpu_baseline leg of bench.py), not product code.
 *
 * C float64 restatement of one controller tick of the reference, end to end:
 *   LPVPrediction            ControllerObject/PathFollowingLPVMPC.py:166-258
 *   Curvature                Utilities/utilities.py:31-50
 *   _buildMatEqConst         PathFollowingLPVMPC.py:477-529
 *   _buildMatCost            PathFollowingLPVMPC.py:382-473
 *   _buildMatIneqConst       PathFollowingLPVMPC.py:329-378
 *   osqp_solve_qp            PathFollowingLPVMPC.py:273-325   (-> osqp_ref_solve, oracle/osqp_ref.c)
 * assembled directly in sparse (CSC) form in the reference's variable order z = [x_0..x_N, u_0..u_{N-1}]
 * and row order (inequalities first, then equalities), looped over a batch with OpenMP.
 * It is checked against oracle/lpv_ref.py (itself pinned to the golden vectors) in tests/test_oracle_osqp.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf, delta, adaptive_rho_tolerance;
    int max_iter, check_termination, scaling, adaptive_rho, adaptive_rho_interval, polish, polish_refine_iter,
        scaled_termination;
} osqp_ref_settings;
typedef struct {
    int iter, status_val, status_polish, rho_updates;
    double obj_val, pri_res, dua_res, rho_estimate, rho_final;
} osqp_ref_info;
void osqp_ref_default_settings(osqp_ref_settings *s);
int osqp_ref_solve_ws(int n, int m, const int *Pp, const int *Pi, const double *Px, const double *q, const int *Ap,
                      const int *Ai, const double *Ax, const double *l, const double *u, const int *perm,
                      const osqp_ref_settings *settings, const double *x_ws, const double *y_ws, double *x_out,
                      double *y_out, osqp_ref_info *info_out);

static double curvature(const double *tab, int rows, double s) {   /* UTIL:31-50 */
    const double L = tab[(rows - 1) * 6 + 3] + tab[(rows - 1) * 6 + 4];
    while (s > L) s -= L;
    for (int i = 0; i < rows; i++) if (s >= tab[i * 6 + 3] && s < tab[i * 6 + 3] + tab[i * 6 + 4]) return tab[i * 6 + 5];
    return NAN;   /* the reference raises here */
}

/* [A|B] of one controller stage, row-major 6x8 (CTRL:203-246) */
static void ctrl_stage(const double *p, double dt, double Cf, double Cr, double vx, double vy, double epsi, double ey,
                       double cur, double delta, double ab[48]) {
    const double lf = p[0], lr = p[1], m = p[2], I = p[3], mu = p[6];
    const double sd = sin(delta), cd = cos(delta), se = sin(epsi), ce = cos(epsi), den = 1.0 - ey * cur;
    memset(ab, 0, sizeof(double) * 48);
    ab[0 * 8 + 0] = 1.0 + dt * (-mu);
    ab[0 * 8 + 1] = dt * ((sd * Cf) / (m * vx));
    ab[0 * 8 + 2] = dt * ((sd * Cf * lf) / (m * vx) + vy);
    ab[1 * 8 + 1] = 1.0 + dt * (-(Cr + Cf * cd) / (m * vx));
    ab[1 * 8 + 2] = dt * (-(lf * Cf * cd - lr * Cr) / (m * vx) - vx);
    ab[2 * 8 + 1] = dt * (-(lf * Cf * cd - lr * Cr) / (I * vx));
    ab[2 * 8 + 2] = 1.0 + dt * (-(lf * lf * Cf * cd + lr * lr * Cr) / (I * vx));
    ab[3 * 8 + 0] = dt * ((1.0 / den) * (-ce * cur));
    ab[3 * 8 + 1] = dt * ((1.0 / den) * (se * cur));
    ab[3 * 8 + 2] = dt * 1.0; ab[3 * 8 + 3] = 1.0;
    ab[4 * 8 + 0] = dt * (ce / den); ab[4 * 8 + 1] = dt * (se / den); ab[4 * 8 + 4] = 1.0;
    ab[5 * 8 + 0] = dt * se; ab[5 * 8 + 1] = dt * ce; ab[5 * 8 + 5] = 1.0;
    ab[0 * 8 + 6] = dt * (-(sd * Cf) / m); ab[0 * 8 + 7] = dt * 1.0;
    ab[1 * 8 + 6] = dt * ((cd * Cf) / m);
    ab[2 * 8 + 6] = dt * ((lf * Cf * cd) / I);
}

/*
 * One controller tick per instance: LPV roll-out from x0 (CTRL:166-258), QP assembly, OSQP restatement.
 * params = [lf lr m Iz Cf Cr mu max_vel]; vel_ref [B][N+1] (entry N = vel_ref[-1]); curv_ref [B][N];
 * u_prev [B][N][2]; u_old [B][2]; outputs xPred [B][N+1][6], uPred [B][N][2], status/iters [B].
 * Returns 0.  nthreads <= 0: OpenMP default.
 */
int oracle_ctrl_tick_batch(int B, int N, double dt, const double *params, const double *Q, const double *R,
                           const double *dR, const double *track, int track_rows, const double *x0,
                           const double *u_prev, const double *vel_ref, const double *curv_ref, const double *u_old,
                           double cf_new, int lap, double *xPred, double *uPred, int *status, int *iters,
                           int nthreads, const double *z_ws, const double *y_ws, int shift, double *z_sol, double *y_sol) {
    /* z_ws [B][nz] / y_ws [B][m] (optional): previous solution in the reference's variable / row order, used as a
     * warm start, shifted by one stage when shift != 0 (stage k takes stage k+1, the last stage is kept).
     * z_sol / y_sol (optional) receive this tick's primal / dual solution in the same orders. */
    const int nx = 6, nu = 2, nz = (N + 1) * nx + N * nu, mi = 6 * N, me = (N + 1) * nx, m = mi + me;
    const double max_vel = params[7];
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    /* KKT ordering: stage by stage (variables, box rows, dynamics rows into the next stage) */
    int *perm = (int *)malloc(sizeof(int) * (nz + m)); int pc = 0;
    for (int r = 0; r < nx; r++) perm[pc++] = nz + mi + r;                      /* x_0 = x0 rows first */
    for (int k = 0; k <= N; k++) {
        for (int a = 0; a < nx; a++) perm[pc++] = k * nx + a;
        if (k < N) {
            for (int j = 0; j < nu; j++) perm[pc++] = (N + 1) * nx + k * nu + j;
            perm[pc++] = nz + 2 * k; perm[pc++] = nz + 2 * k + 1;
            for (int t = 0; t < 4; t++) perm[pc++] = nz + 2 * N + 4 * k + t;
            for (int r = 0; r < nx; r++) perm[pc++] = nz + mi + (k + 1) * nx + r;
        }
    }
    osqp_ref_settings st; osqp_ref_default_settings(&st);
#pragma omp parallel
    {
        double *AB = (double *)malloc(sizeof(double) * N * 48);
        /* P upper-triangular CSC */
        int *Pp = (int *)malloc(sizeof(int) * (nz + 1)), *Pi = (int *)malloc(sizeof(int) * (nz * 8));
        double *Px = (double *)malloc(sizeof(double) * (nz * 8)), *q = (double *)malloc(sizeof(double) * nz);
        int *Ap = (int *)malloc(sizeof(int) * (nz + 1)), *Ai = (int *)malloc(sizeof(int) * (nz * 10));
        double *Ax = (double *)malloc(sizeof(double) * (nz * 10));
        double *l = (double *)malloc(sizeof(double) * m), *u = (double *)malloc(sizeof(double) * m);
        double *xo = (double *)malloc(sizeof(double) * nz), *yo = (double *)malloc(sizeof(double) * m);
        double *xw = (double *)malloc(sizeof(double) * nz), *yw = (double *)malloc(sizeof(double) * m);
#pragma omp for schedule(dynamic, 4)
        for (int b = 0; b < B; b++) {
            /* ---- LPV roll-out ---- */
            double stt[6]; memcpy(stt, x0 + (size_t)b * 6, sizeof(stt));
            for (int i = 0; i < N; i++) {
                const double cur = lap == 0 ? curvature(track, track_rows, stt[4]) : curv_ref[(size_t)b * N + i];
                const double *uu = u_prev + ((size_t)b * N + i) * 2;
                double *ab = AB + i * 48;
                ctrl_stage(params, dt, cf_new, cf_new, vel_ref[(size_t)b * (N + 1) + i], stt[1], stt[3], stt[5], cur, uu[0], ab);
                double nxt[6];
                for (int r = 0; r < 6; r++) {
                    double acc = 0; for (int a = 0; a < 6; a++) acc += ab[r * 8 + a] * stt[a];
                    nxt[r] = acc + (ab[r * 8 + 6] * uu[0] + ab[r * 8 + 7] * uu[1]);
                }
                memcpy(stt, nxt, sizeof(stt));
            }
            /* ---- cost: P = 2 M0 (upper), q ---- */
            int pn = 0;
            for (int k = 0; k <= N; k++) for (int a = 0; a < nx; a++) {
                const int col = k * nx + a; Pp[col] = pn;
                for (int r = 0; r <= a; r++) { const double v = 2.0 * Q[r * nx + a]; if (v != 0.0) { Pi[pn] = k * nx + r; Px[pn++] = v; } }
                const double vr = vel_ref[(size_t)b * (N + 1) + k];
                q[col] = -2.0 * vr * Q[0 * nx + a];
            }
            for (int k = 0; k < N; k++) for (int j = 0; j < nu; j++) {
                const int col = (N + 1) * nx + k * nu + j; Pp[col] = pn;
                if (k >= 1) { const double v = -2.0 * dR[j]; if (v != 0.0) { Pi[pn] = col - nu; Px[pn++] = v; } }
                for (int r = 0; r <= j; r++) {
                    double v = 2.0 * R[r * nu + j]; if (r == j) v += 2.0 * (k < N - 1 ? 2.0 : 1.0) * dR[j];
                    if (v != 0.0) { Pi[pn] = (N + 1) * nx + k * nu + r; Px[pn++] = v; }
                }
                q[col] = (k == 0) ? -2.0 * u_old[(size_t)b * 2 + j] * dR[j] : 0.0;
            }
            Pp[nz] = pn;
            /* ---- constraints: rows [2N state ineq | 4N input ineq | (N+1)*nx equalities] ---- */
            int an = 0;
            for (int k = 0; k <= N; k++) for (int a = 0; a < nx; a++) {
                const int col = k * nx + a; Ap[col] = an;
                if (a == 0 && k < N) { Ai[an] = 2 * k; Ax[an++] = -1.0; Ai[an] = 2 * k + 1; Ax[an++] = 1.0; }
                Ai[an] = mi + k * nx + a; Ax[an++] = 1.0;
                if (k < N) for (int r = 0; r < nx; r++) { const double v = -AB[k * 48 + r * 8 + a]; if (v != 0.0) { Ai[an] = mi + (k + 1) * nx + r; Ax[an++] = v; } }
            }
            for (int k = 0; k < N; k++) for (int j = 0; j < nu; j++) {
                const int col = (N + 1) * nx + k * nu + j; Ap[col] = an;
                Ai[an] = 2 * N + 4 * k + 2 * j; Ax[an++] = 1.0; Ai[an] = 2 * N + 4 * k + 2 * j + 1; Ax[an++] = -1.0;
                for (int r = 0; r < nx; r++) { const double v = -AB[k * 48 + r * 8 + nx + j]; if (v != 0.0) { Ai[an] = mi + (k + 1) * nx + r; Ax[an++] = v; } }
            }
            Ap[nz] = an;
            for (int k = 0; k < N; k++) {
                l[2 * k] = -INFINITY; u[2 * k] = -0.01; l[2 * k + 1] = -INFINITY; u[2 * k + 1] = max_vel;
                const int r = 2 * N + 4 * k;
                l[r] = l[r + 1] = l[r + 2] = l[r + 3] = -INFINITY;
                u[r] = 0.249; u[r + 1] = 0.249; u[r + 2] = 4.0; u[r + 3] = 1.0;
            }
            for (int r = 0; r < me; r++) l[mi + r] = u[mi + r] = (r < nx) ? x0[(size_t)b * 6 + r] : 0.0;
            osqp_ref_info info;
            const double *xws = 0, *yws = 0;
            if (z_ws && y_ws) {
                const double *zp = z_ws + (size_t)b * nz, *yp = y_ws + (size_t)b * m;
                for (int k = 0; k <= N; k++) { const int ks = shift ? (k < N ? k + 1 : N) : k;
                    for (int a = 0; a < nx; a++) { xw[k * nx + a] = zp[ks * nx + a]; yw[mi + k * nx + a] = yp[mi + ks * nx + a]; } }
                for (int k = 0; k < N; k++) { const int ks = shift ? (k < N - 1 ? k + 1 : N - 1) : k;
                    for (int j = 0; j < nu; j++) xw[(N + 1) * nx + k * nu + j] = zp[(N + 1) * nx + ks * nu + j];
                    yw[2 * k] = yp[2 * ks]; yw[2 * k + 1] = yp[2 * ks + 1];
                    for (int t2 = 0; t2 < 4; t2++) yw[2 * N + 4 * k + t2] = yp[2 * N + 4 * ks + t2]; }
                xws = xw; yws = yw;
            }
            if (osqp_ref_solve_ws(nz, m, Pp, Pi, Px, q, Ap, Ai, Ax, l, u, perm, &st, xws, yws, xo, yo, &info) != 0) {
                for (int i = 0; i < nz; i++) xo[i] = NAN;          /* KKT factorisation broke down: no answer from the oracle */
                for (int i = 0; i < m; i++) yo[i] = NAN;
                info.status_val = -10; info.iter = 0;
            }
            memcpy(xPred + (size_t)b * (N + 1) * nx, xo, sizeof(double) * (N + 1) * nx);
            memcpy(uPred + (size_t)b * N * nu, xo + (N + 1) * nx, sizeof(double) * N * nu);
            if (z_sol) memcpy(z_sol + (size_t)b * nz, xo, sizeof(double) * nz);
            if (y_sol) memcpy(y_sol + (size_t)b * m, yo, sizeof(double) * m);
            if (status) status[b] = info.status_val;
            if (iters) iters[b] = info.iter;
        }
        free(AB); free(Pp); free(Pi); free(Px); free(q); free(Ap); free(Ai); free(Ax); free(l); free(u); free(xo); free(yo);
        free(xw); free(yw);
    }
    free(perm);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * planner tick: LPVPrediction PLAN:242-320, _buildMatEqConst PLAN:434-486, cost/bounds/OSQP PLAN:86-236
 * (equalities first, then the identity box on every variable, PLAN:200-202).
 * params = [lf lr m Iz Cf Cr mu max_vel min_vel]; states [vx vy wz ey epsi].
 * ---------------------------------------------------------------------------------------------- */
static void plan_stage(const double *p, double dt, double vx, double vy, double ey, double epsi, double cur,
                       double delta, double ab[35]) {
    const double lf = p[0], lr = p[1], m = p[2], I = p[3], Cf = p[4], Cr = p[5], mu = p[6];
    const double sd = sin(delta), cd = cos(delta), A1 = 1.0 / (1.0 - ey * cur), A2 = sin(epsi);
    memset(ab, 0, sizeof(double) * 35);
    ab[0 * 7 + 0] = 1.0 + dt * (-mu);
    ab[0 * 7 + 1] = dt * ((sd * Cf) / (m * vx));
    ab[0 * 7 + 2] = dt * ((sd * Cf * lf) / (m * vx) + vy);
    ab[1 * 7 + 1] = 1.0 + dt * (-(Cr + Cf * cd) / (m * vx));
    ab[1 * 7 + 2] = dt * (-(lf * Cf * cd - lr * Cr) / (m * vx) - vx);
    ab[2 * 7 + 1] = dt * (-(lf * Cf * cd - lr * Cr) / (I * vx));
    ab[2 * 7 + 2] = 1.0 + dt * (-(lf * lf * Cf * cd + lr * lr * Cr) / (I * vx));
    ab[3 * 7 + 1] = dt * 1.0; ab[3 * 7 + 3] = 1.0; ab[3 * 7 + 4] = dt * vx;
    ab[4 * 7 + 0] = dt * (-A1 * cur); ab[4 * 7 + 1] = dt * (A1 * A2 * cur); ab[4 * 7 + 2] = dt * 1.0; ab[4 * 7 + 4] = 1.0;
    ab[0 * 7 + 5] = dt * (-(sd * Cf) / m); ab[0 * 7 + 6] = dt * 1.0;
    ab[1 * 7 + 5] = dt * ((cd * Cf) / m);
    ab[2 * 7 + 5] = dt * ((lf * Cf * cd) / I);
}

int oracle_plan_tick_batch(int B, int N, double dt, const double *params, const double *Q, const double *R,
                           const double *dR, const double *Lcf, const double *track, int track_rows,
                           const double *x0, const double *u_prev, const double *SS, const double *u_old,
                           const double *max_ey, double *xPred, double *uPred, int *status, int *iters, int nthreads) {
    const int nx = 5, nu = 2, nz = (N + 1) * nx + N * nu, me = (N + 1) * nx, m = me + nz;
    const double max_vel = params[7], min_vel = params[8];
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    int *perm = (int *)malloc(sizeof(int) * (nz + m)); int pc = 0;
    for (int r = 0; r < nx; r++) perm[pc++] = nz + r;
    for (int k = 0; k <= N; k++) {
        for (int a = 0; a < nx; a++) { perm[pc++] = k * nx + a; perm[pc++] = nz + me + k * nx + a; }
        if (k < N) {
            for (int j = 0; j < nu; j++) { const int v = (N + 1) * nx + k * nu + j; perm[pc++] = v; perm[pc++] = nz + me + v; }
            for (int r = 0; r < nx; r++) perm[pc++] = nz + (k + 1) * nx + r;
        }
    }
    osqp_ref_settings st; osqp_ref_default_settings(&st);
#pragma omp parallel
    {
        double *AB = (double *)malloc(sizeof(double) * N * 35);
        int *Pp = (int *)malloc(sizeof(int) * (nz + 1)), *Pi = (int *)malloc(sizeof(int) * (nz * 8));
        double *Px = (double *)malloc(sizeof(double) * (nz * 8)), *q = (double *)malloc(sizeof(double) * nz);
        int *Ap = (int *)malloc(sizeof(int) * (nz + 1)), *Ai = (int *)malloc(sizeof(int) * (nz * 10));
        double *Ax = (double *)malloc(sizeof(double) * (nz * 10));
        double *l = (double *)malloc(sizeof(double) * m), *u = (double *)malloc(sizeof(double) * m);
        double *xo = (double *)malloc(sizeof(double) * nz), *yo = (double *)malloc(sizeof(double) * m);
#pragma omp for schedule(dynamic, 4)
        for (int b = 0; b < B; b++) {
            double stt[5]; memcpy(stt, x0 + (size_t)b * 5, sizeof(stt));
            for (int i = 0; i < N; i++) {
                const double cur = curvature(track, track_rows, SS[(size_t)b * (N + 1) + i]);
                const double *uu = u_prev + ((size_t)b * N + i) * 2;
                double *ab = AB + i * 35;
                plan_stage(params, dt, stt[0], stt[1], stt[3], stt[4], cur, uu[0], ab);
                double nxt[5];
                for (int r = 0; r < 5; r++) {
                    double acc = 0; for (int a = 0; a < 5; a++) acc += ab[r * 7 + a] * stt[a];
                    nxt[r] = acc + (ab[r * 7 + 5] * uu[0] + ab[r * 7 + 6] * uu[1]);
                }
                memcpy(stt, nxt, sizeof(stt));
            }
            int pn = 0;
            for (int k = 0; k <= N; k++) for (int a = 0; a < nx; a++) {
                const int col = k * nx + a; Pp[col] = pn;
                for (int r = 0; r <= a; r++) { const double v = 2.0 * Q[r * nx + a]; if (v != 0.0) { Pi[pn] = k * nx + r; Px[pn++] = v; } }
                q[col] = Lcf[a];
            }
            for (int k = 0; k < N; k++) for (int j = 0; j < nu; j++) {
                const int col = (N + 1) * nx + k * nu + j; Pp[col] = pn;
                if (k >= 1) { const double v = -2.0 * dR[j]; if (v != 0.0) { Pi[pn] = col - nu; Px[pn++] = v; } }
                for (int r = 0; r <= j; r++) {
                    double v = 2.0 * R[r * nu + j]; if (r == j) v += 2.0 * (k < N - 1 ? 2.0 : 1.0) * dR[j];
                    if (v != 0.0) { Pi[pn] = (N + 1) * nx + k * nu + r; Px[pn++] = v; }
                }
                q[col] = (k == 0) ? -2.0 * (u_old ? u_old[(size_t)b * 2 + j] : 0.0) * dR[j] : 0.0;
            }
            Pp[nz] = pn;
            int an = 0;
            for (int k = 0; k <= N; k++) for (int a = 0; a < nx; a++) {
                const int col = k * nx + a; Ap[col] = an;
                Ai[an] = k * nx + a; Ax[an++] = 1.0;
                if (k < N) for (int r = 0; r < nx; r++) { const double v = -AB[k * 35 + r * 7 + a]; if (v != 0.0) { Ai[an] = (k + 1) * nx + r; Ax[an++] = v; } }
                Ai[an] = me + col; Ax[an++] = 1.0;
            }
            for (int k = 0; k < N; k++) for (int j = 0; j < nu; j++) {
                const int col = (N + 1) * nx + k * nu + j; Ap[col] = an;
                for (int r = 0; r < nx; r++) { const double v = -AB[k * 35 + r * 7 + nx + j]; if (v != 0.0) { Ai[an] = (k + 1) * nx + r; Ax[an++] = v; } }
                Ai[an] = me + col; Ax[an++] = 1.0;
            }
            Ap[nz] = an;
            for (int r = 0; r < me; r++) l[r] = u[r] = (r < nx) ? x0[(size_t)b * 5 + r] : 0.0;
            const double mey = max_ey[b];
            const double xmin[5] = {min_vel, -1, -2, -mey, -0.8}, xmax[5] = {max_vel, 1, 2, mey, 0.8};
            const double umin[2] = {-0.249, -0.7}, umax[2] = {0.249, 2.0};
            for (int k = 0; k <= N; k++) for (int a = 0; a < nx; a++) { l[me + k * nx + a] = xmin[a]; u[me + k * nx + a] = xmax[a]; }
            for (int k = 0; k < N; k++) for (int j = 0; j < nu; j++) { l[me + (N + 1) * nx + k * nu + j] = umin[j]; u[me + (N + 1) * nx + k * nu + j] = umax[j]; }
            osqp_ref_info info;
            if (osqp_ref_solve_ws(nz, m, Pp, Pi, Px, q, Ap, Ai, Ax, l, u, perm, &st, 0, 0, xo, yo, &info) != 0) {
                /* the KKT factorisation broke down (seen with a diverged roll-out, |A| ~ 1e65): no answer from the oracle */
                for (int i = 0; i < nz; i++) xo[i] = NAN;
                info.status_val = -10; info.iter = 0;
            }
            memcpy(xPred + (size_t)b * (N + 1) * nx, xo, sizeof(double) * (N + 1) * nx);
            memcpy(uPred + (size_t)b * N * nu, xo + (N + 1) * nx, sizeof(double) * N * nu);
            if (status) status[b] = info.status_val;
            if (iters) iters[b] = info.iter;
        }
        free(AB); free(Pp); free(Pi); free(Px); free(q); free(Ap); free(Ai); free(Ax); free(l); free(u); free(xo); free(yo);
    }
    free(perm);
    return 0;
}
