"""Dense QP matrices of the two MPC classes, as the reference leaves them on its objects.

The HIP solve path never materialises the QP (csrc/admm_solve.hip keeps the stage structure); the reference's callers,
however, can read ``Controller.G / .E / .L / .Eu / .M / .q / .F / .b`` (CTRL:79,108-110) and ``Planner.Aeq / .E / .L / .Eu``
(PLAN:108) after a ``solve``.  The drop-in classes expose the same attributes as lazy properties that are assembled HERE, on
the host, from the stage blocks and weights of the last solve -- only when somebody asks for them.

Decision vector z = [x_0 .. x_N, u_0 .. u_{N-1}] (CTRL:479-482), nz = (N+1) n + N d.
"""
import numpy as np

INF = np.inf


def slew_rate_hessian(R, dR, N):
    """Input block Mu of the cost (CTRL:401-425 / PLAN:148-160): N blocks R + 2 diag(dR) on the diagonal, the last one with one
    dR less, and -dR on the second off-diagonals (u_k - u_{k-1} coupling)."""
    R = np.asarray(R, float); dR = np.asarray(dR, float)
    d = R.shape[0]
    Mu = np.zeros((N * d, N * d))
    for k in range(N):
        blk = R + 2.0 * np.diag(dR)
        if k == N - 1:
            blk = blk - np.diag(dR)
        Mu[k * d:(k + 1) * d, k * d:(k + 1) * d] = blk
    idx = np.arange((N - 1) * d)
    Mu[idx + d, idx] = -np.tile(dR, N - 1)
    Mu[idx, idx + d] = -np.tile(dR, N - 1)
    return Mu


def cost_hessian(Q, R, dR, N):
    """M0 = blkdiag(Q x (N+1), Mu) (CTRL:397-432); the QP Hessian is 2 M0."""
    Q = np.asarray(Q, float)
    n = Q.shape[0]
    d = np.asarray(R).shape[0]
    M0 = np.zeros(((N + 1) * n + N * d,) * 2)
    for k in range(N + 1):
        M0[k * n:(k + 1) * n, k * n:(k + 1) * n] = Q
    M0[(N + 1) * n:, (N + 1) * n:] = slew_rate_hessian(R, dR, N)
    return M0


def equality_blocks(A, B, C, N, n, d, steer_hist=()):
    """G z = E x0 + L (+ Eu uOld, which the reference discards: quirk Q1) -- CTRL:477-529 / PLAN:434-486.
    ``steer_hist`` = OldSteering[1 .. delay]: one pinned-steering row per entry (CTRL:518-527)."""
    delay = len(steer_hist)
    rows = n * (N + 1)
    G = np.zeros((rows + delay, rows + N * d))
    G[:rows, :rows] = np.eye(rows)
    L = np.zeros((rows + delay, 1))
    for i in range(N):
        r = slice(n + i * n, n + (i + 1) * n)
        G[r, i * n:(i + 1) * n] = -np.asarray(A[i], float)
        G[r, rows + i * d:rows + (i + 1) * d] = -np.asarray(B[i], float)
        L[r, :] = np.asarray(C[i], float).reshape(n, 1)
    for i in range(delay):
        G[rows + i, rows + i * d] = 1.0
        L[rows + i, 0] = steer_hist[i]
    E = np.zeros((rows + delay, n))
    E[:n, :n] = np.eye(n)
    Eu = np.zeros((rows + delay, d))
    return G, E, L, Eu


def controller_inequalities(N, n, d, max_vel, vx_min=0.01, delta_max=0.249, a_max=4.0, a_min_abs=1.0):
    """F z <= b (CTRL:329-378): per stage k < N  -vx <= -vx_min, vx <= max_vel (all state rows first), then per input
    delta <= dmax, -delta <= dmax, a <= amax, -a <= amin; no terminal-state rows."""
    nz = (N + 1) * n + N * d
    F = np.zeros((6 * N, nz))
    b = np.zeros(6 * N)
    for k in range(N):
        F[2 * k, k * n] = -1.0; F[2 * k + 1, k * n] = 1.0
        b[2 * k], b[2 * k + 1] = -vx_min, max_vel
        c0 = (N + 1) * n + k * d
        r0 = 2 * N + 4 * k
        F[r0, c0] = 1.0; F[r0 + 1, c0] = -1.0; F[r0 + 2, c0 + 1] = 1.0; F[r0 + 3, c0 + 1] = -1.0
        b[r0:r0 + 4] = (delta_max, delta_max, a_max, a_min_abs)
    return F, b


def controller_qp(Q, R, dR, N, A, B, C, x0, u_old, vel_ref, max_vel, steer_hist=(), bounds=None):
    """Everything the reference controller leaves on the object after ``solve`` plus the OSQP form it hands to the solver
    (CTRL:303-308: inequalities first, equalities last).  ``vel_ref``: at least N entries, the last one is the terminal target
    (CTRL:434-438)."""
    Q = np.asarray(Q, float)
    n, d = Q.shape[0], np.asarray(R).shape[0]
    vr = np.asarray(vel_ref, float).reshape(-1)
    M0 = cost_hessian(Q, R, dR, N)
    xtrack = np.zeros((N + 1) * n + N * d)
    xtrack[np.arange(N) * n] = vr[:N]
    xtrack[N * n] = vr[-1]
    q = -2.0 * (xtrack @ M0)
    q[(N + 1) * n:(N + 1) * n + 2] = -2.0 * np.asarray(u_old, float)[:2] * np.asarray(dR, float)      # CTRL:462
    G, E, L, Eu = equality_blocks(A, B, C, N, n, d, steer_hist)
    F, b = controller_inequalities(N, n, d, max_vel, **(bounds or {}))
    beq = E @ np.asarray(x0, float).reshape(n) + L[:, 0]                                               # CTRL:148 (quirk Q1)
    return dict(M=2.0 * M0, q=q, G=G, E=E, L=L, Eu=Eu, F=F, b=b,
                P=2.0 * M0, A=np.vstack((F, G)), l=np.concatenate((np.full(len(b), -INF), beq)), u=np.concatenate((b, beq)))


def planner_qp(Q, R, dR, L_cf, N, A, B, C, x0, u_old, max_ey, min_vel, max_vel, xbox=None, ubox=None):
    """Planner: ``Aeq, E, L, Eu`` (PLAN:108) and the OSQP form of PLAN:145-202 (equalities first, then the identity block with
    the state box on ALL N+1 stages and the input box)."""
    Q = np.asarray(Q, float)
    n, d = Q.shape[0], np.asarray(R).shape[0]
    M0 = cost_hessian(Q, R, dR, N)
    q = np.concatenate((np.tile(np.asarray(L_cf, float), N + 1), np.zeros(N * d)))
    q[(N + 1) * n:(N + 1) * n + 2] = -2.0 * np.asarray(u_old, float)[:2] * np.asarray(dR, float)      # PLAN:167
    Aeq, E, L, Eu = equality_blocks(A, B, C, N, n, d)
    beq = E @ np.asarray(x0, float).reshape(n) + L[:, 0]                                               # PLAN:115 (quirk Q1)
    xlo, xhi = xbox if xbox is not None else (np.array([min_vel, -1.0, -2.0, -max_ey, -0.8]), np.array([max_vel, 1.0, 2.0, max_ey, 0.8]))
    ulo, uhi = ubox if ubox is not None else (np.array([-0.249, -0.7]), np.array([0.249, 2.0]))
    lineq = np.concatenate((np.tile(xlo, N + 1), np.tile(ulo, N)))
    uineq = np.concatenate((np.tile(xhi, N + 1), np.tile(uhi, N)))
    nz = (N + 1) * n + N * d
    return dict(Aeq=Aeq, E=E, L=L, Eu=Eu, P=2.0 * M0, q=q, A=np.vstack((Aeq, np.eye(nz))),
                l=np.concatenate((beq, lineq)), u=np.concatenate((beq, uineq)))
