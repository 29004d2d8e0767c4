"""Oracle (TEST INFRASTRUCTURE): solver-independent KKT optimality certificate.

For   min 1/2 x'Px + q'x   s.t.  l <= Ax <= u   a pair (x, y) is optimal iff
  stationarity    P x + q + A'y = 0
  feasibility     l <= A x <= u
  dual sign       y_i > 0 only where (Ax)_i = u_i,  y_i < 0 only where (Ax)_i = l_i.
``kkt_residuals`` measures the three; ``active_set_optimum`` computes a
high-accuracy optimum with dense numpy linear algebra (primal-dual active-set
iteration started from a guess) so that tests can compare a solver's output
with x* without trusting any ADMM code.
"""
from __future__ import annotations

import numpy as np


def kkt_residuals(P, q, A, l, u, x, y):
    """(stationarity, primal infeasibility, complementarity) in the inf-norm."""
    P = np.asarray(P, float); A = np.asarray(A, float)
    Ax = A @ x
    stat = np.max(np.abs(P @ x + q + A.T @ y)) if x.size else 0.0
    prim = max(0.0, float(np.max(np.maximum(l - Ax, 0.0), initial=0.0)),
               float(np.max(np.maximum(Ax - u, 0.0), initial=0.0)))
    du = np.where(np.isfinite(u), np.abs(u - Ax), np.inf)
    dl = np.where(np.isfinite(l), np.abs(Ax - l), np.inf)
    comp = np.where(y > 0, np.minimum(np.abs(y), du), np.where(y < 0, np.minimum(np.abs(y), dl), 0.0))
    return float(stat), float(prim), float(np.max(comp, initial=0.0))


def active_set_optimum(P, q, A, l, u, x0=None, y0=None, tol=1e-9, max_rounds=200):
    """Primal-dual active-set iteration with dense least-squares KKT solves.

    Returns (x, y, rounds).  Raises RuntimeError if it does not settle."""
    P = np.asarray(P, float); A = np.asarray(A, float)
    n = P.shape[0]; m = A.shape[0]
    eq = np.isfinite(l) & np.isfinite(u) & (np.abs(u - l) <= 1e-12)
    # working set: +1 upper, -1 lower, 0 inactive ; equalities always "upper"
    W = np.zeros(m, dtype=int)
    if y0 is not None and x0 is not None:
        Ax = A @ x0
        W[(y0 > 1e-7) | (np.isfinite(u) & (Ax >= u - 1e-7) & (y0 > 0))] = 1
        W[(y0 < -1e-7) | (np.isfinite(l) & (Ax <= l + 1e-7) & (y0 < 0))] = -1
    W[eq] = 1
    x = np.zeros(n) if x0 is None else np.array(x0, float)
    for rnd in range(max_rounds):
        idx = np.nonzero(W)[0]
        b = np.where(W[idx] > 0, u[idx], l[idx])
        Aw = A[idx]
        k = idx.size
        K = np.zeros((n + k, n + k))
        K[:n, :n] = P
        K[:n, n:] = Aw.T
        K[n:, :n] = Aw
        rhs = np.concatenate([-q, b])
        sol = np.linalg.lstsq(K, rhs, rcond=None)[0]
        # two steps of iterative refinement
        for _ in range(2):
            sol = sol + np.linalg.lstsq(K, rhs - K @ sol, rcond=None)[0]
        x = sol[:n]; lam = sol[n:]
        y = np.zeros(m); y[idx] = lam
        Ax = A @ x
        viol_u = np.where(W == 0, Ax - u, -np.inf)
        viol_l = np.where(W == 0, l - Ax, -np.inf)
        bad_sign = np.zeros(m)
        ine = ~eq
        bad_sign[(W > 0) & ine] = -y[(W > 0) & ine]      # upper-active needs y >= 0
        bad_sign[(W < 0) & ine] = y[(W < 0) & ine]       # lower-active needs y <= 0
        worst_p = max(viol_u.max(initial=-np.inf), viol_l.max(initial=-np.inf))
        worst_d = bad_sign.max(initial=-np.inf)
        if worst_p <= tol and worst_d <= tol:
            return x, y, rnd + 1
        if worst_d > tol and worst_d >= worst_p:
            W[int(np.argmax(bad_sign))] = 0
        elif viol_u.max(initial=-np.inf) >= viol_l.max(initial=-np.inf):
            W[int(np.argmax(viol_u))] = 1
        else:
            W[int(np.argmax(viol_l))] = -1
    raise RuntimeError("active_set_optimum did not settle in %d rounds" % max_rounds)


def osqp_termination_ok(P, q, A, l, u, x, y, eps_abs=1e-3, eps_rel=1e-3):
    """OSQP's own stopping rule (paper section 3.4) evaluated on unscaled data with
    z = clip(Ax, l, u):  ||Ax - z|| <= eps_abs + eps_rel max(||Ax||, ||z||) and
    ||Px + q + A'y|| <= eps_abs + eps_rel max(||Px||, ||A'y||, ||q||)."""
    P = np.asarray(P, float); A = np.asarray(A, float)
    Ax = A @ x
    z = np.clip(Ax, l, u)
    Px = P @ x; Aty = A.T @ y
    inf = lambda v: float(np.max(np.abs(v), initial=0.0))
    pri = inf(Ax - z); dua = inf(Px + q + Aty)
    ep = eps_abs + eps_rel * max(inf(Ax), inf(z))
    ed = eps_abs + eps_rel * max(inf(Px), inf(Aty), inf(q))
    return pri <= ep and dua <= ed, dict(pri=pri, dua=dua, eps_pri=ep, eps_dua=ed)
