"""numpy restatement of the reference's JCQP alternate -- TEST INFRASTRUCTURE ONLY.

QpProblem<double>::runFromDense (src/JCQP/QpProblem.cpp:178-269) as solve_mpc drives it when
update_solver_settings' use_jcqp is 1 (full 12h x 20h problem, SolverMPC.cpp:400-414) or 2 (the
swing-eliminated problem, :558-610): an OSQP-style ADMM with a constant KKT matrix.

PARITY UNPINNED by reference execution: src/JCQP needs Eigen (JCQP/types.h:5-6).  Restated function by
function; the reference's dense pivoted LDL^T solve of the KKT system (CholeskyDenseSolver.cpp:19-156) is
replaced by numpy's LU solve of the SAME matrix (identical in exact arithmetic).
"""
import numpy as np

INFTY, EQL_TOL, RHO_EQ_SCALE, RHO_INFTY = 1e10, 1e-10, 1e3, 1e-6     # QpProblem.h:22-25
BIG = float(np.float32(5e10))                                       # SolverMPC.cpp:15, stored in float


def constraint_rho(l, u, rho):
    """computeConstraintInfos, QpProblem.cpp:276-291."""
    r = np.full(l.size, rho)
    r[np.abs(u - l) < EQL_TOL] = rho * RHO_EQ_SCALE
    r[(l < -INFTY) | (u > INFTY)] = RHO_INFTY
    return r


def run_from_dense(P, q, A, l, u, max_iter, rho, sigma, alpha, terminate):
    """-> (x, iterations run, last residual).  QpProblem.cpp:178-269, :294-407 (cold start: x = z = y = 0)."""
    n, m = q.size, l.size
    R = constraint_rho(l, u, rho)
    kkt = np.zeros((n + m, n + m))
    kkt[:n, :n] = P + sigma * np.eye(n)             # setupLinearSolverCommon :294-306
    kkt[:n, n:] = A.T
    kkt[n:, :n] = A
    kkt[n:, n:] = -np.diag(1.0 / R)
    lu = np.linalg.inv(kkt)                         # constant matrix: factor once
    x, z, y = np.zeros(n), np.zeros(m), np.zeros(m)
    resid, it = np.inf, 0
    for it in range(1, max_iter + 1):
        x_prev, z_prev = x, z                        # stepSetup :308-313
        rhs = np.concatenate([sigma * x_prev - q, z_prev - y / R])      # solveLinearSystem :315-338
        sol = lu @ rhs
        xt = sol[:n]
        zt = z_prev + (sol[n:] - y) / R
        x = alpha * xt + (1 - alpha) * x_prev        # stepX :340-347
        zr = alpha * zt + (1 - alpha) * z_prev
        z = np.clip(zr + y / R, l, u)                # stepZ :349-358
        y = y + R * (zr - z)                         # stepY :360-367
        if it % 10 == 0:                             # :238-247, calcAndDisplayResidual :381-407
            p = np.abs(A @ x - z_prev).max()
            d = np.abs(P @ x + q + A.T @ y).max()
            resid = (d + p) / 4
            if resid < terminate or it >= max_iter:
                break
    return x, it, resid


def mpc_problem(b, i, mode):
    """(P, q, A, l, u, var_index) of instance i as solve_mpc hands it to JCQP: mode 1 = the full
    problem (SolverMPC.cpp:400-407), mode 2 = after the swing elimination (:558-590).  Built from the
    oracle assembly (float-assembled, like the reference)."""
    from . import oracle as O
    H, g, A, lb, ub, _ = O.assemble(b, i)
    if mode == 1:
        return H, g, A, lb, ub, np.arange(g.size)
    ve, Hr, gr, Ar, lr, ur = O.reduce(H, g, A, lb, ub)
    return Hr, gr, Ar, lr, ur, np.flatnonzero(~ve)


def solve(b, i, mode, max_iter=10000, rho=1e-7, sigma=1e-8, alpha=1.5, terminate=0.1):
    """q_soln[12h] like solve_mpc with use_jcqp = mode (defaults: the caller's settings,
    ConvexMPCLocomotion.cpp:644-648)."""
    P, q, A, l, u, vi = mpc_problem(b, i, mode)
    x, it, resid = run_from_dense(P, q, A, l, u, max_iter, rho, sigma, alpha, terminate)
    out = np.zeros(12 * b["horizon"])
    out[vi] = x
    return out, it, resid
