"""fp64 numpy model of the structure-exploiting formulation -- TEST
INFRASTRUCTURE ONLY (lives under oracle/; never imported by the product).

Purpose: an independent, readable derivation check that sits between the
dense float restatement (mpc_oracle.c, which mirrors SolverMPC.cpp line by
line) and the HIP kernels (which exploit the same identities):

  A_ct^3 = 0  =>  Adt^d * Bdt = dt*B0 + c_d*B1 + e_d*B2,
                  B0 = B, B1 = A B, B2 = A^2 B,
                  c_d = (2d+1) dt^2/2,  e_d = ((d+1)^3 - d^3) dt^3/6
  =>  qH = 2 * sum_{p,q} C_pq (x) (B_p^T W B_q) + 2 alpha I,
      C_pq[i,j] = sum_{k>=max(i,j)} coef_p(k-i) coef_q(k-j)   (h x h, depends
      only on h and dt -- a constant table for the whole batch)
  =>  qg_i = 2 * sum_p B_p^T ( sum_{k>=i} coef_p(k-i) W (Adt^{k+1} x0 - xd_k) )

plus a dense dual active-set (Goldfarb-Idnani, Schur-complement form on the
explicit inverse) QP solve that the HIP solver's control flow follows.
"""
import numpy as np


def quat_to_rpy(q):
    w, x, y, z = [np.float64(v) for v in q]
    as_ = min(-2. * (x * z - w * y), .99999)
    yaw = np.arctan2(2 * (x * y + w * z), w * w + x * x - y * y - z * z)
    pitch = np.arcsin(as_)
    roll = np.arctan2(2 * (y * z + w * x), w * w - x * x - y * y + z * z)
    return roll, pitch, yaw


def ct_mats(r, yaw, x_drag, mass=9.0, ibody=(.07, .26, .242), trig=None):
    yaw = np.float64(yaw)
    c, s = (np.cos(yaw), np.sin(yaw)) if trig is None else (np.float64(trig[0]), np.float64(trig[1]))
    Ry = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])
    Ib = np.diag(np.array(ibody, np.float32).astype(np.float64))
    Iinv = np.linalg.inv(Ry @ Ib @ Ry.T)
    A = np.zeros((13, 13))
    A[3, 9] = A[4, 10] = A[5, 11] = 1
    A[11, 9] = x_drag
    A[11, 12] = 1
    A[0:3, 6:9] = Ry.T
    B = np.zeros((13, 12))
    rr = np.asarray(r, np.float64).reshape(3, 4)
    for b in range(4):
        rx, ry, rz = rr[:, b]
        cm = np.array([[0, -rz, ry], [rz, 0, -rx], [-ry, rx, 0]])
        B[6:9, 3 * b:3 * b + 3] = Iinv @ cm
        B[9:12, 3 * b:3 * b + 3] = np.eye(3) / mass
    return A, B


def coef_tables(h, dt):
    """C[p][q] (h x h) and the per-lag coefficient vectors coef[p][d]."""
    d = np.arange(h, dtype=np.float64)
    coef = [np.full(h, dt), (2 * d + 1) * dt * dt / 2,
            ((d + 1) ** 3 - d ** 3) * dt ** 3 / 6]
    C = [[np.zeros((h, h)) for _ in range(3)] for _ in range(3)]
    for p in range(3):
        for q in range(3):
            for i in range(h):
                for j in range(h):
                    k = np.arange(max(i, j), h)
                    C[p][q][i, j] = np.sum(coef[p][k - i] * coef[q][k - j])
    return coef, C


_TABLES = {}


def assemble(b, i, trig=None, rpy=None):
    """(H[12h,12h], g[12h]) in fp64 from instance i of batch dict b.
    trig = (cos yaw, sin yaw) and rpy = (roll, pitch, yaw) override the fp64
    transcendentals with externally evaluated ones (the reference and the GPU
    evaluate them in float: RobotState.cpp:30-35, SolverMPC.cpp:257-267), so that a
    comparison can pin the ALGEBRA to ~1e-13 independently of libm bits."""
    h = b["horizon"]
    dt = np.float64(np.float32(b["dt"]))
    A, B = ct_mats(b["r"][i], b["yaw"][i], np.float64(b["x_drag"][i]), trig=trig)
    Bp = [B, A @ B, A @ A @ B]
    W = np.zeros(13)
    W[:12] = b["weights"][i].astype(np.float64)
    if (h, dt) not in _TABLES:
        _TABLES[(h, dt)] = coef_tables(h, dt)
    coef, C = _TABLES[(h, dt)]
    n = 12 * h
    H = np.zeros((n, n))
    for p in range(3):
        for q in range(3):
            E = Bp[p].T @ (W[:, None] * Bp[q])
            H += np.kron(C[p][q], E)
    H = 2 * (H + np.float64(b["alpha"][i]) * np.eye(n))
    roll, pitch, yaw = quat_to_rpy(b["q"][i]) if rpy is None else [np.float64(v) for v in rpy]
    x0 = np.concatenate([[roll, pitch, yaw], b["p"][i], b["w"][i], b["v"][i],
                         [np.float64(np.float32(-9.8))]]).astype(np.float64)
    Ax, AAx = A @ x0, A @ A @ x0
    traj = b["traj"][i].astype(np.float64).reshape(h, 12)
    e = np.zeros((h, 13))
    for k in range(h):
        t = (k + 1) * dt
        e[k] = x0 + Ax * t + AAx * t * t / 2
        e[k, :12] -= traj[k]
        e[k] *= W
    g = np.zeros(n)
    for ii in range(h):
        for p in range(3):
            s = np.zeros(13)
            for k in range(ii, h):
                s += coef[p][k - ii] * e[k]
            g[12 * ii:12 * ii + 12] += 2 * Bp[p].T @ s
    return H, g


def sweep_inverse(H):
    """In-place symmetric sweep (Gauss-Jordan) -> H^-1; the GPU kernel's
    factorisation-free inversion."""
    A = H.copy()
    n = A.shape[0]
    for k in range(n):
        d = A[k, k]
        col = A[:, k].copy()
        A -= np.outer(col, col) / d
        A[:, k] = col / d
        A[k, :] = col / d
        A[k, k] = -1 / d
    return -A


def stance_constraints(gait, h, mu, f_max):
    """One-sided rows c^T x >= b in REDUCED variable indexing.  Per stance
    foot-step: 4 friction-pyramid rows, fz >= 0, -fz >= -f_max
    (SolverMPC.cpp:352-378 with the BIG_NUMBER uppers dropped as inactive)."""
    stance = [k for k in range(4 * h) if gait[k]]
    mi = np.float64(np.float32(1.0) / np.float32(mu))
    rows = []
    for c, _ in enumerate(stance):
        j = 3 * c
        rows += [((j, mi), (j + 2, 1.0), 0.0), ((j, -mi), (j + 2, 1.0), 0.0),
                 ((j + 1, mi), (j + 2, 1.0), 0.0), ((j + 1, -mi), (j + 2, 1.0), 0.0),
                 ((j + 2, 1.0), (j + 2, 0.0), 0.0),
                 ((j + 2, -1.0), (j + 2, 0.0), -np.float64(np.float32(f_max)))]
    return stance, rows


def dual_active_set(Hinv, g, rows, tol=1e-9, max_iter=500):
    """Goldfarb-Idnani dual active set in Schur-complement form.
    Returns (x, active list, iterations)."""
    n = g.size
    m = len(rows)
    Cm = np.zeros((m, n))
    bv = np.zeros(m)
    for r, (a, b2, rhs) in enumerate(rows):
        Cm[r, a[0]] += a[1]
        Cm[r, b2[0]] += b2[1]
        bv[r] = rhs
    scale = np.sqrt(np.einsum("ij,jk,ik->i", Cm, Hinv, Cm))  # ||c||_{H^-1}
    x = -Hinv @ g
    W, lam = [], []
    M = np.zeros((n, 0))
    it = 0
    while it < max_iter:
        s = Cm @ x - bv
        sn = s / scale
        sn[W] = 0
        p = int(np.argmin(sn))
        if sn[p] >= -tol * max(1.0, np.abs(x).max()):
            break
        lp = 0.0
        while True:
            it += 1
            hc = Hinv @ Cm[p]
            if W:
                S = Cm[W] @ M
                d = M.T @ Cm[p]
                r = np.linalg.solve(S, d)
                z = hc - M @ r
            else:
                r = np.zeros(0)
                z = hc
            delta = Cm[p] @ z
            dependent = delta <= 1e-12 * (Cm[p] @ hc)
            t2 = np.inf if dependent else -(Cm[p] @ x - bv[p]) / delta
            t1, l = np.inf, -1
            for k in range(len(W)):
                if r[k] > 0 and lam[k] / r[k] < t1:
                    t1, l = lam[k] / r[k], k
            t = min(t1, t2)
            if not np.isfinite(t):
                return x, W, -it  # infeasible
            if not dependent:
                x = x + t * z
            lam = [lam[k] - t * r[k] for k in range(len(W))]
            lp += t
            if t == t2:
                W.append(p)
                lam.append(lp)
                M = np.concatenate([M, hc[:, None]], 1)
                break
            W.pop(l)
            lam.pop(l)
            M = np.delete(M, l, 1)
    return x, W, it


def solve(b, i):
    """Full pipeline for instance i -> q_soln[12h] (zeros on swing)."""
    h = b["horizon"]
    H, g = assemble(b, i)
    stance, rows = stance_constraints(b["gait"][i], h, b["mu"], b["f_max"])
    vi = np.array([3 * k + a for k in stance for a in range(3)], int)
    out = np.zeros(12 * h)
    if vi.size == 0:
        return out, 0
    Hinv = sweep_inverse(H[np.ix_(vi, vi)])
    x, W, it = dual_active_set(Hinv, g[vi], rows)
    out[vi] = x
    return out, it
