"""Restatement of the reference's sparse (non-condensed) MPC formulation -- TEST
INFRASTRUCTURE ONLY.

SparseCMPC::run (src/MPC_Ctrl/SparseCMPC.cpp:31-73): variables = the 12 states of every horizon step
followed by one 3-force block per stance (step, foot); equality rows = the discrete dynamics, inequality
rows = force limits and the friction pyramid; cost = 1/2 sum w (x - x_des)^2 + 1/2 alpha |u|^2.  The QP is
handed to the reference's vendored OSQP 0.5.0 (OsqpTriples.cpp:57-142), which oracle/_ref/libosqp_ref.so
IS (compiled unmodified from /root/reference/src/osqp; oracle/osqp_shim.c drives it with the reference's
settings).  The builder below (numpy, fp64 like the reference's doubles) is a restatement -- SparseCMPC.cpp
needs Eigen -- function by function:
    buildX0 :78-81, buildCT :87-135 (+ ori::coordinateRotation / quatToRPY / crossMatrix,
    Utilities/orientation_tools.h:59-87,195-208), buildDT :140-154 + c2d (SparseCMPC_Math.cpp:6-28),
    addX0Constraint :187-226, addDynamicsConstraints :228-278, addForceConstraints :280-291,
    addFrictionConstraints :293-322, addQuadraticStateCost :324-332, addLinearStateCost :343-358,
    addQuadraticControlCost :334-341, getResult :386-407.
Defaults are ConvexMPCLocomotion::initSparseMPC's (ConvexMPCLocomotion.cpp:732-756).
"""
import ctypes as C
import os

import numpy as np
import scipy.linalg
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
SPARSE_WEIGHTS = np.array([0.25, 0.25, 10, 2, 2, 20, 0, 0, 0.3, 0.2, 0.2, 0.2])   # :747
SPARSE_MU, SPARSE_ALPHA, SPARSE_FMAX = 1.0, 4e-5, 120.0                          # :752-753, :736
_lib = None


def quat_to_rpy(q):
    """ori::quatToRPY, orientation_tools.h:195-208 (roll, pitch, yaw)."""
    q = [float(x) for x in q]
    as_ = min(-2. * (q[1] * q[3] - q[0] * q[2]), .99999)
    yaw = np.arctan2(2 * (q[1] * q[2] + q[0] * q[3]), q[0] ** 2 + q[1] ** 2 - q[2] ** 2 - q[3] ** 2)
    pitch = np.arcsin(as_)
    roll = np.arctan2(2 * (q[2] * q[3] + q[0] * q[1]), q[0] ** 2 - q[1] ** 2 - q[2] ** 2 + q[3] ** 2)
    return np.array([roll, pitch, yaw])


def build(p, v, q, w, feet, contacts, traj, dts, weights=SPARSE_WEIGHTS, alpha=SPARSE_ALPHA, mu=SPARSE_MU,
          f_max=SPARSE_FMAX, mass=9.0, ibody=(0.07, 0.26, 0.242)):
    """-> dict(P, q, A, l, u (dense / vectors), T, blocks=[(foot, step)], Ad, Bd).
    feet[foot*3 + axis] = pFoot - position (setFeet :65-67); contacts[T][4]; traj[T][12]; dts[T]."""
    T = len(dts)
    rpy0 = quat_to_rpy(q)                                            # buildX0
    x0 = np.concatenate([rpy0, p, w, v]).astype(np.float64)
    c, s = np.cos(rpy0[2]), np.sin(rpy0[2])
    Ryaw = np.array([[c, s, 0], [-s, c, 0], [0, 0, 1.0]])            # coordinateRotation(Z, yaw)
    Iinv = np.linalg.inv(Ryaw.T @ np.diag(ibody) @ Ryaw)            # buildCT :103-104
    Act = np.zeros((12, 12))
    Act[3, 9] = Act[4, 10] = Act[5, 11] = 1
    Act[0:3, 6:9] = Ryaw
    blocks, Bct = [], []
    for i in range(T):
        for foot in range(4):
            if contacts[i][foot]:
                pf = np.asarray(feet[3 * foot:3 * foot + 3], np.float64)
                cm = np.array([[0, -pf[2], pf[1]], [pf[2], 0, -pf[0]], [-pf[1], pf[0], 0]])
                B = np.zeros((12, 3))
                B[6:9] = Iinv @ cm
                B[9:12] = np.eye(3) / mass
                blocks.append((foot, i))
                Bct.append(B)
    # buildDT / c2d: A <- expm(AB * dt)[0:12, 0:12], every B block <- B * dt (NOT the matching expm block)
    Ad, Bd = [], []
    bi = 0
    for i in range(T):
        AB = np.zeros((24, 24))
        AB[:12, :12] = Act
        for k in range(bi, len(blocks)):
            if blocks[k][1] != i:
                break
            AB[:12, 12 + 3 * blocks[k][0]:15 + 3 * blocks[k][0]] = Bct[k]
        Ad.append(scipy.linalg.expm(AB * dts[i])[:12, :12])
        while bi < len(blocks) and blocks[bi][1] == i:
            Bd.append(Bct[bi] * dts[i])
            bi += 1
    nb = len(blocks)
    nvar = 12 * T + 3 * nb
    g = np.zeros(12)
    g[11] = -9.81                                                    # run() :41-42
    rows_A, lo, hi = [], [], []

    def new_rows(k):
        r = [np.zeros(nvar) for _ in range(k)]
        rows_A.extend(r)
        return r
    first = {i: min([k for k in range(nb) if blocks[k][1] == i], default=None) for i in range(T)}
    for i in range(T):                                               # addX0Constraint + addDynamicsConstraints
        r = new_rows(12)
        for j in range(12):
            r[j][12 * i + j] = 1.0
        if i > 0:
            for a in range(12):
                r[a][12 * (i - 1):12 * i] -= Ad[i][a]
        for k in range(nb):
            if blocks[k][1] == i:
                for a in range(12):
                    r[a][12 * T + 3 * k:12 * T + 3 * k + 3] -= Bd[k][a]
        rhs = (Ad[0] @ x0 if i == 0 else 0.0) + g * dts[i]
        lo.extend(rhs)
        hi.extend(rhs)
    for k in range(nb):                                              # addForceConstraints
        r = new_rows(1)
        r[0][12 * T + 3 * k + 2] = 1.0
        lo.append(0.0)
        hi.append(f_max)
    mi = 1.0 / mu
    for k in range(nb):                                              # addFrictionConstraints
        r = new_rows(4)
        cidx = 12 * T + 3 * k
        for t, (ax, sg) in enumerate(((0, mi), (0, -mi), (1, mi), (1, -mi))):
            r[t][cidx + ax] = sg
            r[t][cidx + 2] = 1.0
        lo.extend([0.0] * 4)
        hi.extend([1e15] * 4)
    Pd = np.zeros(nvar)
    Pd[:12 * T] = np.tile(weights, T)                                # addQuadraticStateCost
    Pd[12 * T:] = alpha                                              # addQuadraticControlCost
    qv = np.zeros(nvar)
    qv[:12 * T] = -(np.asarray(traj, np.float64).reshape(T, 12) * weights).reshape(-1)   # addLinearStateCost
    return {"P": np.diag(Pd), "q": qv, "A": np.array(rows_A), "l": np.array(lo), "u": np.array(hi), "T": T,
            "blocks": blocks, "Ad": Ad, "Bd": Bd, "x0": x0, "g": g}


def _osqp_lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "_ref", "libosqp_ref.so"))
        _lib.osqp_ref_solve.restype = C.c_longlong
    return _lib


def osqp_prepare(P, q, A, l, u):
    """The QP in the compressed-sparse-column arrays OSQP takes (what SparseCMPC's triplets become, OsqpTriples.cpp:62-93),
    built once so that osqp_call times the reference's solver alone."""
    Pc = sp.csc_matrix(sp.triu(sp.csc_matrix(P)))
    Ac = sp.csc_matrix(A)
    arr = lambda a, t: np.ascontiguousarray(a, t)
    return {"n": q.size, "m": l.size,
            "Px": arr(Pc.data, np.float64), "Pi": arr(Pc.indices, np.int64), "Pp": arr(Pc.indptr, np.int64),
            "Ax": arr(Ac.data, np.float64), "Ai": arr(Ac.indices, np.int64), "Ap": arr(Ac.indptr, np.int64),
            "q": arr(q, np.float64), "l": arr(l, np.float64), "u": arr(u, np.float64), "x": np.zeros(q.size)}


def osqp_call(c, eps=1e-5, max_iter=0, polish=False):
    """osqp_setup + osqp_solve + osqp_cleanup of the reference's vendored OSQP 0.5.0 on a prepared QP (every MPC cycle of
    the reference sets the solver up anew, OsqpTriples.cpp:57-142).  -> (x, status, iterations)."""
    lib = _osqp_lib()
    it = C.c_longlong(0)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    st = lib.osqp_ref_solve(C.c_longlong(c["n"]), C.c_longlong(c["m"]), C.c_longlong(c["Px"].size), ptr(c["Px"]), ptr(c["Pi"]),
                            ptr(c["Pp"]), ptr(c["q"]), C.c_longlong(c["Ax"].size), ptr(c["Ax"]), ptr(c["Ai"]), ptr(c["Ap"]),
                            ptr(c["l"]), ptr(c["u"]), C.c_double(eps), C.c_double(eps), C.c_longlong(max_iter),
                            C.c_int(1 if polish else 0), ptr(c["x"]), C.byref(it))
    return c["x"], int(st), int(it.value)


def osqp(P, q, A, l, u, eps=1e-5, max_iter=0, polish=False):
    """The reference's vendored OSQP 0.5.0 (oracle/_ref/libosqp_ref.so), settings of OsqpTriples.cpp:95-103."""
    x, st, it = osqp_call(osqp_prepare(P, q, A, l, u), eps, max_iter, polish)
    return x.copy(), st, it


def first_step_forces(x, prob):
    """getResult :386-407: the 12 first-step forces, foot-major, zeros for feet not in contact at step 0."""
    out = np.zeros(12)
    T = prob["T"]
    for k, (foot, step) in enumerate(prob["blocks"]):
        if step == 0:
            out[3 * foot:3 * foot + 3] = x[12 * T + 3 * k:12 * T + 3 * k + 3]
    return out


def condensed(prob, weights=SPARSE_WEIGHTS, alpha=SPARSE_ALPHA, traj=None):
    """The same QP with the states eliminated through the dynamics rows: (H, g) in the force blocks only,
    min 1/2 u^T H u + g^T u.  Independent route to the exact minimiser (the constraints on u are untouched)."""
    T, blocks, Ad, Bd = prob["T"], prob["blocks"], prob["Ad"], prob["Bd"]
    nb = len(blocks)
    G = np.zeros((12 * T, 3 * nb))
    free = np.zeros((T, 12))
    x = prob["x0"].copy()
    Gk = np.zeros((12, 3 * nb))
    for i in range(T):
        x = Ad[i] @ x + prob["g"] * prob["dts"][i]
        Gk = Ad[i] @ Gk
        for k in range(nb):
            if blocks[k][1] == i:
                Gk[:, 3 * k:3 * k + 3] += Bd[k]
        free[i] = x
        G[12 * i:12 * i + 12] = Gk
    Wd = np.tile(weights, T)
    H = G.T @ (Wd[:, None] * G) + alpha * np.eye(3 * nb)
    gv = G.T @ (Wd * (free.reshape(-1) - np.asarray(traj, np.float64).reshape(-1)))
    return H, gv


def from_batch(b, i, weights=SPARSE_WEIGHTS, alpha=SPARSE_ALPHA, mu=SPARSE_MU, f_max=SPARSE_FMAX):
    """Instance i of a workloads batch dict as solveSparseMPC feeds SparseCMPC (ConvexMPCLocomotion.cpp:689-730)."""
    h = b["horizon"]
    feet = b["r"][i].astype(np.float64).reshape(3, 4).T.reshape(-1)          # axis-major -> foot-major
    contacts = b["gait"][i].reshape(h, 4)
    prob = build(b["p"][i].astype(np.float64), b["v"][i].astype(np.float64), b["q"][i].astype(np.float64),
                 b["w"][i].astype(np.float64), feet, contacts, b["traj"][i].astype(np.float64).reshape(h, 12),
                 [float(np.float32(b["dt"]))] * h, weights, alpha, mu, f_max)
    prob["dts"] = [float(np.float32(b["dt"]))] * h
    return prob
