"""numpy model of the PRIMAL-DUAL ACTIVE-SET pre-solver (round 5) -- TEST INFRASTRUCTURE ONLY.

The reference hands the reduced QP to qpOASES (SolverMPC.cpp:527-557), an active-set method that changes the working set
one row per iteration; so does the GPU's Goldfarb-Idnani engine, whose launch at batch 1024 waits for the one
robot with 13 - 15 such iterations.  A primal-dual active-set (PDAS) iteration changes the whole set at once:

    A+ = (A minus the rows whose multiplier came out negative)  union  (the most violated row of every stance foot-step),
    then x, lambda = the minimiser of the QP with the rows of A+ as EQUALITIES (one k x k solve with S = C_A H^-1 C_A^T).

It stops at a KKT point (no row violated, no multiplier negative) -- the unique minimiser, H > 0 -- typically after 2 - 5
solves where the one-row methods take 2 - 33.  Adding EVERY violated row (the textbook rule, `per_footstep=False`) cycles on
0.2 - 2 % of the robots -- the five rows of a foot-step fight each other: two faces of a pyramid join, two multipliers turn
negative, they leave, the faces are violated again --; adding at most ONE row per foot-step and solve (its most violated one)
has not cycled on any robot of any family tried (tests/test_oracle_cpu.py; a revisited set would still switch to single
changes, and whatever does not converge within `max_it` solves, or needs more than `kp` rows, is left to the Goldfarb-Idnani
engine from scratch: `ok = False`).  This file mirrors the kernel's
data flow (stance slot x 5 row types, type-major slot order, Gaussian elimination without pivoting that skips dependent
rows) so that the HIP code can be checked against it iteration by iteration; tests/test_oracle_cpu.py pins it against the
reference's qpOASES.
"""
import numpy as np

TYPES = 5


def rows_of(nst, mi, fmax):
    """(j1, j2, a1, a2, rhs) of row (s, ty): a1 x[j1] + a2 x[j2] >= rhs.  ty 0..3: the friction pyramid faces of f_block
    (SolverMPC.cpp:366-370), ty 4: fz <= f_max."""
    out = np.zeros((nst, TYPES, 5))
    for s in range(nst):
        for ty in range(4):
            out[s, ty] = (3 * s + ty // 2, 3 * s + 2, mi if ty % 2 == 0 else -mi, 1.0, 0.0)
        out[s, 4] = (3 * s + 2, 3 * s + 2, -1.0, 0.0, -fmax[s])
    return out


def solve(Hinv, g, nst, mi, fmax, tol=1e-9, max_it=12, kp=28, trace=None, per_footstep=True):
    """-> (x, lam[nst, 5], solves, ok, kmax)."""
    n = 3 * nst
    R = rows_of(nst, mi, np.broadcast_to(np.asarray(fmax, float), (nst,)))
    j1, j2 = R[..., 0].astype(int), R[..., 1].astype(int)
    a1, a2, rhs = R[..., 2], R[..., 3], R[..., 4]
    inv_fr = 1.0 / np.sqrt(mi * mi + 1.0)
    nrm = np.ones((nst, TYPES))
    nrm[:, :4] = inv_fr
    xu = -Hinv @ g
    x = xu.copy()
    act = np.zeros((nst, TYPES), bool)
    lam = np.zeros((nst, TYPES))
    seen, single, kmax = [], False, 0
    for it in range(max_it + 1):
        sv = (a1 * x[j1] + a2 * x[j2] - rhs) * nrm
        viol = (sv < -tol) & ~act
        neg = act & (lam < -1e-12)
        if trace is not None:
            trace.append((int(act.sum()), int(viol.sum()), int(neg.sum()), single))
        if not viol.any() and not neg.any():
            return x, lam, it, True, kmax
        if it == max_it:
            break
        if not single:
            add = viol
            if per_footstep:   # at most one row per foot-step joins: its most violated one
                add = np.zeros_like(viol)
                svv = np.where(viol, sv, np.inf)
                tb = svv.argmin(1)
                has = np.isfinite(svv.min(1))
                add[np.arange(nst)[has], tb[has]] = True
            new = (act & ~neg) | add
            key = new.tobytes()
            single = key in seen
            seen.append(key)
        if single:
            new = act.copy()
            if viol.any():
                s, ty = np.unravel_index(np.argmin(np.where(viol, sv, np.inf)), sv.shape)
                new[s, ty] = True
            else:
                s, ty = np.unravel_index(np.argmin(np.where(neg, lam, np.inf)), lam.shape)
                new[s, ty] = False
        act = new
        # slots, type-major (ty, s): the order the kernel's ballots give
        slots = [(s, ty) for ty in range(TYPES) for s in range(nst) if act[s, ty]]
        k = len(slots)
        kmax = max(kmax, k)
        if k > kp:
            break
        lam[:] = 0.0
        if k == 0:
            x = xu.copy()
            continue
        S = np.zeros((k, k))
        r = np.zeros(k)
        M = np.zeros((n, k))
        for a, (s, ty) in enumerate(slots):
            M[:, a] = a1[s, ty] * Hinv[:, j1[s, ty]] + a2[s, ty] * Hinv[:, j2[s, ty]]
            r[a] = rhs[s, ty] - (a1[s, ty] * xu[j1[s, ty]] + a2[s, ty] * xu[j2[s, ty]])
        for b, (s, ty) in enumerate(slots):
            S[b] = a1[s, ty] * M[j1[s, ty]] + a2[s, ty] * M[j2[s, ty]]
        d0 = np.diag(S).copy()
        dead = np.zeros(k, bool)
        for a in range(k):                       # elimination without pivoting; a dependent row is skipped
            if not S[a, a] > 1e-11 * d0[a]:
                dead[a] = True
                continue
            inv = 1.0 / S[a, a]
            for b in range(a + 1, k):
                f = S[b, a] * inv
                S[b, a + 1:] -= f * S[a, a + 1:]
                r[b] -= f * r[a]
        la = np.zeros(k)
        for a in range(k - 1, -1, -1):
            if dead[a]:
                continue
            la[a] = (r[a] - S[a, a + 1:] @ la[a + 1:]) / S[a, a]
        for a, (s, ty) in enumerate(slots):
            if dead[a]:
                act[s, ty] = False
            else:
                lam[s, ty] = la[a]
        x = xu + M @ la
    return x, lam, max_it, False, kmax


def solve_robot(b, i, **kw):
    """Instance i of a batch dict through the fp64 Kronecker model + PDAS -> (q_soln[12h], solves, ok, kmax)."""
    from . import kron_model as K
    h = b["horizon"]
    H, g = K.assemble(b, i)
    stance = [k for k in range(4 * h) if b["gait"][i][k]]
    vi = np.array([3 * k + a for k in stance for a in range(3)], int)
    out = np.zeros(12 * h)
    if vi.size == 0:
        return out, 0, True, 0
    Hinv = K.sweep_inverse(H[np.ix_(vi, vi)])
    mi = np.float64(np.float32(1.0) / np.float32(b["mu"]))
    fm = np.array([np.float64(np.float32(b["f_max"]) * np.float32(b["gait"][i][k])) for k in stance])
    x, lam, it, ok, kmax = solve(Hinv, g[vi], len(stance), mi, fm, **kw)
    out[vi] = x
    return out, it, ok, kmax
