"""The reference pipeline's own fp32 noise floor -- TEST INFRASTRUCTURE ONLY.

The reference assembles the condensed QP in float (fpt = float,
SolverMPC.cpp:395-399) and leaves the ORDER of the float operations to Eigen
(un-vendored, unpinned): its answer is defined only up to the spread between
equally legitimate evaluation orders of that one expression.  This module feeds
one robot through the oracle assembly in six such orders
(oracle_set_accum_mode, mpc_oracle.c) and through the fp64 model
(kron_model.py), solves every variant with the reference's own qpOASES, and
reports

    spread   = max over pairs of float orders   |f_a - f_b|_inf / max(|f_b|_inf, 1 N)
    to_fp64  = max over float orders of the same distance to the fp64-assembled answer

on the twelve first-step forces (what the parity tests bound) and on the whole
12h solution.  The GPU assembles in fp64, so ITS distance to the default float
order is bounded by to_fp64 (plus the 1e-10 assembly parity pinned by
test_assembly_vs_fp64_model).  Used by tests/golden/make_noise_floor.py (the
committed fixture) and by the horizon > 10 GPU tests (per-robot bound).
"""
import numpy as np

from . import kron_model as K
from . import oracle as O

MODES = (0, 1, 2, 3, 4, 5)


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1.0)


def variants(b, i):
    """-> dict: mode -> q_soln[12h] for the six float orders, 'fp64' -> fp64-assembled."""
    out = {}
    n = 12 * b["horizon"]
    H0, g0, A, lb, ub, _ = O.assemble(b, i)
    ve, _, _, Ar, lr, ur = O.reduce(H0, g0, A, lb, ub)

    def solve(H, g):
        _, Hr, gr, _, _, _ = O.reduce(H, g, A, lb, ub)
        q = np.zeros(n)
        if gr.size:
            # (no cap of 100 working-set recalculations here: the floor is a property of the assembly, and long
            #  horizons with many rows at a bound need more of them)
            x, _, _, rc, irc = O.qpoases(Hr, gr, Ar, lr, ur, nwsr=5000)
            assert rc == 0 and irc == 0
            q[~ve] = x
        return q

    for m in MODES:
        O.lib().oracle_set_accum_mode(m)
        try:
            H, g, _, _, _, _ = O.assemble(b, i)
        finally:
            O.lib().oracle_set_accum_mode(0)
        out[m] = solve(H, g)
    out["fp64"] = solve(*K.assemble(b, i))
    return out


def robot_floor(b, i):
    """-> dict(spread12, spread_full, fp64_12, fp64_full) for robot i."""
    v = variants(b, i)
    qs = [v[m] for m in MODES]
    return {
        "spread12": max(rel(a[:12], c[:12]) for a in qs for c in qs),
        "spread_full": max(rel(a, c) for a in qs for c in qs),
        "fp64_12": max(rel(v["fp64"][:12], a[:12]) for a in qs),
        "fp64_full": max(rel(v["fp64"], a) for a in qs),
    }
