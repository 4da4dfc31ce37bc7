"""Wider parity sweep of the large-problem path (192 < n_r <= 432; run on the GPU box): more horizons, gaits, seeds and
robots than tests/test_gpu_parity.py::test_large_problems_beyond_192_rows.  Every robot's solution is compared with the
reference's qpOASES (iteration cap lifted) on the fp64 Kronecker model's reduced QP: solution, objective, feasibility.

    python tools/stress_large.py [robots_per_case]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401
from oracle import kron_model as K  # noqa: E402
from oracle import oracle as O  # noqa: E402
from quadruped_ctrl_amd import workloads as W  # noqa: E402
from quadruped_ctrl_amd.binding import BatchedConvexMPC  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 24


def mixed(B, h, seed):
    """random contact tables (stairs-like) at a long horizon: n_r anywhere between 3 h and 12 h, every route of the call"""
    rng = np.random.default_rng(seed)
    b = W.make_long_horizon(B, h, "stand", seed=seed)
    g = b["gait"].reshape(B, h, 4)
    for i in range(B):
        p = rng.uniform(0.0, 0.6)
        g[i] = (rng.uniform(size=(h, 4)) >= p).astype(g.dtype)
        g[i, 0, rng.integers(4)] = 1
    b["x_drag"][:] = rng.normal(0, 0.4, B).astype(np.float32)
    return b


CASES = [
    ("trot h=34", lambda: W.make_long_horizon(N, 34, "trot", seed=5)),
    ("trot h=36", lambda: W.make_long_horizon(N, 36, "trot", seed=6)),
    ("stand h=18", lambda: W.make_long_horizon(N, 18, "stand", seed=7)),
    ("stand h=27", lambda: W.make_long_horizon(N, 27, "stand", seed=8)),
    ("stand h=36", lambda: W.make_long_horizon(max(N // 2, 4), 36, "stand", seed=9)),
    ("braking h=20", lambda: W.make_standing(N, 20, seed=10)),
    ("braking h=30", lambda: W.make_standing(max(N // 2, 4), 30, seed=11)),
    ("random tables h=24", lambda: mixed(N, 24, 12)),
    ("random tables h=36", lambda: mixed(N, 36, 13)),
]

worst = [0.0, 0.0, 0.0]
for name, mk in CASES:
    b = mk()
    B, h = b["batch"], b["horizon"]
    m = BatchedConvexMPC(0, max_batch=B, max_horizon=36)
    m.setup(b["dt"], h, b["mu"], b["f_max"])
    res = m.solve(b, full=True)
    st = res["status"]
    nst = (b["gait"].reshape(B, -1) != 0).sum(1)
    wx = wf = wi = 0.0
    nbig = nfail = 0
    t0 = time.time()
    for i in range(B):
        if st[i] & 47:
            continue
        H, g = K.assemble(b, i)
        Hf, gf, A, lb, ub, x0 = O.assemble(b, i)
        ve, Hr, gr, Ar, lr, ur = O.reduce(Hf, gf, A, lb, ub)
        vi = np.nonzero(~ve)[0]
        if vi.size == 0:
            continue
        Hm, gm = H[np.ix_(vi, vi)], g[vi]
        xq, y, used, rc, irc = O.qpoases(Hm, gm, Ar, lr, ur, nwsr=100000)
        if rc != 0 or irc != 0:
            nfail += 1
            continue
        nbig += vi.size > 192
        xs = res["soln"][i][~ve]
        f = lambda x: 0.5 * x @ Hm @ x + gm @ x  # noqa: E731
        ax = Ar @ xs
        wi = max(wi, np.maximum(lr - ax, 0).max(), np.maximum(ax - ur, 0).max())
        wf = max(wf, abs(f(xs) - f(xq)) / max(abs(f(xq)), 1e-30))
        wx = max(wx, np.abs(xs - xq).max() / max(np.abs(xq).max(), 1.0))
    print(f"{name:20s} B={B:3d} n_r {3 * nst.min():3d}..{3 * nst.max():3d} ({nbig} beyond 192 rows) iters mean {res['iters'].mean():6.1f} "
          f"max {res['iters'].max():3d} | x {wx:.2e} objective {wf:.2e} infeasibility {wi:.2e} | status!=0 {int(((st & 47) != 0).sum())} "
          f"| qpOASES failures {nfail} | {time.time() - t0:.0f} s of CPU", flush=True)
    worst = [max(worst[0], wx), max(worst[1], wf), max(worst[2], wi)]
    m.close()
print("WORST x %.2e objective %.2e infeasibility %.2e" % tuple(worst))
