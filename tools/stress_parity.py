"""Wide randomized solver-parity sweep (run on the GPU box):
for many workload families and seeds, dump the GPU's own reduced QP (H, g),
solve it with the reference's qpOASES (iteration cap lifted) and compare the
full solutions.  Prints the worst relative error, iteration statistics and how
many robots took the fallback engine.

    python tools/stress_parity.py [robots_per_family]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from oracle import oracle as O  # noqa: E402
from quadruped_ctrl_amd import workloads as W  # noqa: E402
from quadruped_ctrl_amd.binding import BatchedConvexMPC  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400


def hard_commands(B, h, seed):
    cmd = W.make_commands(B, horizon=h, seed=seed, stand_fraction=0.1)
    rec, _, _ = O.pack_commands(cmd, np.float32(0.026))
    rec.update(dt=0.026, mu=0.4, f_max=120.0)
    return rec


def hard_stand(B, h, seed):
    cmd = W.make_commands(B, horizon=h, seed=seed, stand_fraction=1.0)
    rec, _, _ = O.pack_commands(cmd, np.float32(0.026))
    rec.update(dt=0.026, mu=0.4, f_max=40.0)
    return rec


def low_fmax(B, seed):
    b = W.make_config(2, batch=B)
    rng = np.random.default_rng(seed)
    b["f_max"] = 40.0
    b["traj"].reshape(B, 10, 12)[:, :, 10] = rng.uniform(-2, 2, (B, 1))
    return b


FAMILIES = {
    "cfg2_mixed": lambda: W.make_config(2, batch=N),
    "cfg4_random_stairs": lambda: W.make_config(4, batch=N),
    "cfg3_h16": lambda: W.make_config(3, batch=max(N // 4, 8)),
    "standing_h10": lambda: W.make_standing(max(N // 8, 8), 10),
    "hard_commands_h10": lambda: hard_commands(N, 10, 5),
    "hard_commands_h14": lambda: hard_commands(max(N // 4, 8), 14, 6),
    "hard_commands_h16_stand": lambda: dict(hard_stand(max(N // 4, 8), 16, 11)),
    "standing_h14": lambda: W.make_standing(max(N // 8, 8), 14),
    "low_fmax": lambda: low_fmax(N, 9),
}

worst_all = 0.0
for name, mk in FAMILIES.items():
    b = mk()
    B, h = b["batch"], b["horizon"]
    m = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
    m.setup(b["dt"], h, b["mu"], b["f_max"])
    if os.environ.get("QMPC_STRESS_SPLIT"):   # the decoupled path for every robot of the 128- / 192-row classes
        m.set_split(2)
    Hd, gd, ld = m.debug_dump(B)
    res = m.solve(b, full=True)
    m.debug_off()
    Hd, gd = Hd.cpu().numpy(), gd.cpu().numpy()
    worst, nfail, over = 0.0, 0, 0
    for i in range(B):
        H, g, A, lb, ub, x0 = O.assemble(b, i)
        ve, Hr, gr, Ar, lr, ur = O.reduce(H, g, A, lb, ub)
        n = gr.size
        if n == 0:
            continue
        xq, y, used, rc, irc = O.qpoases(Hd[i][:n, :n], gd[i][:n], Ar, lr, ur, nwsr=20000)
        if rc != 0 or irc != 0:
            nfail += 1
            continue
        over += used > 100
        xs = res["soln"][i][~ve]
        worst = max(worst, np.abs(xs - xq).max() / max(np.abs(xq).max(), 1.0))
    st = res["status"]
    print(f"{name:22s} B={B:5d} h={h:2d} worst rel err {worst:.2e} | iters mean {res['iters'].mean():.2f} "
          f"max {res['iters'].max()} | fallback {int(((st & 16) != 0).sum())} | error status {int(((st & 47) != 0).sum())} "
          f"| qpOASES failures {nfail}, over its nWSR=100 cap {over}")
    worst_all = max(worst_all, worst)
    m.close()
print("WORST", worst_all)
