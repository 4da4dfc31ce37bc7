#!/usr/bin/env python
"""CPU study (oracle only, no GPU): how well does the PREVIOUS MPC cycle's working set, slid by one horizon step, predict
this cycle's -- for the robots that matter to a one-round launch, the ones with many active-set iterations?

For closed-loop rollouts (workloads.ConfigRollout seeded from a BASELINE config, or workloads.Rollout) every cycle's QP is
solved by the oracle pipeline (float assembly restatement + the reference's qpOASES); the working set is read off the duals.
Reported per bucket of |W*| (= iterations of a cold dual active set): candidates, hits, misses, wrong guesses.

    python tools/ws_tail_study.py [cfg] [batch] [cycles]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from quadruped_ctrl_amd import workloads as W  # noqa: E402


def working_set(b, i):
    """-> (set of global constraint ids 5 (4 step + foot) + type, nWSR, first-step forces)."""
    H, g, A, lb, ub, _ = O.assemble(b, i)
    ve, Hr, gr, Ar, lr, ur = O.reduce(H, g, A, lb, ub)
    h = b["horizon"]
    if gr.size == 0:
        return set(), 0, np.zeros(12)
    x, y, used, rc, irc = O.qpoases(Hr, gr, Ar, lr, ur)
    yc = y[gr.size:]
    st = np.flatnonzero(b["gait"][i])           # stance foot-steps, reduced constraint block c <-> foot-step st[c]
    ws = set()
    # the reference's rows per foot-step (SolverMPC.cpp:366-370): [mu_inv 0 1], [-mu_inv 0 1], [0 mu_inv 1], [0 -mu_inv 1], [0 0 1]
    # lower bound 0 active on rows 0..3 = friction pyramid faces; upper bound f_max active on row 4
    for c in range(st.size):
        for t in range(5):
            yy = yc[5 * c + t]
            if t < 4 and yy > 1e-9:
                ws.add(5 * int(st[c]) + t)
            if t == 4 and yy < -1e-9:
                ws.add(5 * int(st[c]) + 4)
    q = np.zeros(12 * h)
    q[~ve] = x
    return ws, used, q[:12]


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    cycles = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    b0 = W.make_config(cfg, batch=B)
    ro = W.ConfigRollout(b0, periodic=(cfg != 4))
    prev = None
    rows = []
    for c in range(cycles):
        b = ro.record()
        cur, its, grf = [], np.zeros(B, int), np.zeros((B, 12))
        for i in range(B):
            ws, used, f = working_set(b, i)
            cur.append(ws)
            its[i] = used
            grf[i] = f
        if prev is not None:
            for i in range(B):
                cand = {e - 20 for e in prev[i] if e - 20 >= 0 and b["gait"][i, (e - 20) // 5]}
                rows.append((c, len(cur[i]), len(cand), len(cand & cur[i]), len(prev[i]), its[i]))
        prev = cur
        print(f"cycle {c}: |W| mean {np.mean([len(w) for w in cur]):.2f} max {max(len(w) for w in cur)}  nWSR mean {its.mean():.2f} max {its.max()}", flush=True)
        ro.advance(grf)
    r = np.array(rows)
    r = r[r[:, 0] >= 2]
    print("bucket by |W*| now:   robots   |W*|   candidates   hits   wrong   missing   (prev |W|)")
    for lo, hi in ((0, 0), (1, 2), (3, 4), (5, 7), (8, 11), (12, 99)):
        m = (r[:, 1] >= lo) & (r[:, 1] <= hi)
        if m.sum():
            x = r[m]
            print(f"  {lo:2d}-{hi:2d}: {m.sum():6d}  {x[:, 1].mean():5.2f}  {x[:, 2].mean():6.2f}  {x[:, 3].mean():6.2f}  {(x[:, 2] - x[:, 3]).mean():6.2f}  "
                  f"{(x[:, 1] - x[:, 3]).mean():6.2f}   ({x[:, 4].mean():.2f})")
    # the launch's tail: the robot with the largest |W*| of each cycle
    print("per cycle, the hardest robot: |W*|, candidates, hits")
    for c in sorted(set(r[:, 0])):
        x = r[r[:, 0] == c]
        k = x[:, 1].argmax()
        print(f"  cycle {c}: |W*| {x[k, 1]}  cand {x[k, 2]}  hits {x[k, 3]}  prev|W| {x[k, 4]} nWSR {x[k, 5]};  max over robots with prev|W| < 5: {x[x[:, 4] < 5][:, 1].max()}")


if __name__ == "__main__":
    main()
