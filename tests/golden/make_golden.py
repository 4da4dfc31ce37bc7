"""Generates tests/golden/*.npz IN THE BUILD CONTAINER (needs oracle/_ref, i.e.
the reference's qpOASES compiled from /root/reference).  The fixtures are data:
synthetic inputs (SURVEY.md 8d generators) and the outputs of the reference
pipeline -- oracle assembly (C restatement of SolverMPC.cpp, float) followed by
the reference's own qpOASES 3.2.0 driven as SolverMPC.cpp:527-541.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadruped_ctrl_amd import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {
    "cfg1_trot_h10": lambda: W.make_config(1, batch=64),
    "cfg2_mixed_h10": lambda: W.make_config(2, batch=64),
    "cfg3_trot_h16": lambda: W.make_config(3, batch=32),
    "cfg4_random_stairs_h10": lambda: W.make_config(4, batch=96),
    "standing_h10": lambda: W.make_standing(16, 10),
    "standing_h14": lambda: W.make_standing(8, 14),
}
KEYS = ("p", "v", "q", "w", "r", "yaw", "weights", "traj", "alpha", "x_drag", "gait")


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    for name, mk in CASES.items():
        b = mk()
        q, nwsr, rc = O.solve_batch(b)
        assert (rc == 0).all(), name
        # reduced QP of the first 4 instances as handed to qpOASES (float-assembled)
        Hs, gs = [], []
        for i in range(4):
            H, g, A, lb, ub, x0 = O.assemble(b, i)
            Hs.append(H.astype(np.float32))
            gs.append(g.astype(np.float32))
        np.savez_compressed(
            os.path.join(out, name + ".npz"),
            batch=b["batch"], horizon=b["horizon"], dt=b["dt"], mu=b["mu"], f_max=b["f_max"],
            q_soln=q, nwsr=nwsr, H4=np.stack(Hs), g4=np.stack(gs),
            **{k: b[k] for k in KEYS})
        print(name, b["batch"], "nwsr mean", nwsr.mean(), "max", nwsr.max())


if __name__ == "__main__":
    main()
