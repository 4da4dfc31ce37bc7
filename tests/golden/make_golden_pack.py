"""Generates tests/golden/pack_*.npz: synthetic controller commands
(workloads.make_commands) and the record the oracle's restatement of
ConvexMPCLocomotion::updateMPCIfNeeded / ::solveDenseMPC packing
(oracle_pack_command, ConvexMPCLocomotion.cpp:498-640, Gait.cpp:142-166) and
of the force rotation (:672-680) produce for them.  Fixtures are data only.

    python tests/golden/make_golden_pack.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from quadruped_ctrl_amd import workloads as W  # noqa: E402
from oracle import oracle as O  # noqa: E402

CASES = {"pack_h10": dict(batch=96, horizon=10, seed=11, omni_mode=0),
         "pack_h16_omni": dict(batch=48, horizon=16, seed=12, omni_mode=1)}
REC = ("p", "v", "q", "w", "r", "yaw", "traj", "gait", "weights", "alpha", "x_drag")


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    for name, kw in CASES.items():
        cmd = W.make_commands(**kw)
        dt = np.float32(0.026)
        rec, wpd, xci = O.pack_commands(cmd, dt)
        rng = np.random.default_rng(5)
        grf = rng.normal(0, 40, (cmd["batch"], 12)).astype(np.float32)
        f_ff = O.forces_to_body(cmd["r_body"], grf)
        np.savez_compressed(os.path.join(out, name + ".npz"), dt=dt, grf=grf, f_ff=f_ff, wpd_out=wpd, xci_out=xci,
                            **{"cmd_" + k: v for k, v in cmd.items()}, **{"rec_" + k: rec[k] for k in REC})
        print(name, cmd["batch"], "standing", int((cmd["gait_type"] == 4).sum()))


if __name__ == "__main__":
    main()
