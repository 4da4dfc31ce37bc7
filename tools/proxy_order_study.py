#!/usr/bin/env python
"""Host emulation (batch permuted, size order and hint off): which cost proxy, computable from a robot's record before anything is
solved, orders a multi-round launch best?  werr = sum_k Q_k |e_k + T de_k| -- the weighted tracking error the coasting state would
have at the end of the horizon (T = h dt)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_study import permute, interleave
from tools.size_order_ab import run


def feats(b):
    B = int(b['batch']); h = int(b['horizon'])
    q = b['q']; w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    roll = 2 * (w * x + y * z); pitch = 2 * (w * y - z * x)   # small-angle forms: a proxy
    x0 = np.concatenate([np.stack([roll, pitch, b['yaw']], 1), b['p'], b['w'], b['v']], 1)
    tr = b['traj'].reshape(B, h, 12)
    Q = b['weights']; T = h * float(b['dt'])
    ep = np.abs((x0[:, 3:6] - tr[:, 0, 3:6]) + T * (x0[:, 9:12] - tr[:, 0, 9:12]))
    er = np.abs((x0[:, 0:3] - tr[:, 0, 0:3]) + T * (x0[:, 6:9] - tr[:, 0, 6:9]))
    werr = (Q[:, 3:6] * ep).sum(1) + (Q[:, 0:3] * er).sum(1)
    g = b['gait'].reshape(B, h, 4).astype(int)
    return werr, g.sum((1, 2)), g[:, :3].sum((1, 2))


def main():
    out = []
    for name, b, steps, maxfit in (("cfg4", W.make_config(4, batch=8192), 20, 21), ("cfg2", W.make_config(2), 40, 21), ("cfg2_8192", W.make_config(2, batch=8192), 20, 21),
                                   ("cfg3", W.make_config(3, batch=4096), 10, 32)):
        B = int(b["batch"])
        werr, nst, first3 = feats(b)
        fit = nst <= maxfit
        scores = {"nst": nst.astype(float), "werr_x_nst": werr * nst, "werr_x_first3": werr * first3, "werr": werr,
                  "nst_plus": nst * (1.0 + werr / werr.mean())}
        r = {"workload": name, "plain": B / run(b, 0, steps)[0] * 1e3}
        for k, sc in scores.items():
            fo = np.argsort(np.where(fit, -sc, np.inf), kind="stable")[:fit.sum()]
            perm = interleave(fo, np.nonzero(~fit)[0]) if (~fit).any() else fo
            r[k] = B / run(permute(b, perm), 0, steps)[0] * 1e3
            if (~fit).any():
                po = np.argsort(np.where(~fit, -sc, np.inf), kind="stable")[:(~fit).sum()]
                r[k + "+passes_too"] = B / run(permute(b, interleave(fo, po)), 0, steps)[0] * 1e3
                r[k + "_all"] = B / run(permute(b, np.argsort(-sc, kind="stable")), 0, steps)[0] * 1e3
        out.append(r)
        print(name, {k: (round(v / 1e7, 3) if k != "workload" else v) for k, v in r.items()}, file=sys.stderr)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
