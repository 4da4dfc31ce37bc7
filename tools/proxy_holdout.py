#!/usr/bin/env python
"""The cost proxy (qmpc_robot_keys) on workloads that had NO part in fitting its constants: closed-loop rollouts (the state after
N cycles of the robots' own dynamics with random pushes; mixed gaits / trot / pacing / bounding, horizons 10 and 16), the record's
correlation with the iteration count the GPU reports."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from quadruped_ctrl_amd import workloads as W
from tools.order_hint import handle
from tools.proxy2_study import score2

out = []
for gait, h, B, kick in (("mixed", 10, 4096, 1.0), ("trot", 10, 4096, 1.0), ("pace", 10, 2048, 1.0), ("bound", 10, 2048, 1.0), ("mixed", 16, 2048, 1.0), ("mixed", 10, 4096, 2.0)):
    ro = W.Rollout(B, h, gait, seed=11, kick=kick)
    b = ro.record()
    m = handle(b, B, False, stance=False)
    o = m.alloc_outputs(B, full=False, iters=True)
    cs, cn = [], []
    for c in range(14):
        b = ro.record()
        d = m.upload(b)
        inp, outp = m.make_args(d, o)
        m.solve_async(B, inp, outp)
        torch.cuda.synchronize()
        it = o["iters"].cpu().numpy()
        if c >= 6 and it.std() > 0:
            sc, nst = score2(b)
            cs.append(float(np.corrcoef(sc, it)[0, 1]))
            cn.append(float(np.corrcoef(nst, it)[0, 1]) if nst.std() > 0 else 0.0)
        ro.advance(o["grf"].cpu().numpy())
    m.close()
    r = {"rollout": gait, "horizon": h, "batch": B, "pushes": kick, "corr_score_iters": float(np.mean(cs)), "corr_size_iters": float(np.mean(cn)),
         "iters_mean": float(it.mean()), "iters_max": int(it.max())}
    out.append(r)
    print(r, file=sys.stderr)
print(json.dumps(out, indent=1))
