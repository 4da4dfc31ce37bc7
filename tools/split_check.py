"""Decoupled path (sweep kernel -> work items -> engine kernel) against the one-kernel path on the same robots:
results, iteration counts, hand-backs, time per call.  GPU box:  python tools/split_check.py [B]"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from quadruped_ctrl_amd import workloads as W  # noqa: E402
from quadruped_ctrl_amd.binding import BatchedConvexMPC  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
out = {}
for name, b in (("standing_h10", W.make_standing(B, 10)), ("standing_h14", W.make_standing(B, 14)),
                ("standing_h16", W.make_standing(B, 16)), ("standing_h10_calm", W.make_standing(B, 10, calm=True)),
                ("config4", W.shard(W.make_config(4, batch=8 * B), 0, 8))):
    res = {}
    for split in (False, True):
        m = BatchedConvexMPC(0, max_batch=b["batch"], max_horizon=16)
        m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
        nst = (b["gait"] != 0).sum(1)
        m.set_max_stance(int(nst.max()))
        m.set_min_stance(int(nst.min()))
        m.set_split(split)
        r = m.solve(b, full=True)
        d = m.upload(b)
        o = m.alloc_outputs(b["batch"], full=False, iters=True)
        inp, outp = m.make_args(d, o)
        s = torch.cuda.current_stream(0)
        for _ in range(10):
            m.solve_async(b["batch"], inp, outp, s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            m.solve_async(b["batch"], inp, outp, s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 50
        res[split] = (r, dt)
        m.close()
    a, t_a = res[False]
    c, t_c = res[True]
    den = np.maximum(np.abs(a["soln"]).max(1), 1.0)
    err = np.abs(a["soln"] - c["soln"]).max(1) / den
    out[name] = {"batch": b["batch"], "ms_one_kernel": t_a * 1e3, "ms_split": t_c * 1e3, "qps_one_kernel": b["batch"] / t_a,
                 "qps_split": b["batch"] / t_c, "max_rel_diff": float(err.max()), "worst_robot": int(err.argmax()),
                 "status_one": np.unique(a["status"]).tolist(), "status_split": np.unique(c["status"]).tolist(),
                 "iters_one_mean_max": [float(a["iters"].mean()), int(a["iters"].max())],
                 "iters_split_mean_max": [float(c["iters"].mean()), int(c["iters"].max())],
                 "iters_equal": bool(np.array_equal(a["iters"], c["iters"]))}
    print(name, json.dumps(out[name]), flush=True)
print(json.dumps(out))
