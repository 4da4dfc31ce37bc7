import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
B, h = 512, 10
for label, kw, dem in (("pushes", dict(kick=1.0), None), ("accel", dict(kick=0.0), (1.6, 0.4, 1.0))):
    ro = W.Rollout(B, h, "trot", seed=0, **kw)
    if dem: ro.demand(*dem)
    b = ro.record()
    m = BatchedConvexMPC(0, max_batch=B, max_horizon=16); m.setup(b["dt"], h, b["mu"], b["f_max"])
    c = BatchedConvexMPC(0, max_batch=B, max_horizon=16); c.setup(b["dt"], h, b["mu"], b["f_max"])
    ws = m.warm_start(B, 1)
    prev = None
    tot = dict(cand=0, hit=0, final=0, itw=0, itc=0)
    for cyc in range(20):
        b = ro.record()
        rw = m.solve(b, full=True); rc = c.solve(b, full=True)
        cur = [set(int(x) for x in row if x >= 0) for row in ws.cpu().numpy()]
        if prev is not None and cyc >= 3:
            for i in range(B):
                cand = {e - 20 for e in prev[i] if e - 20 >= 0 and b["gait"][i, (e - 20) // 5]}
                tot["cand"] += len(cand); tot["hit"] += len(cand & cur[i]); tot["final"] += len(cur[i])
            tot["itw"] += rw["iters"].sum(); tot["itc"] += rc["iters"].sum()
        prev = cur
        ro.advance(rc["grf"])
    print(label, {k: int(v) for k, v in tot.items()}, "precision %.2f recall %.2f" % (tot["hit"] / max(tot["cand"], 1), tot["hit"] / max(tot["final"], 1)))
