#!/usr/bin/env python
"""Held-out check of the size order / proxy staging: closed-loop rollouts (workloads.Rollout), NO order hint on either handle;
qmpc_set_size_order off against on, every cycle's record solved by both, results compared bit for bit."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from quadruped_ctrl_amd import workloads as W
from tools.order_hint import handle

out = []
for gait, h, B, kick in (("mixed", 10, 4096, 1.0), ("mixed", 10, 8192, 1.0), ("trot", 10, 4096, 1.0), ("trot", 16, 4096, 1.0), ("mixed", 10, 1024, 1.0), ("trot", 10, 1024, 1.0),
                         ("mixed", 10, 4096, 2.0)):
    ro = W.Rollout(B, h, gait, seed=3, kick=kick)
    b = ro.record()
    hs = {}
    for so in (0, 1):
        hs[so] = handle(b, B, False, stance=False)
        hs[so].set_size_order(so)
    o = {so: hs[so].alloc_outputs(B, full=True, iters=True) for so in (0, 1)}
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    t = {0: [], 1: []}
    same = True
    for c in range(24):
        b = ro.record()
        d = hs[0].upload(b)
        a0 = hs[0].make_args(d, o[0]); a1 = hs[1].make_args(d, o[1])
        hs[0].solve_async(B, *a0); torch.cuda.synchronize()
        order = (0, 1) if c % 2 == 0 else (1, 0)
        for j, so in enumerate(order):
            e[2 * j].record(); hs[so].solve_async(B, *(a0 if so == 0 else a1)); e[2 * j + 1].record()
        torch.cuda.synchronize()
        for j, so in enumerate(order):
            if c >= 2: t[so].append(e[2 * j].elapsed_time(e[2 * j + 1]))
        same = same and bool((o[0]["soln"] == o[1]["soln"]).all()) and bool((o[0]["iters"] == o[1]["iters"]).all())
        ro.advance(o[0]["grf"].cpu().numpy())
    for so in (0, 1): hs[so].close()
    r = {"rollout": gait, "horizon": h, "batch": B, "pushes": kick, "ms_off": float(np.mean(t[0])), "ms_on": float(np.mean(t[1])),
         "gain": float(np.mean(t[0]) / np.mean(t[1]) - 1.0), "bit_identical": same}
    out.append(r)
    print(r, file=sys.stderr)
print(json.dumps(out, indent=1))
