"""Profiling aid: cycles per phase of the decoupled engine's primal-dual active-set start (summed over a robot's solves)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
h = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
b = W.make_standing(B, h)
m = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
m.setup(b["dt"], h, b["mu"], b["f_max"])
m.set_pdas(16)
d = m.upload(b); o = m.alloc_outputs(B); inp, out = m.make_args(d, o)
for _ in range(3): m.solve_async(B, inp, out)
torch.cuda.synchronize()
clk = m.debug_clock(B)
m.solve_async(B, inp, out); torch.cuda.synchronize()
c = clk.cpu().numpy().astype(np.float64)
it = o["iters"].cpu().numpy(); st = o["status"].cpu().numpy()
ok = (st & 256) != 0
ph = c[ok][:, :4]
print(f"standing h{h} B={B}: {ok.sum()} robots answered, solves mean {it[ok].mean():.2f}")
names = ["(1) slacks / set / slots (+barrier)", "(2) build S (gathers)", "(3) LDL^T + substitutions", "(4) x update"]
for k, nm in enumerate(names):
    print(f"   {nm:38s} per robot median {np.median(ph[:, k]):9.0f}  per solve {np.median(ph[:, k] / np.maximum(it[ok], 1)):8.0f}")
print(f"   whole start {np.median(c[ok][:, 13] - c[ok][:, 12]):.0f} cycles median")
