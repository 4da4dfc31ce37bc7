"""Do two independent batches on two HIP streams overlap (tail of one launch with the head of the next)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_ctrl_amd import workloads
from quadruped_ctrl_amd.binding import BatchedConvexMPC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = 2000
b = workloads.make_config(1, batch=B)
ctx = []
for s in range(NS):
    m = BatchedConvexMPC(0, max_batch=B); m.setup(b["dt"], 10, b["mu"], b["f_max"]); m.set_max_stance(20)
    d = m.upload(b); o = m.alloc_outputs(B, full=False, iters=True); inp, out = m.make_args(d, o)
    st = torch.cuda.Stream()
    ctx.append((m, inp, out, st, d, o))
for _ in range(50):
    for (m, inp, out, st, d, o) in ctx: m.solve_async(B, inp, out, st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    m, inp, out, st, d, o = ctx[k % NS]
    m.solve_async(B, inp, out, st)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"B={B} streams={NS}: {B*steps/dt/1e6:.2f} M QP/s, {dt/steps*1e6:.1f} us/step")
