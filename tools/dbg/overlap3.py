"""Scratch: configs[4] split by size class on the host (<= 64 | 65..80 | 81..96 rows), each group its own handle:
one stream one after the other vs three streams at once (fork / join), every launch order."""
import os, sys, time, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_ctrl_amd import workloads
from quadruped_ctrl_amd.binding import BatchedConvexMPC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = 100
b = workloads.make_config(4, batch=B)
nst = (b["gait"].reshape(B, -1) != 0).sum(1)
def subset(d, mask):
    out = {k: (v[mask] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in d.items()}
    out["batch"] = int(mask.sum()); return out
def ctx_for(d, lo, hi):
    n = d["batch"]; m = BatchedConvexMPC(0, max_batch=n)
    m.set_max_stance(hi); m.set_min_stance(lo)
    m.setup(d["dt"], d["horizon"], d["mu"], d["f_max"])
    dv = m.upload(d); o = m.alloc_outputs(n, full=False, iters=True); inp, out = m.make_args(dv, o)
    return m, n, inp, out, dv, o
def timed(fn):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e6
groups = {"<=64": (nst <= 21, 1, 21), "65..80": ((nst >= 22) & (nst <= 26), 22, 26), "81..96": (nst >= 27, 27, 32)}
ctx = {k: ctx_for(subset(b, m), lo, hi) for k, (m, lo, hi) in groups.items()}
s0 = torch.cuda.Stream()
ts = {k: timed(lambda c=c: c[0].solve_async(c[1], c[2], c[3], s0)) for k, c in ctx.items()}
print({k: (ctx[k][1], round(t, 1)) for k, t in ts.items()}, "sum", round(sum(ts.values()), 1))
ss = [torch.cuda.Stream() for _ in range(3)]
def conc(order):
    ev = torch.cuda.Event(); ev.record(s0)
    for s in ss: s.wait_event(ev)
    for s, k in zip(ss, order):
        c = ctx[k]; c[0].solve_async(c[1], c[2], c[3], s)
    for s in ss:
        e = torch.cuda.Event(); e.record(s); s0.wait_event(e)
for order in itertools.permutations(groups):
    print(order, round(timed(lambda: conc(order)), 1), "us")
whole = ctx_for(b, 1, 32)
print("shipped single call", round(timed(lambda: whole[0].solve_async(whole[1], whole[2], whole[3], s0)), 1), "us")
