"""What would a classifying pre-pass buy configs[4]?  Emulation from the host: the batch is split by size class
beforehand (what a pre-pass kernel would do), the two classes are solved (a) one after the other on one stream, (b) on
two streams at once, in either launch order; compared with the shipped single call on the mixed batch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_ctrl_amd import workloads
from quadruped_ctrl_amd.binding import BatchedConvexMPC

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
steps = 300
b = workloads.make_config(4, batch=B)
nst = (b["gait"].reshape(B, -1) != 0).sum(1)
small = nst * 3 <= 64


def subset(d, mask):
    out = {k: (v[mask] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in d.items()}
    out["batch"] = int(mask.sum())
    return out


def ctx_for(d, hint_lo=None, hint_hi=None):
    n = d["batch"]
    m = BatchedConvexMPC(0, max_batch=n)
    m.setup(d["dt"], d["horizon"], d["mu"], d["f_max"])
    if hint_hi:
        m.set_max_stance(hint_hi)
    if hint_lo:
        m.set_min_stance(hint_lo)
    dv = m.upload(d)
    o = m.alloc_outputs(n, full=False, iters=True)
    inp, out = m.make_args(dv, o)
    return m, n, inp, out, dv, o


def timed(fn):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e6


whole = ctx_for(b)
s0 = torch.cuda.Stream()
t_whole = timed(lambda: whole[0].solve_async(whole[1], whole[2], whole[3], s0))
c1 = ctx_for(subset(b, small), hint_hi=21)
c4 = ctx_for(subset(b, ~small), hint_lo=22, hint_hi=32)
t_1 = timed(lambda: c1[0].solve_async(c1[1], c1[2], c1[3], s0))
t_4 = timed(lambda: c4[0].solve_async(c4[1], c4[2], c4[3], s0))
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
ev_f, ev_j = torch.cuda.Event(), torch.cuda.Event()


def both(first, second):
    # one "call": fork from s0, the two classes on two streams, join back on s0
    ev_f.record(s0)
    sa.wait_event(ev_f)
    sb.wait_event(ev_f)
    first[0].solve_async(first[1], first[2], first[3], sa)
    second[0].solve_async(second[1], second[2], second[3], sb)
    ev_j.record(sa)
    s0.wait_event(ev_j)
    ev_j2 = torch.cuda.Event()
    ev_j2.record(sb)
    s0.wait_event(ev_j2)


t_14 = timed(lambda: both(c1, c4))
t_41 = timed(lambda: both(c4, c1))
print(f"configs[4] B={B}: {int(small.sum())} robots <= 64 rows, {int((~small).sum())} above")
print(f"shipped single call, mixed batch          {t_whole:7.1f} us  {B / t_whole:6.2f} M QP/s")
print(f"pre-classified, one stream: {t_1:6.1f} + {t_4:6.1f} = {t_1 + t_4:7.1f} us  {B / (t_1 + t_4):6.2f} M QP/s")
print(f"pre-classified, two streams, 64-row first {t_14:7.1f} us  {B / t_14:6.2f} M QP/s")
print(f"pre-classified, two streams, 96-row first {t_41:7.1f} us  {B / t_41:6.2f} M QP/s")
