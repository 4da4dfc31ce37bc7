"""Would it pay to start the engines of the HARDEST robots of a decoupled-path call (standing: sweep kernel -> work
items -> engine kernel) while the sweeps of the others are still running?  Emulation from the host: the batch is split by
the iteration counts of a previous solve (what the order hint knows) into the H hardest robots and the rest, solved (a)
as shipped, one call; (b) as two calls on two streams, hardest first (fork / join on the caller's stream)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_ctrl_amd import workloads
from quadruped_ctrl_amd.binding import BatchedConvexMPC

hor = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
steps = 100
b = workloads.make_standing(B, hor)


def subset(d, idx):
    out = {k: (v[idx] if isinstance(v, np.ndarray) and v.shape[:1] == (B,) else v) for k, v in d.items()}
    out["batch"] = int(len(idx))
    return out


def ctx_for(d, cap=None):
    n = d["batch"]
    m = BatchedConvexMPC(0, max_batch=cap or max(n, B), max_horizon=16)   # (the decoupled path is chosen by the HANDLE's size)
    m.set_max_stance(4 * int(d["horizon"]))   # (all feet down: only the class that solves them is launched)
    m.set_min_stance(4 * int(d["horizon"]))
    m.setup(d["dt"], d["horizon"], d["mu"], d["f_max"])
    dv = m.upload(d)
    o = m.alloc_outputs(n, full=False, iters=True)
    inp, out = m.make_args(dv, o)
    return m, n, inp, out, dv, o


def timed(fn):
    for _ in range(10):
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
it = whole[5]["iters"].cpu().numpy()
order = np.argsort(-it, kind="stable")
print(f"standing h={hor} B={B}: iterations mean {it.mean():.1f} max {it.max()};  shipped single call {t_whole:7.1f} us  {B / t_whole:6.3f} M QP/s")
sa, sb = torch.cuda.Stream(priority=-1), torch.cuda.Stream()   # the hardest robots' stream at high priority: its workgroups are dispatched first
for H in (32, 64, 128, 256, 512):
    ch, cr = ctx_for(subset(b, order[:H])), ctx_for(subset(b, order[H:]))
    ev_f = torch.cuda.Event()

    def both():
        ev_f.record(s0)
        sa.wait_event(ev_f)
        sb.wait_event(ev_f)
        ch[0].solve_async(ch[1], ch[2], ch[3], sa)
        cr[0].solve_async(cr[1], cr[2], cr[3], sb)
        e1, e2 = torch.cuda.Event(), torch.cuda.Event()
        e1.record(sa)
        e2.record(sb)
        s0.wait_event(e1)
        s0.wait_event(e2)

    t_h = timed(lambda: ch[0].solve_async(ch[1], ch[2], ch[3], s0))
    t_r = timed(lambda: cr[0].solve_async(cr[1], cr[2], cr[3], s0))
    t_b = timed(both)
    print(f"  hardest {H:4d} on their own stream: alone {t_h:6.1f} us, the rest alone {t_r:6.1f} us, together {t_b:7.1f} us  {B / t_b:6.3f} M QP/s ({100 * (t_whole / t_b - 1):+.1f} %)")
    ch[0].close(); cr[0].close()
