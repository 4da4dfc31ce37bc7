"""Scratch: time and phase-stamp a batch whose robots all have `nst` stance foot-steps (random tables, h = 10).
usage: python tools/dbg/class_probe.py <nst> <B> [B ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
nst = int(sys.argv[1])
for B in [int(x) for x in sys.argv[2:]]:
    h = 10
    rng = np.random.default_rng(5)
    d = W._states(rng, B, h, stairs=True)
    g = np.zeros((B, 4 * h), np.uint8)
    for i in range(B):
        idx = rng.permutation(4 * h)[:nst]
        g[i, idx] = 1
        if g[i, :4].sum() == 0:
            g[i, idx[0]] = 0; g[i, rng.integers(0, 4)] = 1
    b = W._finish(d, B, h, g)
    m = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
    m.set_min_stance(nst); m.set_max_stance(nst)
    m.setup(b["dt"], h, b["mu"], b["f_max"])
    dd = m.upload(b); o = m.alloc_outputs(B); inp, out = m.make_args(dd, o)
    for _ in range(5): m.solve_async(B, inp, out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): m.solve_async(B, inp, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 50
    clk = m.debug_clock(B)
    m.solve_async(B, inp, out); torch.cuda.synchronize()
    c = clk.cpu().numpy().astype(np.float64)
    it = o["iters"].cpu().numpy()
    ph = np.diff(c[:, :8], axis=1)
    print(f"nst={nst} B={B}: {dt*1e6:.1f} us/call  {B/dt:.3e} QP/s  iters {it.mean():.2f}/{it.max()}  phases(median) " +
          " ".join(f"{int(np.median(ph[:,k]))}" for k in range(7)) + f"  total {int(np.median(c[:,7]-c[:,0]))} max {int((c[:,7]-c[:,0]).max())}", flush=True)
    m.close()
