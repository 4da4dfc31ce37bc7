import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
B = 1024
b = W.make_config(1, batch=B)
for so in (0, 1):
    mpc = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
    mpc.set_max_stance(int((b["gait"] != 0).sum(1).max())); mpc.set_min_stance(int((b["gait"] != 0).sum(1).min()))
    mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
    mpc.set_order_hint(0); mpc.set_size_order(so)
    d = mpc.upload(b); o = mpc.alloc_outputs(B); inp, out = mpc.make_args(d, o)
    for _ in range(3): mpc.solve_async(B, inp, out)
    torch.cuda.synchronize()
    clk = mpc.debug_clock(B)
    mpc.solve_async(B, inp, out); torch.cuda.synchronize()
    c = clk.cpu().numpy().astype(np.float64); it = o["iters"].cpu().numpy()
    t0 = c[:, 0].min()
    top = np.argsort(-it)[:10]
    print("size_order", so, "launch length (cycles)", (c[:, 7].max() - t0))
    for r in top:
        print("  robot %4d iters %2d rank %4d start %6.0f  inputs %6.0f E/s %6.0f asm %6.0f sweep %6.0f xu %6.0f as %6.0f end %7.0f" % (
            r, it[r], c[r, 14], c[r, 0] - t0, c[r, 1] - c[r, 0], c[r, 2] - c[r, 1], c[r, 3] - c[r, 2], c[r, 4] - c[r, 3], c[r, 5] - c[r, 4], c[r, 6] - c[r, 5], c[r, 7] - t0))
    d_ = np.diff(c[:, :8], axis=1)
    print("  medians: inputs %.0f E/s %.0f asm %.0f sweep %.0f xu %.0f as %.0f" % tuple(np.median(d_[:, k]) for k in range(6)))
    if so:
        rk = c[:, 14]
        print("  ranks: min %d max %d; robots with rank < 8: %d, < 42: %d; corr(rank, iters) %.2f" % (rk.min(), rk.max(), (rk < 8).sum(), (rk < 42).sum(), np.corrcoef(rk, it)[0, 1]))
    mpc.close()
