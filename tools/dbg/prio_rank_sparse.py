import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
from oracle import sparse_model as SM
from tools.order_hint import timed
B = 1024
b = dict(W.make_config(1, batch=B))
b["mu"] = SM.SPARSE_MU
b["weights"] = np.tile(SM.SPARSE_WEIGHTS.astype(np.float32), (B, 1))
for so in (0, 1, 0, 1):
    mpc = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
    mpc.set_max_stance(int((b["gait"] != 0).sum(1).max())); mpc.set_min_stance(int((b["gait"] != 0).sum(1).min()))
    mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
    mpc.set_robot(9.0, (0.07, 0.26, 0.242), -9.81); mpc.set_model(1)
    mpc.set_order_hint(0); mpc.set_size_order(so)
    d = mpc.upload(b); o = mpc.alloc_outputs(B); inp, out = mpc.make_args(d, o)
    ms = timed(mpc, B, inp, out, 200)
    clk = mpc.debug_clock(B)
    mpc.solve_async(B, inp, out); torch.cuda.synchronize()
    c = clk.cpu().numpy().astype(np.float64); it = o["iters"].cpu().numpy()
    d_ = np.diff(c[:, :8], axis=1)
    print("size_order", so, "ms %.4f qps %.3e" % (ms, B / ms * 1e3), "iters mean %.3f" % it.mean(),
          "medians: inputs %.0f E/s %.0f asm %.0f sweep %.0f xu %.0f as %.0f" % tuple(np.median(d_[:, k]) for k in range(6)),
          "ranks", np.unique(c[:, 14], return_counts=True) if so else "")
    mpc.close()
