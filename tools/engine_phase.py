"""Phase stamps of ONE iteration (QMPC_EDBG_ITER, default 20) of the decoupled engine: engine wave and holder 1.
GPU box:  python tools/engine_phase.py s10 256   (s<h> = standing at horizon h)"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from quadruped_ctrl_amd import workloads as W  # noqa: E402
from quadruped_ctrl_amd.binding import BatchedConvexMPC  # noqa: E402

cfg, B = sys.argv[1], int(sys.argv[2])
b = W.make_standing(B, int(cfg[1:])) if cfg[0] == "s" else W.make_config(int(cfg), batch=B)
mpc = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
nst = (b["gait"] != 0).sum(1)
mpc.set_max_stance(int(nst.max()))
mpc.set_min_stance(int(nst.min()))
mpc.set_split(2)   # the decoupled path whatever the handle's size
d = mpc.upload(b)
o = mpc.alloc_outputs(B)
inp, out = mpc.make_args(d, o)
for _ in range(3):
    mpc.solve_async(B, inp, out)
torch.cuda.synchronize()
clk = mpc.debug_clock(B)
mpc.solve_async(B, inp, out)
torch.cuda.synchronize()
c = clk.cpu().numpy().astype(np.float64)
it = o["iters"].cpu().numpy()
ok = (it > 22) & (c[:, 7] > c[:, 0]) & (c[:, 0] > 0)
m = lambda a: float(np.median(a[ok]))
print(f"{cfg} B={B}: robots {ok.sum()}  ENGINE wave, median cycles: select {m(c[:,1]-c[:,0]):.0f} | request+(A) {m(c[:,2]-c[:,1]):.0f} | "
      f"own events + columns {m(c[:,3]-c[:,2]):.0f} | (B)+sums {m(c[:,4]-c[:,3]):.0f} | delta,ratio {m(c[:,5]-c[:,4]):.0f} | "
      f"step,event {m(c[:,7]-c[:,5]):.0f} | whole {m(c[:,7]-c[:,0]):.0f}")
okh = ok & (c[:, 11] > c[:, 8]) & (c[:, 8] > 0)
mh = lambda a: float(np.median(a[okh]))
print(f"   HOLDER 1: request read + ingest {mh(c[:,9]-c[:,8]):.0f} | accumulate {mh(c[:,10]-c[:,9]):.0f} | partial sums out {mh(c[:,11]-c[:,10]):.0f}"
      f" | (A)->(B) {mh(c[:,11]-c[:,8]):.0f}")
okb = (c[:, 13] > c[:, 12]) & (c[:, 12] > 0) & (c[:, 6] > c[:, 13])
print(f"   block start: median {np.median((c[:,13]-c[:,12])[okb]):.0f} cycles (max {np.max((c[:,13]-c[:,12])[okb]):.0f}); "
      f"rest of the item (iteration + output) median {np.median((c[:,6]-c[:,13])[okb]):.0f} (max {np.max((c[:,6]-c[:,13])[okb]):.0f}); iters mean {it.mean():.1f} max {it.max()}")
