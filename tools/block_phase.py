import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
B=256
b = W.make_standing(B, 10)
mpc = BatchedConvexMPC(0, max_batch=B, max_horizon=16); mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
mpc.set_max_stance(40); mpc.set_min_stance(40); mpc.set_split(2); mpc.set_block_start(True)
d = mpc.upload(b); o = mpc.alloc_outputs(B); inp, out = mpc.make_args(d, o)
for _ in range(3): mpc.solve_async(B, inp, out)
torch.cuda.synchronize()
clk = mpc.debug_clock(B); mpc.solve_async(B, inp, out); torch.cuda.synchronize()
c = clk.cpu().numpy().astype(np.float64)
ok = (c[:,7] > c[:,0]) & (c[:,0] > 0)
m = lambda a: float(np.median(a[ok]))
print("robots", ok.sum(), "forced addition at 20 records, cycles: accumulate (y on the fly)", m(c[:,3]-c[:,0]), "| barrier", m(c[:,4]-c[:,3]), "| delta, record, x, lam", m(c[:,5]-c[:,4]), "| barrier", m(c[:,7]-c[:,5]), "| whole", m(c[:,7]-c[:,0]))
it = o["iters"].cpu().numpy()
print("working-set changes mean", it.mean(), "max", it.max())
