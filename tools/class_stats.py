"""Per size class: robots, active-set iterations, fallback (Schur engine re-run) and failure counts of one
batched solve (GPU).  usage: python tools/class_stats.py --config 4 | --workload standing --horizon 14"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime load order)
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=4)
ap.add_argument("--batch", type=int, default=None)
ap.add_argument("--workload", choices=["config", "standing", "trot"], default="config")
ap.add_argument("--horizon", type=int, default=10)
a = ap.parse_args()
if a.workload == "config":
    b = W.make_config(a.config, batch=a.batch) if a.batch else W.make_config(a.config)
else:
    b = (W.make_standing if a.workload == "standing" else W.make_trot)(a.batch or 1024, a.horizon)
B = b["batch"]
mpc = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
res = mpc.solve(b, full=True)
rows = 3 * b["gait"].astype(int).sum(1)
st, it = res["status"], res["iters"]
for lo, hi in ((0, 64), (65, 96), (97, 128), (129, 192)):
    m = (rows >= lo) & (rows <= hi)
    if not m.any():
        continue
    print(f"rows {lo:3d}..{hi:3d}: robots {int(m.sum()):6d}  iters mean {it[m].mean():6.2f} max {int(it[m].max()):3d}  "
          f"fallback {int(((st[m] & 16) != 0).sum()):5d}  compacted {int(((st[m] & 64) != 0).sum()):5d}  spilled {int(((st[m] & 128) != 0).sum()):5d}  errors {int(((st[m] & 47) != 0).sum())}")
    hist = np.bincount(np.minimum(it[m], 120) // 4)
    print("   iters histogram (bins of 4):", hist.tolist())
    odd = m & ((st & (16 | 47)) != 0)
    if odd.any():
        print("   fallback / error robots: iters", it[odd].tolist(), "status", st[odd].tolist())
