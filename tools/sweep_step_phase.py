"""Profiling aid (build with -DQMPC_SWEEP_STAMP=<k0>): shader-clock stamps inside ONE step of the multi-wave sweep (thread 0's view).
usage: QMPC_LIB=variants/<v>/libqmpc.so python tools/sweep_step_phase.py <config> <batch>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
cfg = sys.argv[1] if len(sys.argv) > 1 else "3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
b = (W.make_standing(B, int(cfg[1:])) if cfg[0] == "s" else W.make_config(int(cfg), batch=B))
mpc = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
mpc.set_min_stance(int((b["gait"] != 0).sum(1).min())); mpc.set_max_stance(int((b["gait"] != 0).sum(1).max()))
mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
if os.environ.get("QMPC_NO_SPLIT") is None and cfg[0] == "s": mpc.set_split(0)
d = mpc.upload(b); o = mpc.alloc_outputs(B); inp, out = mpc.make_args(d, o)
for _ in range(3): mpc.solve_async(B, inp, out)
torch.cuda.synchronize()
clk = mpc.debug_clock(2 * B)
mpc.solve_async(B, inp, out); torch.cuda.synchronize()
c = clk.cpu().numpy().astype(np.float64)[B:, :6]
c = c[(c > 0).all(1)]
d = np.diff(c, axis=1)
names = ["loads + F (nf)", "first block fmacs", "publish", "rest of the fmacs", "finalise + barrier"]
print(f"cfg{cfg} B={B}: one sweep step, thread 0 (median cycles over {len(c)} robots); whole step {np.median(c[:,5]-c[:,0]):.0f}")
for k, nm in enumerate(names):
    print(f"  {nm:22s} {np.median(d[:,k]):8.0f}")
