"""Shader-clock stamps of the large-problem producer (build with QMPC_EXTRA_HIPFLAGS="-DQMPC_BIG_STAMP=<step>").
usage: python tools/big_phase.py <horizon> <trot|stand> [robots]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
h = int(sys.argv[1]); gait = sys.argv[2]; B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
b = W.make_long_horizon(B, h, gait)
m = BatchedConvexMPC(0, max_batch=B, max_horizon=36)
m.setup(b["dt"], h, b["mu"], b["f_max"])
d = m.upload(b); o = m.alloc_outputs(B); inp, out = m.make_args(d, o)
for _ in range(2): m.solve_async(B, inp, out)
torch.cuda.synchronize()
clk = m.debug_clock(B)
m.solve_async(B, inp, out); torch.cuda.synchronize()
c = clk.cpu().numpy().astype(np.float64)
it = o["iters"].cpu().numpy()
nst = (b["gait"].reshape(B, -1) != 0).sum(1)
keep = (it < 20) & (3 * nst > 192) & (c[:, 5] > 0)
c = c[keep]
med = lambda x: float(np.median(x))
print(f"h={h} {gait} B={B}: {keep.sum()} robots on the large path with < 20 iterations; n_r {3*nst.min()}..{3*nst.max()}")
print("  one block step: panel loads %.0f | factor, L^-1, P^-1 (one wave) %.0f | F %.0f | pivot columns + update %.0f | last barrier %.0f | whole %.0f" % (
    med(c[:, 1] - c[:, 0]), med(c[:, 2] - c[:, 1]), med(c[:, 3] - c[:, 2]), med(c[:, 4] - c[:, 3]), med(c[:, 5] - c[:, 4]), med(c[:, 5] - c[:, 0])))
print("  the one-wave part: factor %.0f | L^-1, P^-1 and the barrier %.0f" % (med(c[:, 7] - c[:, 1]), med(c[:, 2] - c[:, 7])))
print("  kernel: g + fill of H %.0f | block sweep %.0f | mirror %.0f | x_u %.0f | whole tail %.0f" % (
    med(c[:, 9] - c[:, 8]), med(c[:, 10] - c[:, 9]), med(c[:, 14] - c[:, 10]), med(c[:, 15] - c[:, 14]), med(c[:, 15] - c[:, 8])))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
m.debug_off()
e0.record(); m.solve_async(B, inp, out); e1.record(); torch.cuda.synchronize()
print("  one call: %.3f ms" % e0.elapsed_time(e1))
