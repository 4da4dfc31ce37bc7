"""Where the 64-row class's five-workgroups-per-CU instantiation starts to pay: batch sweep, qmpc_set_dense 0 against 2 (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
for cfg in (1, 2):
    for B in (1024, 1536, 2048, 3072, 4096, 8192):
        b = W.make_config(cfg, batch=B)
        m = BatchedConvexMPC(0, max_batch=B)
        m.set_max_stance(int((b["gait"] != 0).sum(1).max()))
        m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
        d = m.upload(b); o = m.alloc_outputs(B); inp, out = m.make_args(d, o)
        r = {}
        for mode in (0, 2):
            m.set_dense(mode)
            for _ in range(20): m.solve_async(B, inp, out)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(200): m.solve_async(B, inp, out)
            torch.cuda.synchronize(); r[mode] = B * 200 / (time.perf_counter() - t0)
        print(f"configs[{cfg}] B={B}: four per CU {r[0]:.3e}  five per CU {r[2]:.3e}  ({r[2] / r[0] - 1:+.1%})", flush=True)
        m.close()
