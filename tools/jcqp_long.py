"""Throughput of the JCQP alternate on the large-problem path (use_jcqp = 1 / 2 at horizons above 16).  GPU box: python tools/jcqp_long.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
for h, gait, mode, B in ((20, "stand", 1, 1024), (36, "trot", 2, 1024), (36, "stand", 1, 1024)):
    b = W.make_long_horizon(B, h, gait)
    m = BatchedConvexMPC(0, max_batch=B, max_horizon=36)
    m.setup(b["dt"], h, b["mu"], b["f_max"])
    m.settings_jcqp(mode)
    d = m.upload(b); o = m.alloc_outputs(B); inp, out = m.make_args(d, o)
    m.solve_async(B, inp, out); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        m.solve_async(B, inp, out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    it = o["iters"].float().mean().item()
    print(f"use_jcqp={mode} h={h} {gait} (n = {12 * h if mode == 1 else int(3 * (b['gait'] != 0).sum(1).max())}): {B / dt:.3e} robots/s "
          f"({dt * 1e3:.1f} ms per {B}), ADMM iterations mean {it:.0f}, failed {int(((o['status'] & 46) != 0).sum())}")
    m.close()
