"""Where do sporadic slow calls come from?  Long runs of per-call HIP-event timing (no profiler) for
  cfg4   configs[4], 8192 robots (64-row kernel + 96-row list kernel with overflow slices)      -- the kernel VERDICT r5 flagged
  cfg3   configs[3], 4096 robots (96-row kernel as the first class, no overflow slices taken)
  cfg1   configs[1], 1024 robots (64-row kernel only)
  ctrl   a control that contains no code of this library: torch's elementwise kernel on a 64 MiB tensor (~50 us)
each for `seconds` of GPU time; prints every call beyond 1.5 x the run's median with its wall-clock position, so that a
periodic system-level stall shows up as such.   python tools/outlier_long.py [seconds]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quadruped_ctrl_amd import workloads  # noqa: E402
from quadruped_ctrl_amd.binding import BatchedConvexMPC  # noqa: E402


def log(name, step, seconds, block=1000):
    st = torch.cuda.current_stream(0)
    for _ in range(200):
        step()
    torch.cuda.synchronize()
    ms, wall = [], []
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < seconds:
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(block + 1)]
        w0 = time.perf_counter() - t_start
        ev[0].record(st)
        for k in range(block):
            step()
            ev[k + 1].record(st)
        torch.cuda.synchronize()
        w1 = time.perf_counter() - t_start
        d = [ev[k].elapsed_time(ev[k + 1]) for k in range(block)]
        ms += d
        cum = np.cumsum(d)
        wall += list(w0 + (w1 - w0) * cum / cum[-1])
    ms = np.array(ms)
    wall = np.array(wall)
    med = float(np.median(ms))
    idx = np.flatnonzero(ms > 1.5 * med)
    return {"run": name, "calls": int(ms.size), "gpu_seconds": float(ms.sum() / 1e3), "median_ms": med,
            "p99_ms": float(np.percentile(ms, 99)), "p9999_ms": float(np.percentile(ms, 99.99)), "max_ms": float(ms.max()),
            "max_over_median": float(ms.max() / med), "n_over_1.5x": int(idx.size), "n_over_2x": int((ms > 2 * med).sum()),
            "slow_calls": [{"call": int(i), "ms": float(ms[i]), "excess_ms": float(ms[i] - med), "wall_s": float(wall[i])} for i in idx[:60]]}


def solver(config, batch):
    b = workloads.make_config(config, batch=batch)
    mpc = BatchedConvexMPC(0, max_batch=batch, max_horizon=16)
    mpc.set_max_stance(int((b["gait"] != 0).sum(1).max()))
    mpc.set_min_stance(int((b["gait"] != 0).sum(1).min()))
    mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
    mpc.set_order_hint(0)
    d = mpc.upload(b)
    o = mpc.alloc_outputs(batch, full=False, iters=True)
    inp, out = mpc.make_args(d, o)
    st = torch.cuda.current_stream(0)
    return mpc, (d, o), (lambda: mpc.solve_async(batch, inp, out, st))


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    x = torch.zeros(8 << 20, dtype=torch.float64, device="cuda:0")
    print(json.dumps(log("ctrl_torch_elementwise_64MiB", lambda: x.add_(1.0), seconds)), flush=True)
    for name, cfg, batch in (("cfg4_8192", 4, 8192), ("cfg3_4096", 3, 4096), ("cfg1_1024", 1, 1024)):
        mpc, keep, step = solver(cfg, batch)
        r = log(name, step, seconds)
        c = mpc.debug_read_counts()
        r["overflow_counters_last_calls"] = {"slices_taken": int(max(c[0][7], c[1][7])), "probes_busy": int(max(c[0][16], c[1][16])),
                                             "timeouts": int(max(c[0][17], c[1][17]))}
        print(json.dumps(r), flush=True)
        mpc.close()
