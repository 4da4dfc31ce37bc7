#!/usr/bin/env python
"""Multi-add at the unconstrained minimiser (qmpc_set_multi_add, experimental): static BASELINE shards, plain order, the same
inputs every step (nothing here depends on a previous cycle), cold vs thresholds T = 2, 3, 4, 6.  Per variant: QP/s (median of
9 regions), iterations mean / max, largest relative difference to the cold solution, error-status differences.

    python tools/multi_add.py > gpurun_out/multi_add.json
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quadruped_ctrl_amd import workloads as W  # noqa: E402
from quadruped_ctrl_amd.binding import BatchedConvexMPC  # noqa: E402


def timed(m, B, inp, out, steps, repeats=9):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(10):
        m.solve_async(B, inp, out)
    torch.cuda.synchronize()
    ts = []
    for _ in range(repeats):
        e0.record()
        for _ in range(steps):
            m.solve_async(B, inp, out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / steps)
    return float(np.median(ts))


def main():
    Ts = [0] + [int(a) for a in sys.argv[1:]] if len(sys.argv) > 1 else [0, 2, 3, 4, 6]
    jobs = [("configs[1] trot 1024", W.make_config(1), 200), ("configs[2] mixed 4096", W.make_config(2), 60),
            ("configs[3] trot h16 4096", W.make_config(3, batch=4096), 30), ("configs[4] random 8192", W.make_config(4, batch=8192), 30),
            ("mixed gaits 1024", W.make_config(2, batch=1024), 100), ("random contacts 1024", W.make_config(4, batch=1024), 60)]
    out = []
    for name, b, steps in jobs:
        B = int(b["batch"])
        ref = None
        row = {"workload": name, "batch": B, "variants": {}}
        for T in Ts:
            m = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
            m.set_max_stance(int((b["gait"] != 0).sum(1).max()))
            m.set_min_stance(int((b["gait"] != 0).sum(1).min()))
            m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
            m.set_order_hint(0)
            m.set_multi_add(T)
            d = m.upload(b)
            o = m.alloc_outputs(B, full=True, iters=True)
            inp, outp = m.make_args(d, o)
            ms = timed(m, B, inp, outp, steps)
            sol, it, st = o["soln"].cpu().numpy(), o["iters"].cpu().numpy(), o["status"].cpu().numpy()
            if T == 0:
                ref = (sol.copy(), st.copy(), ms)
            diff = float((np.abs(sol - ref[0]).max(1) / np.maximum(np.abs(ref[0]).max(1), 1.0)).max())
            row["variants"][f"T{T}"] = {"qps": B / ms * 1e3, "ms": ms, "iters_mean": float(it.mean()), "iters_max": int(it.max()),
                                         "max_rel_diff_to_cold": diff, "status_error_bits_differ": int((((st ^ ref[1]) & 47) != 0).sum()),
                                         "failed": int(((st & 47) != 0).sum())}
            print(f"# {name:26s} T={T}: {B / ms * 1e3:.3e} QP/s ({100 * (ref[2] / ms - 1):+5.1f} %)  iters {it.mean():.2f}/{it.max()}  diff {diff:.1e}  "
                  f"failed {int(((st & 47) != 0).sum())}", file=sys.stderr)
            m.close()
        out.append(row)
    print(json.dumps({"multi_add": out}, indent=1))


if __name__ == "__main__":
    main()
