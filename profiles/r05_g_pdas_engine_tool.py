#!/usr/bin/env python
"""Primal-dual active-set start of the decoupled engine (qmpc_set_pdas) against the one-row iteration, on the workloads of
the 128- / 192-row classes: QP/s of both (median of 7 regions of K steps), iteration statistics, how many robots the start
answered, largest relative difference of the solutions, error-status differences; --model: the numpy twin
(oracle/pdas_model.py) on a sample -- same number of solves.

    python tools/pdas_engine.py [--cap 16] [--model] > gpurun_out/pdas_engine.json
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quadruped_ctrl_amd import workloads as W  # noqa: E402
from quadruped_ctrl_amd.binding import BatchedConvexMPC  # noqa: E402


def timed(m, B, inp, out, steps, repeats=7):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(5):
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--cap", type=int, default=16)
    ap.add_argument("--model", action="store_true")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    low = W.make_standing(1024, 10)
    low["f_max"] = 40.0
    jobs = [("standing h10 braking", W.make_standing(1024, 10), 30), ("standing h10 calm", W.make_standing(1024, 10, calm=True), 30),
            ("standing h14 braking", W.make_standing(1024, 14), 20), ("standing h16 braking", W.make_standing(1024, 16), 20),
            ("standing h10 f_max 40", low, 20), ("trot h24 (192-row class)", W.make_long_horizon(1024, 24, "trot"), 20),
            ("bounding h36", W.make_long_horizon(512, 36, "bound"), 10)]
    if a.quick:
        jobs = jobs[:1]
    out = []
    for name, b, steps in jobs:
        B = int(b["batch"])
        res = {}
        for cap in (0, a.cap):
            m = BatchedConvexMPC(0, max_batch=B, max_horizon=max(16, int(b["horizon"])))
            m.set_max_stance(int((b["gait"] != 0).sum(1).max()))
            m.set_min_stance(int((b["gait"] != 0).sum(1).min()))
            m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
            m.set_pdas(cap)
            d = m.upload(b)
            o = m.alloc_outputs(B, full=True, iters=True)
            inp, outp = m.make_args(d, o)
            ms = timed(m, B, inp, outp, steps)
            res[cap] = dict(ms=ms, sol=o["soln"].cpu().numpy(), it=o["iters"].cpu().numpy(), st=o["status"].cpu().numpy())
            m.close()
        c, p = res[0], res[a.cap]
        diff = np.abs(p["sol"] - c["sol"]).max(1) / np.maximum(np.abs(c["sol"]).max(1), 1.0)
        by = (p["st"] & 256) != 0
        row = {"workload": name, "batch": B, "qps_one_row": B / c["ms"] * 1e3, "qps_pdas": B / p["ms"] * 1e3, "gain": c["ms"] / p["ms"] - 1.0,
               "iters_one_row": [float(c["it"].mean()), int(c["it"].max())],
               "solves_pdas_robots": [float(p["it"][by].mean()) if by.any() else 0.0, int(p["it"][by].max()) if by.any() else 0],
               "answered_by_pdas": int(by.sum()), "left_to_the_iteration": int((~by).sum()),
               "max_rel_diff": float(diff.max()), "status_error_bits_differ": int((((p["st"] ^ c["st"]) & 47) != 0).sum()),
               "failed": int(((p["st"] & 47) != 0).sum())}
        if a.model:
            from oracle import pdas_model as PM
            same, idx = 0, list(range(0, B, max(1, B // 24)))
            for i in idx:
                q, it, ok, kmax = PM.solve_robot(b, i, max_it=a.cap, kp=64)
                same += int(ok == bool(by[i]) and (not ok or it == p["it"][i]))
            row["model_agrees_on"] = f"{same}/{len(idx)}"
        out.append(row)
        print(f"# {name:26s}: {row['qps_one_row']:.3e} -> {row['qps_pdas']:.3e} QP/s ({100 * row['gain']:+5.1f} %)  one-row iters {row['iters_one_row'][0]:.2f}/{row['iters_one_row'][1]}  "
              f"PDAS solves {row['solves_pdas_robots'][0]:.2f}/{row['solves_pdas_robots'][1]}  answered {row['answered_by_pdas']}  left {row['left_to_the_iteration']}  "
              f"diff {row['max_rel_diff']:.1e}  status-differs {row['status_error_bits_differ']} failed {row['failed']}" + (f"  model {row['model_agrees_on']}" if a.model else ""), file=sys.stderr)
    print(json.dumps({"pdas_engine": out}, indent=1))


if __name__ == "__main__":
    main()
