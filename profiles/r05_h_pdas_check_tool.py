#!/usr/bin/env python
"""Primal-dual active-set pre-solver (qmpc_set_pdas) against the one-row iteration: static BASELINE shards, plain order.
Per workload: QP/s of both (median of 9 regions), iteration statistics, how many robots the pre-solver answered, the
largest relative difference of the solutions, error-status differences; with --model also the numpy twin
(oracle/pdas_model.py) on a sample: same number of solves.

    python tools/pdas_check.py [--cap 12] > gpurun_out/pdas_check.json
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--cap", type=int, default=12)
    ap.add_argument("--model", action="store_true")
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    lowf = W.make_config(1, batch=1024)
    lowf["f_max"] = 30.0
    jobs = [("configs[1] trot 1024", W.make_config(1), 200), ("configs[2] mixed 4096", W.make_config(2), 60),
            ("configs[4] random 8192", W.make_config(4, batch=8192), 30), ("configs[3] trot h16 4096", W.make_config(3, batch=4096), 30),
            ("mixed gaits 1024", W.make_config(2, batch=1024), 100), ("trot f_max 30, 1024", lowf, 60),
            ("trot 16384", W.make_config(1, batch=16384), 20)]
    if a.quick:
        jobs = jobs[:2]
    out = []
    for name, b, steps in jobs:
        B = int(b["batch"])
        res = {}
        for cap in (0, a.cap):
            m = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
            m.set_max_stance(int((b["gait"] != 0).sum(1).max()))
            m.set_min_stance(int((b["gait"] != 0).sum(1).min()))
            m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
            m.set_order_hint(0)
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
               "iters_one_row": [float(c["it"].mean()), int(c["it"].max())], "solves_pdas_robots": [float(p["it"][by].mean()) if by.any() else 0.0, int(p["it"][by].max()) if by.any() else 0],
               "answered_by_pdas": int(by.sum()), "handed_over": int((~by).sum()), "handed_over_iters_max": int(p["it"][~by].max()) if (~by).any() else 0,
               "max_rel_diff": float(diff.max()), "status_error_bits_differ": int((((p["st"] ^ c["st"]) & 47) != 0).sum()),
               "failed": int(((p["st"] & 47) != 0).sum())}
        if a.model:
            from oracle import pdas_model as PM
            same = 0
            idx = list(range(0, B, max(1, B // 48)))
            for i in idx:
                q, it, ok, kmax = PM.solve_robot(b, i, max_it=a.cap, kp=28)
                same += int(ok == bool(by[i]) and (not ok or it == p["it"][i]))
            row["model_agrees_on"] = f"{same}/{len(idx)}"
        out.append(row)
        print(f"# {name:26s}: {row['qps_one_row']:.3e} -> {row['qps_pdas']:.3e} QP/s ({100 * row['gain']:+5.1f} %)  one-row iters {row['iters_one_row'][0]:.2f}/{row['iters_one_row'][1]}  "
              f"PDAS solves {row['solves_pdas_robots'][0]:.2f}/{row['solves_pdas_robots'][1]}  answered {row['answered_by_pdas']}  handed over {row['handed_over']} (their iters max {row['handed_over_iters_max']})  "
              f"diff {row['max_rel_diff']:.1e}  status-differs {row['status_error_bits_differ']} failed {row['failed']}" + (f"  model {row['model_agrees_on']}" if a.model else ""), file=sys.stderr)
    print(json.dumps({"pdas_check": out}, indent=1))


if __name__ == "__main__":
    main()
