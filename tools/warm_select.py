#!/usr/bin/env python
"""Selective warm start: cold / warm for every robot / warm only for the robots the previous cycle found
hard (qmpc_set_warm_start_min_iters), on closed-loop rollouts seeded from the BASELINE configs (workloads.ConfigRollout) and
on workloads.Rollout.  Every cycle ALL handles solve the same record (the rollout advances with the cold solution).
Reported per variant: ms / cycle (HIP events around the solve alone, mean over the steady cycles), the DISTRIBUTION of the
launch's maximum iteration count (what a one-round launch waits for), mean iterations, how many robots started warm, the
largest relative difference to the cold solution and whether the error bits of the status agree.

    python tools/warm_select.py [--cycles 24] > gpurun_out/warm_select.json
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


def handle(b, B, warm, min_iters, stance):
    m = BatchedConvexMPC(0, max_batch=B, max_horizon=max(16, int(b["horizon"])))
    if stance:
        m.set_max_stance(stance[1])
        m.set_min_stance(stance[0])
    m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
    m.set_order_hint(1)
    ws = None
    if warm:
        ws = m.warm_start(B, shift_steps=1)
        m.warm_start_min_iters(min_iters)
    return m, ws


def run(name, ro, B, cycles, variants, stance=None):
    b = ro.record()
    hs = {}
    for v, (warm, mi) in variants.items():
        m, ws = handle(b, B, warm, mi, stance)
        hs[v] = dict(m=m, ws=ws, o=m.alloc_outputs(B, full=True), ms=[], itmax=[], itmean=[], diff=0.0, stbad=0, nwarm=[])
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for c in range(cycles):
        b = ro.record()
        ref = None
        for v, H in hs.items():
            d = H["m"].upload(b)
            i_, o_ = H["m"].make_args(d, H["o"])
            if H["ws"] is not None:
                keep = H["ws"].clone()
                hk = None
            # one untimed call would advance the warm state twice: time the ONE call of the cycle (after a cold handle's
            # warm-up call has put clocks / caches in the same state for everybody)
            if v == "cold":
                H["m"].solve_async(B, i_, o_)
                torch.cuda.synchronize()
            ev[0].record()
            H["m"].solve_async(B, i_, o_)
            ev[1].record()
            torch.cuda.synchronize()
            it = H["o"]["iters"].cpu().numpy()
            st = H["o"]["status"].cpu().numpy()
            sol = H["o"]["soln"].cpu().numpy()
            if v == "cold":
                ref = (sol.copy(), st.copy())
            else:
                H["diff"] = max(H["diff"], float((np.abs(sol - ref[0]).max(1) / np.maximum(np.abs(ref[0]).max(1), 1.0)).max()))
                H["stbad"] += int((((st ^ ref[1]) & 47) != 0).sum())
            if c >= 3:
                H["ms"].append(ev[0].elapsed_time(ev[1]))
                H["itmax"].append(int(it.max()))
                H["itmean"].append(float(it.mean()))
        ro.advance(hs["cold"]["o"]["grf"].cpu().numpy())
    out = {"scenario": name, "batch": B, "cycles": cycles, "variants": {}}
    for v, H in hs.items():
        im = np.array(H["itmax"])
        out["variants"][v] = {"ms_per_cycle": float(np.mean(H["ms"])), "ms_median": float(np.median(H["ms"])),
                              "iters_mean": float(np.mean(H["itmean"])),
                              "launch_max_iters": {"min": int(im.min()), "median": float(np.median(im)), "mean": float(im.mean()), "max": int(im.max()),
                                                   "per_cycle": [int(x) for x in im]},
                              "max_rel_diff_to_cold": H["diff"], "status_error_bits_differ": H["stbad"]}
        H["m"].close()
    base = out["variants"]["cold"]["ms_per_cycle"]
    for v, r in out["variants"].items():
        print(f"# {name:34s} B={B:5d} {v:12s}: {r['ms_per_cycle']:.4f} ms/cycle ({100 * (base / r['ms_per_cycle'] - 1):+5.1f} %)  iters mean {r['iters_mean']:.2f}  "
              f"launch max: median {r['launch_max_iters']['median']:.0f} mean {r['launch_max_iters']['mean']:.1f} max {r['launch_max_iters']['max']}  "
              f"diff {r['max_rel_diff_to_cold']:.1e} status-differs {r['status_error_bits_differ']}", file=sys.stderr)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=24)
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    variants = {"cold": (False, 0), "warm_all": (True, 0), "warm_ge3": (True, 3), "warm_ge5": (True, 5), "warm_ge8": (True, 8)}
    out = []
    jobs = [("configs[1] closed loop", lambda: W.ConfigRollout(W.make_config(1), periodic=True), 1024, (20, 20)),
            ("Rollout trot h10 pushes", lambda: W.Rollout(1024, 10, "trot", seed=3), 1024, None),
            ("Rollout mixed h10 pushes", lambda: W.Rollout(1024, 10, "mixed", seed=3), 1024, None),
            ("configs[2] closed loop", lambda: W.ConfigRollout(W.make_config(2), periodic=True), 4096, (16, 20)),
            ("configs[4] closed loop, 8192", lambda: W.ConfigRollout(W.make_config(4, batch=8192), periodic=False), 8192, None)]
    if a.quick:
        jobs = jobs[:2]
    for name, mk, B, stance in jobs:
        out.append(run(name, mk(), B, a.cycles, variants, stance))
    print(json.dumps({"warm_select": out}, indent=1))


if __name__ == "__main__":
    main()
