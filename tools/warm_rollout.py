#!/usr/bin/env python
"""Sliding-window rollout: cold start vs warm start across MPC cycles.

Every cycle BOTH solvers get the same record (the rollout is advanced with the cold solution, so the
two never diverge); the warm handle keeps its working-set buffer between cycles (qmpc_set_warm_start,
shift = 1 horizon step).  Reports per gait: mean / max active-set iterations, mean solve time per cycle
(HIP events around the solve alone), and the largest relative difference between the two solutions.

    python tools/warm_rollout.py [--batch 1024] [--cycles 40] > gpurun_out/warm_rollout.json
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


def run(gait, horizon, B, cycles, seed=0, kick=1.0, demand=None, label=None):
    ro = W.Rollout(B, horizon, gait, seed=seed, kick=kick)
    if demand:
        ro.demand(*demand)
    b = ro.record()
    cold = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
    warm = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
    for m in (cold, warm):
        m.setup(b["dt"], horizon, b["mu"], b["f_max"])
    warm.warm_start(B, shift_steps=1)
    oc, ow = cold.alloc_outputs(B, full=True), warm.alloc_outputs(B, full=True)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    rows = []
    for c in range(cycles):
        b = ro.record()
        d = cold.upload(b)
        ic, outc = cold.make_args(d, oc)
        iw, outw = warm.make_args(d, ow)
        cold.solve_async(B, ic, outc)          # untimed: same clocks / caches for both timed solves
        torch.cuda.synchronize()
        e[0].record(); cold.solve_async(B, ic, outc); e[1].record()
        e[2].record(); warm.solve_async(B, iw, outw); e[3].record()
        torch.cuda.synchronize()
        sc, sw = oc["soln"].cpu().numpy(), ow["soln"].cpu().numpy()
        bad = int(((oc["status"].cpu().numpy() & 47) != 0).sum() + ((ow["status"].cpu().numpy() & 47) != 0).sum())
        diff = float((np.abs(sc - sw).max(1) / np.maximum(np.abs(sc).max(1), 1.0)).max())
        itc, itw = oc["iters"].cpu().numpy(), ow["iters"].cpu().numpy()
        rows.append(dict(cycle=c, cold_ms=e[0].elapsed_time(e[1]), warm_ms=e[2].elapsed_time(e[3]),
                         cold_iters=float(itc.mean()), warm_iters=float(itw.mean()), cold_max=int(itc.max()),
                         warm_max=int(itw.max()), max_rel_diff=diff, failed=bad,
                         fallback=int(((ow["status"].cpu().numpy() & 16) != 0).sum())))
        ro.advance(oc["grf"].cpu().numpy())
    cold.close(); warm.close()
    steady = rows[3:]                      # the first cycles have nothing to warm-start from
    agg = lambda k: float(np.mean([r[k] for r in steady]))
    return {"scenario": label or f"{gait}, random pushes x{kick}", "gait": gait, "horizon": horizon, "batch": B, "cycles": cycles,
            "cold_iters_mean": agg("cold_iters"), "warm_iters_mean": agg("warm_iters"),
            "cold_iters_max": max(r["cold_max"] for r in steady), "warm_iters_max": max(r["warm_max"] for r in steady),
            "cold_ms_mean": agg("cold_ms"), "warm_ms_mean": agg("warm_ms"),
            "max_rel_diff": max(r["max_rel_diff"] for r in rows), "failed": sum(r["failed"] for r in rows),
            "warm_fallbacks": sum(r["fallback"] for r in rows), "first_cycles": rows[:4]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1024)
    ap.add_argument("--cycles", type=int, default=40)
    a = ap.parse_args()
    out = [run("trot", 10, a.batch, a.cycles), run("mixed", 10, a.batch, a.cycles), run("stand", 10, a.batch, a.cycles),
           run("trot", 16, a.batch, a.cycles), run("stand", 14, min(a.batch, 512), a.cycles),
           # sustained demands: accelerate from 0.5 to 1.6 m/s with a lateral command and a turn, light pushes
           run("trot", 10, a.batch, a.cycles, kick=0.25, demand=(1.6, 0.4, 1.0), label="trot, accelerate + turn, light pushes"),
           run("mixed", 10, a.batch, a.cycles, kick=0.25, demand=(1.6, 0.4, 1.0), label="mixed gaits, accelerate + turn, light pushes"),
           run("stand", 10, a.batch, a.cycles, kick=0.25, demand=(0.0, 0.0, 0.0), label="stand, braking from 0.5 m/s, light pushes"),
           run("trot", 10, a.batch, a.cycles, kick=0.0, demand=(1.6, 0.4, 1.0), label="trot, accelerate + turn, no pushes")]
    for r in out:
        print(f"# {r['scenario']:50s} h={r['horizon']:2d}: iters {r['cold_iters_mean']:.2f} -> {r['warm_iters_mean']:.2f} "
              f"(max {r['cold_iters_max']} -> {r['warm_iters_max']}), ms/cycle {r['cold_ms_mean']:.4f} -> {r['warm_ms_mean']:.4f}, "
              f"max rel diff {r['max_rel_diff']:.1e}, failed {r['failed']}, fallbacks {r['warm_fallbacks']}", file=sys.stderr)
    print(json.dumps({"warm_rollout": out}, indent=1))


if __name__ == "__main__":
    main()
