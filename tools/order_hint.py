#!/usr/bin/env python
"""Order hint (qmpc_set_order_hint): multi-round launches take the robots hardest first, by the iteration counts the
handle's previous call left.  Two measurements:

  (a) the bench workloads (BASELINE configs 2 / 3 / 4, trot at 16384), the same inputs every step -- the hint is EXACT,
      the upper bound of what the order can give;
  (b) closed-loop rollouts (workloads.Rollout: every cycle the contact table advances one segment, the state is
      integrated with the returned forces, random pushes): the hint is the PREVIOUS cycle's count of the same robot -- what a
      controller gets.  Both handles solve every cycle's record; results must be bit-identical.

    python tools/order_hint.py > gpurun_out/order_hint.json
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


def handle(b, B, hint, stance=True):
    m = BatchedConvexMPC(0, max_batch=B, max_horizon=max(16, int(b["horizon"])))
    if stance:
        m.set_max_stance(int((b["gait"] != 0).sum(1).max()))
        m.set_min_stance(int((b["gait"] != 0).sum(1).min()))
    m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
    m.set_order_hint(1 if hint else 0)
    return m


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


def static(name, b, steps):
    B = int(b["batch"])
    res = {"workload": name, "batch": B}
    outs = {}
    for hint in (False, True):
        m = handle(b, B, hint)
        d = m.upload(b)
        o = m.alloc_outputs(B, full=True, iters=True)
        inp, out = m.make_args(d, o)
        ms = timed(m, B, inp, out, steps)
        res["ms_hint" if hint else "ms_plain"] = ms
        res["qps_hint" if hint else "qps_plain"] = B / ms * 1e3
        outs[hint] = (o["soln"].cpu().numpy().copy(), o["iters"].cpu().numpy().copy(), o["status"].cpu().numpy().copy())
        m.close()
    res["bit_identical"] = bool((outs[False][0] == outs[True][0]).all() and (outs[False][1] == outs[True][1]).all())
    res["failed"] = int(((outs[True][2] & 47) != 0).sum())
    res["iters_mean"], res["iters_max"] = float(outs[True][1].mean()), int(outs[True][1].max())
    res["gain"] = res["ms_plain"] / res["ms_hint"] - 1.0
    return res


def rollout(gait, horizon, B, cycles, kick=1.0):
    ro = W.Rollout(B, horizon, gait, seed=3, kick=kick)
    b = ro.record()
    plain, hinted = handle(b, B, False, stance=False), handle(b, B, True, stance=False)
    op, oh = plain.alloc_outputs(B, full=True, iters=True), hinted.alloc_outputs(B, full=True, iters=True)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    rows, prev_it = [], None
    same = True
    for c in range(cycles):
        b = ro.record()
        d = plain.upload(b)
        ip, outp = plain.make_args(d, op)
        ih, outh = hinted.make_args(d, oh)
        plain.solve_async(B, ip, outp)   # untimed: same clocks / caches for both timed solves (the plain handle keeps no hint)
        torch.cuda.synchronize()
        e[0].record(); plain.solve_async(B, ip, outp); e[1].record()
        e[2].record(); hinted.solve_async(B, ih, outh); e[3].record()   # ONE call per cycle on this handle: its hint is the previous CYCLE's
        torch.cuda.synchronize()
        itp = op["iters"].cpu().numpy()
        same = same and bool((op["soln"].cpu().numpy() == oh["soln"].cpu().numpy()).all()) and bool((itp == oh["iters"].cpu().numpy()).all())
        corr = float(np.corrcoef(prev_it, itp)[0, 1]) if prev_it is not None and itp.std() > 0 and prev_it.std() > 0 else None
        rows.append(dict(cycle=c, plain_ms=e[0].elapsed_time(e[1]), hint_ms=e[2].elapsed_time(e[3]), iters=float(itp.mean()),
                         iters_max=int(itp.max()), corr_prev=corr))
        prev_it = itp.copy()
        ro.advance(op["grf"].cpu().numpy())
    plain.close(); hinted.close()
    st = rows[2:]
    agg = lambda k: float(np.mean([r[k] for r in st if r[k] is not None]))
    return {"scenario": f"rollout {gait} h={horizon}, pushes x{kick}", "batch": B, "cycles": cycles, "plain_ms": agg("plain_ms"),
            "hint_ms": agg("hint_ms"), "gain": agg("plain_ms") / agg("hint_ms") - 1.0, "iters_mean": agg("iters"),
            "iters_max": max(r["iters_max"] for r in st), "corr_with_previous_cycle": agg("corr_prev"), "bit_identical": same}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=24)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--static-only", action="store_true")
    ap.add_argument("--single-round", action="store_true", help="one-round launches of other workloads (the priority side of the hint)")
    a = ap.parse_args()
    out = {"static": [], "rollout": []}
    jobs = [("configs[2] mixed gaits 4096", W.make_config(2), 60), ("configs[3] trot h16, 4096 per GPU", W.make_config(3, batch=4096), 30),
            ("configs[4] random contacts, 8192 per GPU", W.make_config(4, batch=8192), 30)]
    jobs += [("configs[1] trot, batch 1024", W.make_config(1), 200)]
    if not a.quick:
        jobs += [("trot h10, batch 16384", W.make_config(1, batch=16384), 30),
                 ("configs[2] at 8192", W.make_config(2, batch=8192), 40)]
    if a.single_round:
        jobs = [("configs[1] trot, batch 1024", W.make_config(1), 200), ("mixed gaits, batch 1024", W.make_config(2, batch=1024), 100),
                ("mixed gaits, batch 1280 (handle of 4096)", W.make_config(2, batch=1280), 100),
                ("random contacts, batch 1024", W.make_config(4, batch=1024), 60), ("trot h16, batch 512", W.make_config(3, batch=512), 60),
                ("trot, batch 512", W.make_config(1, batch=512), 200), ("trot, batch 256", W.make_config(1, batch=256), 200),
                ("standing h10, batch 256 (one-kernel path)", W.make_standing(256, 10), 40),
                ("standing h10 calm, batch 256", W.make_standing(256, 10, calm=True), 40)]
    for name, b, steps in jobs:
        r = static(name, b, steps)
        out["static"].append(r)
        print(f"# {name:44s} {r['qps_plain']:.3e} -> {r['qps_hint']:.3e} QP/s ({100 * r['gain']:+.1f} %)  iters {r['iters_mean']:.2f}/{r['iters_max']}  "
              f"bit-identical {r['bit_identical']}  failed {r['failed']}", file=sys.stderr)
    ros = [("mixed", 10, 4096), ("trot", 16, 4096), ("trot", 10, 1024), ("mixed", 10, 1024)] + ([] if a.quick else [("mixed", 10, 8192), ("trot", 10, 8192), ("trot", 16, 512)])
    for gait, h, B in ([] if a.static_only else ros):
        r = rollout(gait, h, B, a.cycles)
        out["rollout"].append(r)
        print(f"# {r['scenario']:36s} B={B}: {r['plain_ms']:.4f} -> {r['hint_ms']:.4f} ms/cycle ({100 * r['gain']:+.1f} %)  iters {r['iters_mean']:.2f}/{r['iters_max']}  "
              f"corr(prev cycle) {r['corr_with_previous_cycle']:.2f}  bit-identical {r['bit_identical']}", file=sys.stderr)
    print(json.dumps({"order_hint": out}, indent=1))


if __name__ == "__main__":
    main()
