#!/usr/bin/env python
"""Does a STATIC cost proxy (reduced size n_r = 3 x stance foot-steps, known from the contact table before anything is
computed) recover part of what the previous-cycle order hint gives in plain launch order?  Host-side emulation: the
batch is permuted so that plain launch order IS the proxy order; nothing in the library changes.

    python tools/size_order_study.py > gpurun_out/size_order_study.json
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
from tools.order_hint import handle, timed  # noqa: E402


def permute(b, perm):
    B = int(b["batch"])
    o = {}
    for k, v in b.items():
        o[k] = v[perm].copy() if isinstance(v, np.ndarray) and v.ndim >= 1 and v.shape[0] == B else v
    return o


def run(b, hint, steps):
    B = int(b["batch"])
    m = handle(b, B, hint)
    d = m.upload(b)
    o = m.alloc_outputs(B, full=True, iters=True)
    inp, out = m.make_args(d, o)
    ms = timed(m, B, inp, out, steps)
    it = o["iters"].cpu().numpy().copy()
    so = o["soln"].cpu().numpy().copy()
    m.close()
    return ms, it, so


def interleave(a, b):
    """merge two index lists proportionally (b spread evenly through a)"""
    n = len(a) + len(b)
    if len(b) == 0:
        return np.asarray(a)
    pos_b = set((np.arange(len(b)) * n // len(b)).tolist())
    out, ia, ib = [], 0, 0
    for k in range(n):
        if (k in pos_b and ib < len(b)) or ia >= len(a):
            out.append(b[ib]); ib += 1
        else:
            out.append(a[ia]); ia += 1
    return np.array(out)


def study(name, b, steps, dump):
    B = int(b["batch"])
    h = int(b["horizon"])
    nst = (b["gait"].reshape(B, -1) != 0).sum(1)
    res = {"workload": name, "batch": B}
    ms0, it0, so0 = run(b, False, steps)
    msh, _, _ = run(b, True, steps)
    res["ms_plain"], res["ms_exact_hint"] = ms0, msh
    res["corr_nr_iters"] = float(np.corrcoef(nst, it0)[0, 1])
    rng = np.random.default_rng(1)
    fit = 3 * nst <= 64
    key_a = np.where(fit, 1000 - nst, 2000 - nst)     # robots of the first class by size (descending), then the ones it hands on (descending)
    key_b = np.where(fit, 1000 - nst, 2000 + nst)     # ... the ones it hands on ascending
    key_c = np.where(fit, 0, 1)                       # hand-overs last, nothing else changed
    key_d = np.where(fit, 1, 0)                       # hand-overs first, nothing else changed
    orders = {
        "interleaved_desc": interleave(np.argsort(np.where(fit, -nst, 1000), kind="stable")[:fit.sum()],
                                       np.argsort(np.where(~fit, -nst, 1000), kind="stable")[:(~fit).sum()]),
        "interleaved_fitdesc_passorig": interleave(np.argsort(np.where(fit, -nst, 1000), kind="stable")[:fit.sum()], np.nonzero(~fit)[0]),
        "fit_desc_then_pass_desc": np.argsort(key_a, kind="stable"),
        "fit_desc_then_pass_asc": np.argsort(key_b, kind="stable"),
        "pass_last": np.argsort(key_c, kind="stable"),
        "pass_first": np.argsort(key_d, kind="stable"),
        "size_desc": np.argsort(-nst, kind="stable"),
        "size_asc": np.argsort(nst, kind="stable"),
        "iters_desc_exact": np.argsort(-it0, kind="stable"),
        "cost_desc_exact": np.argsort(-(nst * 3 * 450 + it0 * 3200), kind="stable"),
        "random": rng.permutation(B),
    }
    for k, perm in orders.items():
        bp = permute(b, perm)
        ms, it, so = run(bp, False, steps)
        res["ms_" + k] = ms
        res["same_" + k] = bool((so == so0[perm]).all() and (it == it0[perm]).all())
    for k in list(res):
        if k.startswith("ms_"):
            res["qps_" + k[3:]] = B / res[k] * 1e3
    if dump:
        np.savez(os.path.join(dump, "size_order_%s.npz" % name), nst=nst, iters=it0)
    return res


def one(name, order):
    """one order only, many calls: for rocprofv3 --kernel-trace --stats"""
    b = W.make_config(4, batch=8192) if name == "cfg4" else W.make_config(2)
    B = int(b["batch"])
    nst = (b["gait"].reshape(B, -1) != 0).sum(1)
    fit = 3 * nst <= 64
    keys = {"plain": np.arange(B), "size_desc": -nst, "fit_desc_then_pass_desc": np.where(fit, 1000 - nst, 2000 - nst),
            "pass_last": np.where(fit, 0, 1), "pass_first": np.where(fit, 1, 0)}
    if order == "interleaved_desc":
        perm = interleave(np.argsort(np.where(fit, -nst, 1000), kind="stable")[:fit.sum()],
                          np.argsort(np.where(~fit, -nst, 1000), kind="stable")[:(~fit).sum()])
    else:
        perm = np.argsort(keys[order], kind="stable")
    print(order, run(permute(b, perm), False, 20)[0])


if __name__ == "__main__":
    if len(sys.argv) > 2:
        one(sys.argv[1], sys.argv[2])
        sys.exit(0)
    dump = os.path.join(ROOT, "gpurun_out")
    os.makedirs(dump, exist_ok=True)
    out = []
    out.append(study("cfg2", W.make_config(2), 40, dump))
    out.append(study("cfg4", W.make_config(4, batch=8192), 20, dump))
    out.append(study("cfg2_8192", W.make_config(2, batch=8192), 20, dump))
    print(json.dumps(out, indent=1))
