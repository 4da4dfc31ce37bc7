#!/usr/bin/env python
"""qmpc_set_size_order on / off, plain order (no hint), bit-identity and rates.
    python tools/size_order_ab.py > gpurun_out/size_order_ab.json"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from quadruped_ctrl_amd import workloads as W  # noqa: E402
from tools.order_hint import handle, timed  # noqa: E402


def run(b, so, steps, hints=True):
    B = int(b["batch"])
    m = handle(b, B, False, stance=hints)
    m.set_size_order(so)
    d = m.upload(b)
    o = m.alloc_outputs(B, full=True, iters=True)
    inp, out = m.make_args(d, o)
    ms = timed(m, B, inp, out, steps)
    r = (o["soln"].cpu().numpy().copy(), o["iters"].cpu().numpy().copy(), o["status"].cpu().numpy().copy(), o["grf"].cpu().numpy().copy())
    m.close()
    return ms, r


def ab(name, b, steps, hints=True):
    B = int(b["batch"])
    res = {"workload": name, "batch": B, "stance_hints": hints}
    ms = {}
    outs = {}
    for so in (0, 1, 0, 1):
        t, r = run(b, so, steps, hints)
        ms.setdefault(so, []).append(t)
        outs[so] = r
    res["ms_off"], res["ms_on"] = ms[0], ms[1]
    res["qps_off"] = B / min(ms[0]) * 1e3
    res["qps_on"] = B / min(ms[1]) * 1e3
    res["gain"] = min(ms[0]) / min(ms[1]) - 1.0
    res["bit_identical"] = bool(all((outs[0][k] == outs[1][k]).all() for k in range(4)))
    res["failed"] = int(((outs[1][2] & 47) != 0).sum())
    return res


if __name__ == "__main__":
    out = []
    for pct in ([int(a) for a in sys.argv[1:]] or [50]):
        os.environ["QMPC_SO_FIRST_PCT"] = str(pct)
        for name, b, steps, hints in (("cfg2", W.make_config(2), 40, True), ("cfg4", W.make_config(4, batch=8192), 20, True),
                                      ("cfg2_8192", W.make_config(2, batch=8192), 20, True), ("cfg2_2048", W.make_config(2, batch=2048), 40, True),
                                      ("cfg2_16384", W.make_config(2, batch=16384), 10, True), ("cfg4_16384", W.make_config(4, batch=16384), 10, True),
                                      ("cfg4_nohints", W.make_config(4, batch=8192), 20, False), ("cfg3_nohints", W.make_config(3, batch=4096), 10, False),
                                      ("cfg1_8192_nohints", W.make_config(1, batch=8192), 20, False),
                                      ("cfg3", W.make_config(3, batch=4096), 10, True), ("cfg1_8192", W.make_config(1, batch=8192), 20, True),
                                      ("cfg1_4096", W.make_config(1, batch=4096), 30, True), ("cfg1_16384", W.make_config(1, batch=16384), 10, True)):
            r = ab(name, b, steps, hints)
            r["so_first_pct"] = pct
            out.append(r)
    print(json.dumps(out, indent=1))
