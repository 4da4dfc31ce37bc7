#!/usr/bin/env python
"""Handles of 1100 ... 2047 robots whose every robot fits the 64-row class: the four-per-CU instantiation (default below 2048)
against the five-per-CU one (qmpc_set_dense 2), size order on, no hint."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from tools.order_hint import handle, timed
out = []
for cfg in (1, 2):
    for B in (1100, 1280, 1400, 1600, 1800, 2000):
        b = W.make_config(cfg, batch=B)
        r = {"cfg": cfg, "batch": B}
        for dense in (1, 2, 1, 2):
            m = handle(b, B, False)
            m.set_dense(dense)
            d = m.upload(b); o = m.alloc_outputs(B, full=False, iters=True); inp, outp = m.make_args(d, o)
            ms = timed(m, B, inp, outp, 60)
            r.setdefault("qps_dense%d" % dense, []).append(B / ms * 1e3)
            m.close()
        r["gain_dense2"] = max(r["qps_dense2"]) / max(r["qps_dense1"]) - 1
        out.append(r)
        print(cfg, B, "4/CU %.3e  5/CU %.3e  %+.1f %%" % (max(r["qps_dense1"]), max(r["qps_dense2"]), 100 * r["gain_dense2"]), file=sys.stderr)
print(json.dumps(out, indent=1))
