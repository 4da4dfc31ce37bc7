#!/usr/bin/env python
"""Uniform-size batches (every robot fits the first class, same contact-table size): does the tracking-error proxy order help?
Host emulation, size order and hint off."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_study import permute
from tools.size_order_ab import run
from tools.proxy_order_study import feats
out = []
for name, b, steps in (("cfg3_4096", W.make_config(3, batch=4096), 10), ("cfg1_8192", W.make_config(1, batch=8192), 20), ("cfg1_4096", W.make_config(1, batch=4096), 30),
                       ("cfg1_16384", W.make_config(1, batch=16384), 10)):
    B = int(b["batch"])
    werr, nst, first3 = feats(b)
    r = {"workload": name}
    for k, perm in (("plain", np.arange(B)), ("werr_desc", np.argsort(-werr, kind="stable")), ("werr_x_first3_desc", np.argsort(-werr * first3, kind="stable")),
                    ("werr_asc", np.argsort(werr, kind="stable"))):
        bp = permute(b, perm)
        r[k] = B / min(run(bp, 0, steps)[0], run(bp, 0, steps)[0]) * 1e3
    out.append(r)
    print(name, {k: (round(v / 1e7, 3) if k != "workload" else v) for k, v in r.items()}, file=sys.stderr)
print(json.dumps(out, indent=1))
