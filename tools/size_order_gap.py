#!/usr/bin/env python
"""Where the size order's gain goes: physically sorted data (plain order == the ideal size order) with the library's
size order off / on, against the original data with it off / on."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_study import permute, interleave
from tools.size_order_ab import run

out = []
for name, b, steps in (("cfg2_8192", W.make_config(2, batch=8192), 20), ("cfg4", W.make_config(4, batch=8192), 20), ("cfg2", W.make_config(2), 40)):
    B = int(b["batch"])
    nst = (b["gait"].reshape(B, -1) != 0).sum(1)
    fit = 3 * nst <= 64
    perm = interleave(np.argsort(np.where(fit, -nst, 1000), kind="stable")[:fit.sum()], np.nonzero(~fit)[0]) if (~fit).any() else np.argsort(-nst, kind="stable")
    bs = permute(b, perm)
    r = {"workload": name}
    for tag, data in (("orig", b), ("sorted", bs)):
        for so in (0, 1):
            r["%s_so%d" % (tag, so)] = B / min(run(data, so, steps)[0], run(data, so, steps)[0]) * 1e3
    out.append(r)
print(json.dumps(out, indent=1))
