#!/usr/bin/env python
"""Mixed-size batches: size first, tracking-error proxy second (host emulation)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_study import permute, interleave
from tools.size_order_ab import run
from tools.proxy_order_study import feats
out = []
for name, b, steps, maxfit in (("cfg2", W.make_config(2), 40, 21), ("cfg2_8192", W.make_config(2, batch=8192), 20, 21), ("cfg4", W.make_config(4, batch=8192), 20, 21)):
    B = int(b["batch"])
    werr, nst, first3 = feats(b)
    fit = nst <= maxfit
    sc = werr * first3
    po = np.argsort(np.where(~fit, -sc, np.inf), kind="stable")[:(~fit).sum()]
    r = {"workload": name}
    def go(fo):
        perm = interleave(fo, po) if (~fit).any() else fo
        bp = permute(b, perm)
        return B / min(run(bp, 0, steps)[0], run(bp, 0, steps)[0]) * 1e3
    r["size"] = go(np.argsort(np.where(fit, -nst, np.inf), kind="stable")[:fit.sum()])
    r["size_then_proxy_desc"] = go(np.lexsort((-sc, np.where(fit, -nst, np.inf)))[:fit.sum()])
    r["size_then_proxy_asc"] = go(np.lexsort((sc, np.where(fit, -nst, np.inf)))[:fit.sum()])
    # coarse size (pairs of stance counts) then proxy
    r["size2_then_proxy_desc"] = go(np.lexsort((-sc, np.where(fit, -(nst // 2), np.inf)))[:fit.sum()])
    r["size4_then_proxy_desc"] = go(np.lexsort((-sc, np.where(fit, -(nst // 4), np.inf)))[:fit.sum()])
    out.append(r)
    print(name, {k: (round(v / 1e7, 3) if k != "workload" else v) for k, v in r.items()}, file=sys.stderr)
print(json.dumps(out, indent=1))
