#!/usr/bin/env python
"""Host emulation of the second proxy: base (coasting tracking error x early stance) + 0.15 x the support-asymmetry term
(first support phase: length x friction saturation needed to balance gravity's moment, beyond 1.5)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_study import permute, interleave
from tools.size_order_ab import run
from tools.proxy_order_study import feats


def score2(b, beta=0.15):
    B = int(b['batch']); h = int(b['horizon'])
    werr, nst, first3 = feats(b)
    base = werr / b['weights'][:, :6].sum(1) * first3
    g = (b['gait'].reshape(B, h, 4) != 0).astype(float)
    r = b['r'].reshape(B, 3, 4); n = g.sum(2); nn = np.where(n > 0, n, 1)
    cx = (g * r[:, 0:1, :]).sum(2) / nn; cy = (g * r[:, 1:2, :]).sum(2) / nn
    hz = np.abs((g * r[:, 2:3, :]).sum(2) / nn)
    sat = np.sqrt(cx * cx + cy * cy) / (np.maximum(hz, 1e-3) * float(b['mu']))
    pat = np.zeros(B)
    for i in range(B):
        k = 0
        while k < h and n[i, k] == 0: k += 1
        if k == h: continue
        k0 = k
        while k < h and (g[i, k] == g[i, k0]).all(): k += 1
        s1 = sat[i, k0]
        sa = min(s1, 1.0) if s1 > 0.6 else 0.0
        pat[i] = max(0.0, (k - k0) * sa - 1.5)
    return base + beta * pat, nst


if __name__ == "__main__":
    out = []
    for name, b, steps, maxfit in (("cfg2", W.make_config(2), 40, 21), ("cfg2_8192", W.make_config(2, batch=8192), 20, 21), ("cfg4", W.make_config(4, batch=8192), 20, 21),
                                   ("cfg1_8192", W.make_config(1, batch=8192), 20, 21), ("cfg3", W.make_config(3, batch=4096), 10, 32), ("cfg2_16384", W.make_config(2, batch=16384), 10, 21)):
        B = int(b["batch"])
        sc, nst = score2(b)
        fit = nst <= maxfit
        r = {"workload": name}
        def go(fo, po=None):
            perm = interleave(fo, po if po is not None else np.nonzero(~fit)[0]) if (~fit).any() else fo
            bp = permute(b, perm)
            return B / min(run(bp, 0, steps)[0], run(bp, 0, steps)[0]) * 1e3
        po = np.argsort(np.where(~fit, -sc, np.inf), kind="stable")[:(~fit).sum()]
        r["plain"] = B / min(run(b, 0, steps)[0], run(b, 0, steps)[0]) * 1e3
        r["size(+passes by score)"] = go(np.argsort(np.where(fit, -nst, np.inf), kind="stable")[:fit.sum()], po)
        r["score(+passes by score)"] = go(np.argsort(np.where(fit, -sc, np.inf), kind="stable")[:fit.sum()], po)
        r["size then score"] = go(np.lexsort((-sc, np.where(fit, -nst, np.inf)))[:fit.sum()], po)
        out.append(r)
        print(name, {k: (round(v / 1e7, 3) if k != "workload" else v) for k, v in r.items()}, file=sys.stderr)
    print(json.dumps(out, indent=1))
