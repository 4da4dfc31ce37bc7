#!/usr/bin/env python
"""The BASELINE workloads re-drawn with other random seeds (same distributions, other robots): size order / proxy staging off / on,
and the score's correlation with the iteration count.  The proxy's constants were fitted on the default seed's draws."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_ab import ab, run
from tools.proxy2_study import score2
out = []
for seed in (777, 4242):
    W.SEED0 = seed
    for name, b, steps in (("cfg1", W.make_config(1), 200), ("cfg2", W.make_config(2), 40), ("cfg3", W.make_config(3, batch=4096), 10), ("cfg4", W.make_config(4, batch=8192), 20)):
        r = ab(name, b, steps)
        it = run(b, 1, 2)[1][1]
        sc, nst = score2(b)
        r.update(seed=seed, corr_score_iters=float(np.corrcoef(sc, it)[0, 1]), iters_max=int(it.max()))
        out.append(r)
        print(seed, name, "off %.3e on %.3e gain %+.1f %% corr %.3f ident %s" % (r["qps_off"], r["qps_on"], 100 * r["gain"], r["corr_score_iters"], r["bit_identical"]), file=sys.stderr)
print(json.dumps(out, indent=1))
