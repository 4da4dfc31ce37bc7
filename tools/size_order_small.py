#!/usr/bin/env python
"""Batches of 1.1 ... 2.5 rounds: where does the size order start to pay?  QMPC_SO_MIN_DIV from the command line."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_ctrl_amd import workloads as W
from tools.size_order_ab import ab
out = []
for div in (sys.argv[1:] or ["2"]):
    os.environ["QMPC_SO_MIN_DIV"] = div
    for cfg, B in ((2, 2560 + 64), (2, 2304), (2, 2048 + 128), (2, 1400), (2, 1600), (2, 1800), (2, 3072), (4, 2304), (4, 3000), (1, 2304), (1, 1400), (1, 1700)):
        b = W.make_config(cfg, batch=B)
        r = ab("cfg%d_%d" % (cfg, B), b, 40)
        r["min_div"] = div
        out.append(r)
        print(div, r["workload"], "off %.3e on %.3e gain %+.1f %% ident %s" % (r["qps_off"], r["qps_on"], 100 * r["gain"], r["bit_identical"]), file=sys.stderr)
print(json.dumps(out, indent=1))
