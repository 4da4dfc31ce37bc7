#!/usr/bin/env python
"""One-round launches, no hint: the sweep's issue priority staged by the tracking-error proxy's rank (qmpc_set_size_order on) or
not at all (off).  QMPC_PRIO_HARD_DIV / QMPC_PRIO_MID_DIV from the command line: pairs 'hard,mid'."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_ab import run

out = []
jobs = (("cfg1", W.make_config(1), 200), ("cfg2_1024", W.make_config(2, batch=1024), 100), ("cfg4_1024", W.make_config(4, batch=1024), 60),
        ("cfg1_768", W.make_config(1, batch=768), 200), ("cfg3_512", W.make_config(3, batch=512), 60), ("cfg1_1280_h4096", W.make_config(1, batch=1280), 100))
for pair in (sys.argv[1:] or ["0,0"]):
    hd, md = pair.split(",")
    os.environ["QMPC_PRIO_HARD_DIV"], os.environ["QMPC_PRIO_MID_DIV"] = hd, md
    for name, b, steps in jobs:
        B = int(b["batch"])
        ms = {0: [], 1: []}
        outs = {}
        for so in (0, 1, 0, 1, 0, 1):
            t, r = run(b, so, steps)
            ms[so].append(t)
            outs[so] = r
        rec = {"divs": pair, "workload": name, "batch": B, "qps_off": B / min(ms[0]) * 1e3, "qps_on": B / min(ms[1]) * 1e3,
               "bit_identical": bool(all((outs[0][k] == outs[1][k]).all() for k in range(4)))}
        rec["gain"] = rec["qps_on"] / rec["qps_off"] - 1
        out.append(rec)
        print(pair, name, "off %.3e on %.3e gain %+.1f %% ident %s" % (rec["qps_off"], rec["qps_on"], 100 * rec["gain"], rec["bit_identical"]), file=sys.stderr)
print(json.dumps(out, indent=1))
