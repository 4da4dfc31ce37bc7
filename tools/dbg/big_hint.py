import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
import order_hint as oh
from quadruped_ctrl_amd import workloads as W
for name, b, steps in (("trot 65536", W.make_config(1, batch=65536), 10), ("mixed 65536", W.make_config(2, batch=65536), 10), ("mixed 16384", W.make_config(2, batch=16384), 20), ("random contacts 32768", W.make_config(4, batch=32768), 10)):
    r = oh.static(name, b, steps)
    print(f"# {name:28s} {r['qps_plain']:.3e} -> {r['qps_hint']:.3e} ({100*r['gain']:+.1f} %) bit-identical {r['bit_identical']} failed {r['failed']}")
