import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from quadruped_ctrl_amd import workloads as W
from tools.size_order_ab import run
for name, b in (("cfg1", W.make_config(1)), ("cfg2_1024", W.make_config(2, batch=1024)), ("cfg3_4096", W.make_config(3, batch=4096)), ("cfg1_8192", W.make_config(1, batch=8192))):
    ms, r = run(b, 0, 5)
    np.savez(os.path.join(ROOT, "gpurun_out", "iters_%s.npz" % name), iters=r[1])
