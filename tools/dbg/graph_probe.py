"""Scratch: which part of a captured solve faults?  usage: python tools/dbg/graph_probe.py <case>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
case = sys.argv[1]
b = W.make_config(4, batch=768) if "cfg4" in case else W.make_config(1, batch=512)
B, h = b["batch"], b["horizon"]
m = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
if "hint" in case:
    m.set_max_stance(20)
m.setup(b["dt"], h, b["mu"], b["f_max"])
d = m.upload(b); o = m.alloc_outputs(B, full=True); inp, out = m.make_args(d, o)
s = torch.cuda.Stream()
m.solve_async(B, inp, out, stream=s)
s.synchronize()
eager = o["soln"].clone()
print(case, "eager ok", int(o["status"].max()), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s):
    m.solve_async(B, inp, out, stream=s)
print("captured", flush=True)
import ctypes as C
def cmp(tag):
    torch.cuda.synchronize()
    buf = (C.c_int * 768)()
    m.lib.qmpc_debug_read_counts(m.h, buf)
    a = np.array(buf[:]).reshape(3, 256)
    print("   counters nonzero:", {f"s{si}[{k}]": int(a[si, k]) for si in range(3) for k in np.nonzero(a[si])[0][:12]})
    df = (o["soln"] - eager).abs().max(1).values
    st = o["status"].cpu().numpy()
    print(tag, "equal", torch.equal(o["soln"], eager), "robots differing", int((df > 0).sum()), "max diff", float(df.max()),
          "status uniq", np.unique(st)[:6], "soln absmax", float(o["soln"].abs().max()), flush=True)
for rep in range(5):
    o["soln"].zero_(); o["status"].fill_(-1)
    g.replay()
    cmp(f"replay {rep}")
    if rep == 1:
        o["soln"].zero_(); o["status"].fill_(-1)
        if "sstream" in case:
            m.solve_async(B, inp, out, stream=s); s.synchronize()
        else:
            m.solve_async(B, inp, out)
        cmp("eager between")
