import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
B=1024
b=W.make_config(1,batch=B)
mpc=BatchedConvexMPC(0,max_batch=B,max_horizon=16); mpc.set_max_stance(20); mpc.set_min_stance(20)
mpc.setup(b["dt"],b["horizon"],b["mu"],b["f_max"])
d=mpc.upload(b); o=mpc.alloc_outputs(B); inp,out=mpc.make_args(d,o)
for _ in range(5): mpc.solve_async(B,inp,out)
torch.cuda.synchronize()
clk=mpc.debug_clock(B)
mpc.solve_async(B,inp,out); torch.cuda.synchronize()
c=clk.cpu().numpy().astype(np.float64); it=o["iters"].cpu().numpy()
# (the shader clock is per XCD: workgroup i runs on XCD i % 8; times relative to the XCD's first start)
x=np.arange(B)%8
base=np.array([c[x==k,0].min() for k in range(8)])[x]
s=c[:,0]-base; e=c[:,7]-base
print("start spread: pct", {q:int(np.percentile(s,q)) for q in (0,10,50,90,99,100)})
print("end: pct", {q:int(np.percentile(e,q)) for q in (50,90,99,100)})
hard=np.argsort(-it)[:8]
print("hard robots: idx", hard.tolist(), "iters", it[hard].tolist(), "start", s[hard].astype(int).tolist(), "end", e[hard].astype(int).tolist())
order=np.argsort(s); print("start vs blockIdx corr", np.corrcoef(np.arange(B), s)[0,1])
