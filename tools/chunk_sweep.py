"""Time per call of the decoupled path against the number of chunks (qmpc_set_chunks).  GPU box: python tools/chunk_sweep.py"""
import os, sys, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
for name, b in (("standing_h10", W.make_standing(1024, 10)), ("standing_h10_4096", W.make_standing(4096, 10)), ("standing_h14", W.make_standing(1024, 14)),
                ("config4_8192", W.shard(W.make_config(4, batch=65536), 0, 8))):
    row = {}
    for nch in (1, 2, 3, 4, 8):
        m = BatchedConvexMPC(0, max_batch=b["batch"], max_horizon=16)
        m.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
        nst = (b["gait"] != 0).sum(1)
        m.set_max_stance(int(nst.max())); m.set_min_stance(int(nst.min()))
        m.set_chunks(nch)
        d = m.upload(b); o = m.alloc_outputs(b["batch"], full=False, iters=True); inp, outp = m.make_args(d, o)
        s = torch.cuda.current_stream(0)
        for _ in range(10): m.solve_async(b["batch"], inp, outp, s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(40): m.solve_async(b["batch"], inp, outp, s)
        torch.cuda.synchronize(); row[nch] = round((time.perf_counter() - t0) / 40 * 1e3, 4)
        m.close()
    print(name, json.dumps(row), flush=True)
