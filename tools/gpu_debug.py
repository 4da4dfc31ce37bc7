"""Scratch GPU diagnostic (not a test): stage-wise comparison of the HIP path
against the oracle models.  Run on the GPU box via gpurun."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
from oracle import oracle as O, kron_model as K


def check(name, b, nfull=8):
    B, h = b["batch"], b["horizon"]
    mpc = BatchedConvexMPC(0, max_batch=max(B, 64), max_horizon=16)
    mpc.setup(b["dt"], h, b["mu"], b["f_max"])
    Hd, gd, ld = mpc.debug_dump(B)
    t = time.time()
    res = mpc.solve(b, full=True)
    dt = time.time() - t
    Hd = Hd.cpu().numpy(); gd = gd.cpu().numpy()
    mpc.debug_off()
    ref, nwsr, rc = O.solve_batch(b)
    num = np.abs(res["grf"].astype(np.float64) - ref[:, :12]).max(1)
    den = np.maximum(np.abs(ref[:, :12]).max(1), 1.0)
    e2e = num / den
    numx = np.abs(res["soln"] - ref).max(1) / np.maximum(np.abs(ref).max(1), 1.0)
    print(f"[{name}] B={B} h={h} status!=0: {(res['status']!=0).sum()} (bits {np.unique(res['status'])}) "
          f"iters mean {res['iters'].mean():.2f} max {res['iters'].max()} | e2e f0 rel max {e2e.max():.3e} "
          f"median {np.median(e2e):.3e} | full-x rel max {numx.max():.3e} | wall {dt*1e3:.1f} ms", flush=True)
    # stage-wise on a few instances
    eH = []; eg = []; ex = []
    for i in range(min(nfull, B)):
        Hk, gk = K.assemble(b, i)
        stance = [k for k in range(4 * h) if b["gait"][i][k]]
        vi = np.array([3 * k + a for k in stance for a in range(3)], int)
        n = vi.size
        if n == 0: continue
        Hg = Hd[i][:n, :n]; gg = gd[i][:n]
        eH.append(np.abs(Hg - Hk[np.ix_(vi, vi)]).max() / np.abs(Hk).max())
        eg.append(np.abs(gg - gk[vi]).max() / np.abs(gk).max())
        # solver parity: real qpOASES on the GPU's own H,g
        Hf, gf, A, lb, ub, x0 = O.assemble(b, i)
        Hfull = np.zeros_like(Hf); gfull = np.zeros_like(gf)
        Hfull[np.ix_(vi, vi)] = Hg; gfull[vi] = gg
        ve, Hr, gr, Ar, lr, ur = O.reduce(Hf, gf, A, lb, ub)
        xq, y, used, r1, r2 = O.qpoases(Hg, gg, Ar, lr, ur)
        ex.append(np.abs(res["soln"][i][vi] - xq).max() / max(np.abs(xq).max(), 1))
    if eH:
        print(f"   stage: H vs fp64 model {max(eH):.2e}  g {max(eg):.2e}  x vs qpOASES(same H,g) {max(ex):.2e}", flush=True)
    bad = np.argsort(-e2e)[:3]
    for i in bad:
        print(f"   worst i={i} err {e2e[i]:.2e} st {res['status'][i]} it {res['iters'][i]} nwsr {nwsr[i]} gpu {res['grf'][i][:6].round(3)} ref {ref[i][:6].round(3)}")
    mpc.close()


if __name__ == "__main__":
    which = sys.argv[1:] or ["c1", "c2", "c4", "stand", "c3", "drag"]
    if "c1" in which: check("cfg1 trot", W.make_config(1, batch=256))
    if "c2" in which: check("cfg2 mixed", W.make_config(2, batch=256))
    if "c4" in which: check("cfg4 random+stairs", W.make_config(4, batch=512))
    if "stand" in which: check("standing h10 (n=120)", W.make_standing(64))
    if "c3" in which: check("cfg3 trot h16 (n=96)", W.make_config(3, batch=64))
    if "stand16" in which: check("standing h16 (n=192)", W.make_standing(16, horizon=16))
    if "drag" in which:
        b = W.make_config(1, batch=64); b["x_drag"][:] = 0.37
        check("cfg1 + x_drag", b)
