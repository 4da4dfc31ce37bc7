"""Scratch profiling aid: per-phase shader-clock breakdown of the solve kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from quadruped_ctrl_amd import workloads as W
from quadruped_ctrl_amd.binding import BatchedConvexMPC
cfg = sys.argv[1] if len(sys.argv) > 1 else "1"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
# "1".."4" = BASELINE configs; "s10" / "s14" / "s16" = standing at that horizon; "c10".. = calm standing
b = (W.make_standing(B, int(cfg[1:]), calm=(cfg[0] == "c")) if cfg[0] in "sc" else W.make_config(int(cfg), batch=B))
mpc = BatchedConvexMPC(0, max_batch=B, max_horizon=16)
if not os.environ.get("QMPC_PHASE_NO_HINT"):   # (the caller's stance bounds, as bench.py gives them: they select the instantiation)
    mpc.set_max_stance(int((b["gait"] != 0).sum(1).max()))
    mpc.set_min_stance(int((b["gait"] != 0).sum(1).min()))
mpc.setup(b["dt"], b["horizon"], b["mu"], b["f_max"])
mpc.set_order_hint(0)
d = mpc.upload(b); o = mpc.alloc_outputs(B); inp, out = mpc.make_args(d, o)
for _ in range(3): mpc.solve_async(B, inp, out)
torch.cuda.synchronize()
clk = mpc.debug_clock(B)
mpc.solve_async(B, inp, out); torch.cuda.synchronize()
c = clk.cpu().numpy().astype(np.float64)
it = o["iters"].cpu().numpy()
if os.environ.get("QMPC_PHASE_SWEEP_ONLY"):
    # decoupled path: the engine kernel stamps the same rows from iteration 20 on; keep the robots it left alone
    # (phases 0..4 are the sweep kernel's; 5 and 6 are meaningless here)
    keep = it < 20
    c, it = c[keep], it[keep]
    print(f"(sweep kernel's stamps of the {keep.sum()} robots with fewer than 20 iterations)")
names = ["0 inputs", "1 E/s", "2 asm H,g", "3 sweep", "4 x_u", "5 active set", "6 out"]
d = np.diff(c[:, :8], axis=1)
print(f"cfg{cfg} B={B}: per-phase shader cycles (median / mean / max over blocks); iters mean {it.mean():.2f}")
for k, nm in enumerate(names):
    print(f"  {nm:14s} {np.median(d[:,k]):9.0f} {d[:,k].mean():9.0f} {d[:,k].max():9.0f}")
tot = c[:, 7] - c[:, 0]
print(f"  total          {np.median(tot):9.0f} {tot.mean():9.0f} {tot.max():9.0f}")
sel = it > 0
if sel.any():
    print(f"  active-set cycles per iteration (blocks with it>0): {np.median(d[sel,5]/it[sel]):.0f}")

ok = (it > 0) & (c[:, 10] > 0)
if ok.any():
    print("  first AS iteration (median cycles): select %d | z, r, delta %d | ratio test, step, event write %d" % (
        np.median(c[ok, 8] - c[ok, 5]), np.median(c[ok, 9] - c[ok, 8]), np.median(c[ok, 10] - c[ok, 9])))
order = np.argsort(c[:, 0])
early, late = order[: len(order) // 2], order[len(order) // 2:]
print("  blocks by start time: early-half median total %.0f (sweep %.0f) | late-half median total %.0f (sweep %.0f)" % (
    np.median(tot[early]), np.median(d[early, 3]), np.median(tot[late]), np.median(d[late, 3])))
qs = [50, 90, 99, 99.9, 100]
print("  iters percentiles", {q: int(np.percentile(it, q)) for q in qs}, " AS cycles by iters:",
      {int(k): int(np.median(d[it == k, 5])) for k in np.unique(it)[:16]})

print("  stage 0 detail (tid 0): loads landed %d | compute %d | barrier %d ;  stage 2: g %d | H asm %d" % tuple(
    np.median(x) for x in (c[:,11]-c[:,0], c[:,12]-c[:,11], c[:,1]-c[:,12], c[:,13]-c[:,2], c[:,3]-c[:,13])))


fixed = tot - d[:, 5]
print("  fixed part (total - active set) percentiles", {q: int(np.percentile(fixed, q)) for q in [1, 10, 50, 90, 99, 100]},
      "| sweep percentiles", {q: int(np.percentile(d[:, 3], q)) for q in [1, 10, 50, 90, 99, 100]})
slow = np.argsort(tot)[-5:]
print("  five slowest blocks: total", tot[slow].astype(int).tolist(), "iters", it[slow].tolist(), "sweep", d[slow, 3].astype(int).tolist(),
      "fixed", fixed[slow].astype(int).tolist())
