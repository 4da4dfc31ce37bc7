#!/usr/bin/env python
"""CPU pre-check of tests/test_gpu_parity.py::test_full_shard_vs_oracle (no GPU needed).

The GPU's assembly IS the fp64 Kronecker model to 1e-10 (test_assembly_vs_fp64_model_*), its solve the exact minimiser
to 1e-12, so `oracle/kron_model.assemble` -> the reference's qpOASES is a stand-in for the GPU answer (up to the float
transcendentals the kernel evaluates, ~1e-7).  For one GPU's shard of every BASELINE config this prints, against the oracle
pipeline (float assembly restatement + qpOASES at nWSR = 100): the error distribution, the robots over north_star's flat
1e-4 and, for those only, the reference's own float evaluation-order spread (oracle/noise_floor.py) and whether
err < max(1e-4, 1.5 x spread).  TEST / ANALYSIS TOOLING: imports oracle/.

    python tools/full_shard_floor.py [cfg ...]   (default 1 2 3 4)
"""
import os
import sys
import time
from multiprocessing import Pool

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import kron_model as K  # noqa: E402
from oracle import noise_floor as NF  # noqa: E402
from oracle import oracle as O  # noqa: E402
from quadruped_ctrl_amd import workloads as W  # noqa: E402

SHARDS = {1: (1024, 1), 2: (4096, 1), 3: (4096, 4), 4: (8192, 8)}
_B = None


def _load(cfg):
    global _B
    B, world = SHARDS[cfg]
    _B = W.shard(W.make_config(cfg, batch=B * world), 0, world)
    return _B


def _fp64(args):
    cfg, lo, hi = args
    b = _B if _B is not None else _load(cfg)
    out = np.zeros((hi - lo, 12 * b["horizon"]))
    for i in range(lo, hi):
        H0, g0, A, lb, ub, _ = O.assemble(b, i)
        ve, _, _, Ar, lr, ur = O.reduce(H0, g0, A, lb, ub)
        _, Hr, gr, _, _, _ = O.reduce(*K.assemble(b, i), A, lb, ub)
        if gr.size:
            x, _, _, rc, irc = O.qpoases(Hr, gr, Ar, lr, ur, nwsr=5000)
            out[i - lo][~ve] = x
    return out


def _floor(args):
    cfg, i = args
    b = _B if _B is not None else _load(cfg)
    f = NF.robot_floor(b, i)
    return f["spread12"], f["spread_full"]


def main():
    cfgs = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    for cfg in cfgs:
        b = _load(cfg)
        B = b["batch"]
        t0 = time.time()
        ref, nwsr, bad = O.solve_packed(O.pack_updates(b), b)
        with Pool(8, initializer=_load, initargs=(cfg,)) as pool:
            step = 64
            gpu_like = np.concatenate(pool.map(_fp64, [(cfg, lo, min(lo + step, B)) for lo in range(0, B, step)]))
            e12 = np.abs(gpu_like[:, :12] - ref[:, :12]).max(1) / np.maximum(np.abs(ref[:, :12]).max(1), 1.0)
            efu = np.abs(gpu_like - ref).max(1) / np.maximum(np.abs(ref).max(1), 1.0)
            over = np.flatnonzero((e12 > 1e-4) | (efu > 1e-4))
            fl = pool.map(_floor, [(cfg, int(i)) for i in over])
        print(f"configs[{cfg}] shard of {B}: oracle failures {bad}, nWSR max {nwsr.max()}; first-step err median {np.median(e12):.2e} "
              f"p99 {np.percentile(e12, 99):.2e} max {e12.max():.2e} (robot {e12.argmax()}), over 1e-4: {(e12 > 1e-4).sum()} = "
              f"{(e12 > 1e-4).mean():.5f}; whole solution max {efu.max():.2e}, over 1e-4: {(efu > 1e-4).sum()} = {(efu > 1e-4).mean():.5f}  "
              f"[{time.time() - t0:.0f} s]")
        worst = 0.0
        for i, (s12, sfu) in zip(over, fl):
            ok12 = e12[i] < max(1e-4, 1.5 * s12)
            okfu = efu[i] < max(1e-4, 1.5 * sfu)
            worst = max(worst, e12[i] / max(s12, 1e-30) if e12[i] > 1e-4 else 0.0)
            if len(over) <= 40 or not (ok12 and okfu):
                print(f"   robot {i}: first-step {e12[i]:.3e} (spread {s12:.3e}) whole {efu[i]:.3e} (spread {sfu:.3e})"
                      + ("" if ok12 and okfu else "   OUTSIDE 1.5 x spread"))
        print(f"   largest first-step err / spread among the robots over 1e-4: {worst:.2f}")


if __name__ == "__main__":
    main()
