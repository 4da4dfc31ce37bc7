"""GPU suite (-m gpu): the wide randomized parity sweeps of tools/stress_parity.py / tools/stress_large.py, trimmed to
run inside the driver's suite (~1900 robots, about a minute of host time).

(a) SOLVER parity per family: the GPU dumps its own reduced QP (H_red, g_red, fp64) and the full solution; the
    reference's qpOASES (oracle/_ref, iteration cap lifted -- the reference caps at nWSR = 100, SolverMPC.cpp:435) solves
    that very QP; relative difference of the solutions <= 1e-8 (measured ~1e-12).  Beyond 192 rows the QP is not dumped
    (448 x 448 work items): there the fp64 Kronecker model's QP stands in (objective, feasibility, solution).
(b) END-TO-END parity against the ORACLE PIPELINE (float restatement of SolverMPC.cpp:296-525 + the reference's
    qpOASES at its own nWSR = 100) for every robot the reference solves under its cap: first-step GRF within
    max(1e-4, 1.5 x spread_i), spread_i = the pairwise spread of the reference's six float evaluation orders on that robot
    (tests/test_gpu_parity.py::bound_for), at every horizon; the spread is evaluated only for robots over the flat 1e-4.
    Printed for every family: frac_over_1e-4.
"""
import time

import numpy as np
import pytest

from oracle import kron_model as K
from oracle import oracle as O
from quadruped_ctrl_amd import workloads as W
from test_gpu_parity import bound_for, rel_f0, report

pytestmark = pytest.mark.gpu


def hard_commands(B, h, seed, stand_fraction=0.1, f_max=120.0):
    cmd = W.make_commands(B, horizon=h, seed=seed, stand_fraction=stand_fraction)
    rec, _, _ = O.pack_commands(cmd, np.float32(0.026))
    rec.update(dt=0.026, mu=0.4, f_max=f_max)
    return rec


def low_fmax(B, seed):
    b = W.make_config(2, batch=B)
    rng = np.random.default_rng(seed)
    b["f_max"] = 40.0
    b["traj"].reshape(B, 10, 12)[:, :, 10] = rng.uniform(-2, 2, (B, 1))
    return b


def random_tables(B, h, seed):
    """random contact tables at a long horizon: n_r anywhere between 3 h and 12 h -- every route of the call in one batch"""
    rng = np.random.default_rng(seed)
    b = W.make_long_horizon(B, h, "stand", seed=seed)
    g = b["gait"].reshape(B, h, 4)
    for i in range(B):
        p = rng.uniform(0.0, 0.6)
        g[i] = (rng.uniform(size=(h, 4)) >= p).astype(g.dtype)
        g[i, 0, rng.integers(4)] = 1
    b["x_drag"][:] = rng.normal(0, 0.4, B).astype(np.float32)
    return b


# family -> (maker, decoupled path for every robot of the 128- / 192-row classes?, robots checked against the oracle pipeline)
SMALL = {
    "cfg2_mixed_gaits_h10": (lambda: W.make_config(2, batch=320), False, 320),
    "cfg4_random_contacts_stairs_h10": (lambda: W.make_config(4, batch=320), False, 320),
    "cfg3_trot_h16": (lambda: W.make_config(3, batch=96), False, 96),
    "hard_commands_h10": (lambda: hard_commands(256, 10, 5), False, 0),
    "hard_commands_h14": (lambda: hard_commands(64, 14, 6), False, 0),
    "low_fmax_h10": (lambda: low_fmax(160, 9), False, 0),
    "standing_h10_decoupled": (lambda: W.make_standing(160, 10), True, 48),
    "standing_h14_decoupled": (lambda: W.make_standing(96, 14), True, 32),
    "standing_h16_tight_fmax_decoupled": (lambda: hard_commands(48, 16, 11, stand_fraction=1.0, f_max=40.0), True, 0),
    "trot_h24": (lambda: W.make_long_horizon(64, 24, "trot", seed=21), False, 24),
    "bounding_h36": (lambda: W.make_long_horizon(32, 36, "bound", seed=22), False, 8),
    "random_tables_h20": (lambda: random_tables(64, 20, 23), False, 0),
}


@pytest.mark.parametrize("family", list(SMALL))
def test_stress_family_solver_and_pipeline_parity(family, mpc_factory):
    mk, split, npipe = SMALL[family]
    b = mk()
    B, h = b["batch"], b["horizon"]
    m = mpc_factory(b)
    if split:
        m.set_split(2)
    Hd, gd, ld = m.debug_dump(B)
    res = m.solve(b, full=True)
    m.debug_off()
    Hd, gd = Hd.cpu().numpy(), gd.cpu().numpy()
    st = res["status"]
    t0 = time.time()
    worst, nbig, nchk = 0.0, 0, 0
    for i in range(B):
        H, g, A, lb, ub, x0 = O.assemble(b, i)
        ve, Hr, gr, Ar, lr, ur = O.reduce(H, g, A, lb, ub)
        n = gr.size
        if n == 0:
            assert not res["soln"][i].any()
            continue
        if n > 192:   # (no dump beyond 192 rows: test_stress_large_problems covers those routes)
            nbig += 1
            continue
        xq, y, used, rc, irc = O.qpoases(Hd[i][:n, :n], gd[i][:n], Ar, lr, ur, nwsr=20000)
        assert rc == 0 and irc == 0, (family, i)
        xs = res["soln"][i][~ve]
        worst = max(worst, np.abs(xs - xq).max() / max(np.abs(xq).max(), 1.0))
        nchk += 1
    print(f"{family}: B={B} h={h} solver vs uncapped qpOASES on the GPU's own QP: worst {worst:.2e} over {nchk} robots "
          f"({nbig} beyond 192 rows skipped) | iters mean {res['iters'].mean():.2f} max {res['iters'].max()} | spilled "
          f"{int(((st & 128) != 0).sum())} handed back {int(((st & 16) != 0).sum())} | {time.time() - t0:.1f} s host")
    assert ((st & 47) == 0).all(), np.unique(st)
    assert worst < 1e-8
    if npipe:
        idx = np.arange(min(npipe, B))
        ref, nwsr, rc = O.solve_batch(b, idx)
        ok = idx[(rc == 0) & (nwsr < 100)]      # robots the reference solves under its own cap
        pos = np.nonzero((rc == 0) & (nwsr < 100))[0]
        err = rel_f0(res["grf"][ok], ref[pos])
        bd = bound_for(b, ok, err=err)
        report(f"{family} end to end vs the oracle pipeline ({len(ok)} of {len(idx)} under nWSR = 100)", err, bd)
        print(f"   frac_over_1e-4 = {(err > 1e-4).mean():.4f}")
        assert len(ok) >= len(idx) // 4
        assert (err < bd).all()


LARGE = {
    "trot_h36_nr216": (lambda: W.make_long_horizon(24, 36, "trot", seed=6), 8),
    "stand_h18_nr216": (lambda: W.make_long_horizon(24, 18, "stand", seed=7), 8),
    "stand_h27_nr324": (lambda: W.make_long_horizon(16, 27, "stand", seed=8), 6),
    "stand_h36_nr432": (lambda: W.make_long_horizon(10, 36, "stand", seed=9), 4),
    "braking_h20_nr240": (lambda: W.make_standing(16, 20, seed=10), 6),
    "random_tables_h36": (lambda: random_tables(24, 36, 13), 6),
}


@pytest.mark.parametrize("family", list(LARGE))
def test_stress_large_problems(family, mpc_factory):
    """The large-problem path (192 < n_r <= 432): every robot against the reference's qpOASES (cap lifted) on the fp64
    Kronecker model's reduced QP -- solution <= 1e-6, objective <= 1e-12, feasibility <= 1e-9 -- and, NEW,
    END TO END against the oracle pipeline (float assembly + qpOASES at nWSR = 100) within the reference's own float-order
    spread for the robots the reference solves under its cap."""
    mk, npipe = LARGE[family]
    b = mk()
    B, h = b["batch"], b["horizon"]
    m = mpc_factory(b)
    res = m.solve(b, full=True)
    st = res["status"]
    assert ((st & 47) == 0).all(), np.unique(st)
    nst = (b["gait"].reshape(B, -1) != 0).sum(1)
    wx = wf = wi = 0.0
    nbig = 0
    t0 = time.time()
    for i in range(B):
        H, g = K.assemble(b, i)
        Hf, gf, A, lb, ub, x0 = O.assemble(b, i)
        ve, Hr, gr, Ar, lr, ur = O.reduce(Hf, gf, A, lb, ub)
        vi = np.nonzero(~ve)[0]
        if vi.size == 0:
            continue
        Hm, gm = H[np.ix_(vi, vi)], g[vi]
        xq, y, used, rc, irc = O.qpoases(Hm, gm, Ar, lr, ur, nwsr=100000)
        assert rc == 0 and irc == 0
        nbig += vi.size > 192
        xs = res["soln"][i][~ve]
        f = lambda x: 0.5 * x @ Hm @ x + gm @ x  # noqa: E731
        ax = Ar @ xs
        wi = max(wi, np.maximum(lr - ax, 0).max(), np.maximum(ax - ur, 0).max())
        wf = max(wf, abs(f(xs) - f(xq)) / max(abs(f(xq)), 1e-30))
        wx = max(wx, np.abs(xs - xq).max() / max(np.abs(xq).max(), 1.0))
    print(f"{family}: B={B} n_r {3 * nst.min()}..{3 * nst.max()} ({nbig} beyond 192 rows) iters mean {res['iters'].mean():.1f} max "
          f"{res['iters'].max()} | vs qpOASES on the fp64 model's QP: x {wx:.2e} objective {wf:.2e} infeasibility {wi:.2e} | "
          f"{time.time() - t0:.1f} s host")
    assert nbig > 0
    assert wx < 1e-6 and wf < 1e-12 and wi < 1e-9
    # end to end: the oracle pipeline on the hardest-to-fake subset -- robots beyond 192 rows first
    order = np.argsort(-nst)[:npipe]
    t0 = time.time()
    ref, nwsr, rc = O.solve_batch(b, order)
    under = (rc == 0) & (nwsr < 100)
    ok = order[under]
    print(f"   oracle pipeline: {int(under.sum())} of {len(order)} robots under the reference's nWSR = 100 cap "
          f"(nWSR {nwsr.tolist()}), {time.time() - t0:.1f} s host")
    if len(ok):
        err = rel_f0(res["grf"][ok], ref[under])
        bd = bound_for(b, ok, err=err)
        report(f"{family} end to end vs the oracle pipeline (n_r {3 * nst[ok].min()}..{3 * nst[ok].max()})", err, bd)
        print(f"   frac_over_1e-4 = {(err > 1e-4).mean():.4f}")
        assert (err < bd).all()
