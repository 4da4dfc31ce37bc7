"""CPU suite: host-side logic and the C-ABI surface (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from oracle import oracle as O
from quadruped_ctrl_amd import binding, gait, workloads as W

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_once():
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    _build_once()
    # the caller's surface (qmpc.h: at most 25 entry points), the expert knobs and the test hooks: three headers, one library
    decl = lambda f: set(re.findall(r"^(?:int|const char\*) (qmpc_[a-z_]+)\s*\(", open(os.path.join(ROOT, "include", f)).read(), re.M))
    core, expert, debug = decl("qmpc.h"), decl("qmpc_expert.h"), decl("qmpc_debug.h")
    assert len(core) <= 25, sorted(core)
    assert not any(n.startswith(("qmpc_set_debug", "qmpc_debug_")) for n in core | expert)
    assert all(n.startswith(("qmpc_set_debug", "qmpc_debug_")) for n in debug)
    assert not (core & expert) and not (core & debug) and not (expert & debug)
    declared = sorted(core | expert | debug)
    assert set(declared) == set(binding.EXPORTS), (declared, binding.EXPORTS)
    lib = C.CDLL(binding.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.qmpc_abi_version() == binding.ABI_VERSION


def test_shim_exports_reference_symbols():
    _build_once()
    path = os.path.join(ROOT, "quadruped_ctrl_amd", "libconvexmpc_shim.so")
    if not os.path.exists(path):
        pytest.skip("shim not built yet")
    lib = C.CDLL(path)
    # src/MPC_Ctrl/convexMPC_interface.h:40-48
    for name in ("setup_problem", "update_problem_data", "get_solution",
                 "update_solver_settings", "update_problem_data_floats"):
        assert hasattr(lib, name), name


def test_argument_validation_without_gpu():
    _build_once()
    lib = binding.load_library()
    h = C.c_void_p()
    assert lib.qmpc_create(0, 0, 10, C.byref(h)) == 1          # bad batch
    assert lib.qmpc_create(0, 16, 99, C.byref(h)) == 1         # horizon > max
    assert lib.qmpc_setup(None, 0.026, 10, 0.4, 120.0) == 1
    assert lib.qmpc_destroy(None) == 1


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(binding.QmpcError):
        binding.BatchedConvexMPC(0)


def test_mpc_table_matches_reference_rule():
    # Gait.cpp:142-166 restated in C (oracle) vs the vectorised host version
    for n, off, dur in [(10, (0, 5, 5, 0), (5, 5, 5, 5)), (10, (5, 5, 0, 0), (4, 4, 4, 4)),
                        (14, (0, 4, 7, 11), (7, 7, 7, 7)), (16, (0, 8, 8, 0), (8, 8, 8, 8)),
                        (14, (0, 0, 0, 0), (14, 14, 14, 14))]:
        for it in range(n):
            assert np.array_equal(gait.mpc_table(n, off, dur, it), O.mpc_table(n, off, dur, it))
    # trot, iteration 0: step i uses phase (i+1)%10; feet 0,3 stance for phase<5
    t = gait.mpc_table(10, (0, 5, 5, 0), (5, 5, 5, 5), 0).reshape(10, 4)
    assert t[0].tolist() == [1, 0, 0, 1] and t[4].tolist() == [0, 1, 1, 0]
    assert np.all(t.sum(1) == 2)
    g = gait.OffsetDurationGait(10, (0, 5, 5, 0), (5, 5, 5, 5), "Trotting")
    g.setIterations(13, 13 * 3 + 5)
    assert g.iteration == 3 and abs(g.phase - (13 * 3 + 5) / 130.0) < 1e-7
    assert np.array_equal(g.getMpcTable(), gait.mpc_table(10, (0, 5, 5, 0), (5, 5, 5, 5), 3))


@pytest.mark.parametrize("idx,B,h", [(0, 1, 10), (1, 1024, 10), (2, 4096, 10), (3, 16384, 16), (4, 65536, 10)])
def test_workload_shapes(idx, B, h):
    small = 32
    b = W.make_config(idx, batch=small)
    assert b["horizon"] == h and b["batch"] == small
    assert b["traj"].shape == (small, 12 * h) and b["gait"].shape == (small, 4 * h)
    assert b["gait"].dtype == np.uint8 and b["p"].dtype == np.float32
    np.testing.assert_allclose(np.linalg.norm(b["q"], axis=1), 1, atol=1e-6)
    if idx == 4:
        assert np.all(b["gait"].reshape(small, h, 4)[:, 0].sum(1) >= 1)
    if idx in (1, 3):
        assert np.all(b["gait"].sum(1) == 2 * h)
    # deterministic
    b2 = W.make_config(idx, batch=small)
    assert all(np.array_equal(b[k], b2[k]) for k in ("p", "gait", "traj"))


def test_shard_is_disjoint_cover():
    b = W.make_config(2, batch=37)
    parts = [W.shard(b, r, 4) for r in range(4)]
    assert sum(p["batch"] for p in parts) == 37
    assert np.array_equal(np.concatenate([p["p"] for p in parts]), b["p"])
    assert np.array_equal(np.concatenate([p["gait"] for p in parts]), b["gait"])


# ---------------------------------------------------------------- caller-side packer
def _load_pack_golden(name):
    z = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    cmd = {k[4:]: z[k] for k in z.files if k.startswith("cmd_")}
    for k in ("batch", "horizon", "omni_mode"):
        cmd[k] = int(cmd[k])
    cmd["body_height"] = float(cmd["body_height"])
    rec = {k[4:]: z[k] for k in z.files if k.startswith("rec_")}
    return z, cmd, rec


@pytest.mark.parametrize("name", ["pack_h10", "pack_h16_omni"])
def test_oracle_pack_matches_golden(name):
    z, cmd, rec = _load_pack_golden(name)
    got, wpd, xci = O.pack_commands(cmd, np.float32(z["dt"]))
    for k, v in rec.items():
        assert np.array_equal(got[k], v), k            # float code restated op by op: bit exact
    assert np.array_equal(wpd, z["wpd_out"]) and np.array_equal(xci, z["xci_out"])
    assert np.array_equal(O.forces_to_body(cmd["r_body"], z["grf"]), z["f_ff"])


def test_oracle_pack_properties():
    """What ConvexMPCLocomotion.cpp:498-640 guarantees, checked on the restatement."""
    h = 10
    cmd = W.make_commands(256, horizon=h, seed=21)
    dt = np.float32(0.026)
    rec, wpd, xci = O.pack_commands(cmd, dt)
    tr = rec["traj"].reshape(-1, h, 12)
    stand = cmd["gait_type"] == 4
    assert stand.any() and (~stand).any()
    # standing: every row is the same trajInitial (:514-531)
    assert (tr[stand] == tr[stand][:, :1]).all()
    assert np.array_equal(tr[stand][:, 0, 3:5], cmd["stand_traj"][stand][:, 0:2])
    assert np.array_equal(wpd[stand], cmd["world_position_desired"][stand])     # untouched
    mv = ~stand
    # moving: rows other than yaw/x/y are constant; x,y,yaw are running float sums (:566-573)
    for j in (0, 1, 5, 6, 7, 8, 9, 10, 11):
        assert (tr[mv][:, :, j] == tr[mv][:, :1, j]).all(), j
    vw = tr[mv][:, 0, 9:11]
    for k in range(1, h):
        assert np.array_equal(tr[mv][:, k, 3], tr[mv][:, k - 1, 3] + dt * vw[:, 0])
        assert np.array_equal(tr[mv][:, k, 2], tr[mv][:, k - 1, 2] + dt * cmd["vel_des"][mv][:, 2])
    # the desired position is pulled to within 0.1 m of the estimate (:534-545)
    assert (np.abs(wpd[mv] - cmd["position"][mv][:, :2]) <= 0.1 + 1e-6).all()
    assert (np.abs(cmd["world_position_desired"][mv] - cmd["position"][mv][:, :2]) > 0.1).any()
    # v_des_world = rBody^T v_des_robot (:507): planar rotation by -yaw keeps the norm
    assert np.allclose(np.linalg.norm(vw, axis=1), np.linalg.norm(cmd["vel_des"][mv][:, :2], axis=1), atol=1e-6)
    # x_drag is the integral BEFORE its update; the update needs |vx| > 0.3 (:632-640)
    assert np.array_equal(rec["x_drag"], cmd["x_comp_integral"])
    moved = xci != cmd["x_comp_integral"]
    assert moved.any() and (~moved).any()
    assert (np.abs(cmd["v_world"][moved][:, 0]) > 0.3).all()
    # foot offsets, axis-major (:611-613), and the contact table (Gait.cpp:142-166)
    pf = cmd["p_foot"].reshape(-1, 4, 3)
    assert np.array_equal(rec["r"].reshape(-1, 3, 4), np.transpose(pf - cmd["position"][:, None, :], (0, 2, 1)))
    for i in range(0, 256, 17):
        assert np.array_equal(rec["gait"][i], gait.mpc_table(h, cmd["gait_offsets"][i], cmd["gait_durations"][i],
                                                             cmd["gait_iteration"][i]))
    assert np.array_equal(rec["weights"][3], np.float32([2.5, 2.5, 10, 50, 50, 100, 0, 0, 0.5, 0.2, 0.2, 0.1]))
    assert (rec["alpha"] == np.float32(4e-5)).all()


def test_pack_argument_validation_without_gpu():
    _build_once()
    lib = binding.load_library()
    cs, rs = binding.Command(), binding.Record()
    assert lib.qmpc_pack(None, 4, C.byref(cs), C.byref(rs), None) == 1
    assert lib.qmpc_forces_to_body(None, 4, None, None, None, None) == 1


@pytest.mark.parametrize("cfg,periodic", [(1, True), (2, True), (4, False)])
def test_config_rollout_continues_a_baseline_config(cfg, periodic):
    """bench.py's closed-loop leg (workloads.ConfigRollout): cycle 0 IS the BASELINE config's record; every later cycle has the
    contact table advanced by one step (periodic gaits: rolled -- OffsetDurationGait with iteration + 1, Gait.cpp:142-166,187-193;
    random tables: shifted, a fresh last row, at least one stance foot at step 0), the same layout and dtypes, finite states."""
    B = 24
    b = W.make_config(cfg, batch=B)
    ro = W.ConfigRollout(b, periodic=periodic)
    assert ro.record() is b
    h = b["horizon"]
    prev = b["gait"].reshape(B, h, 4).copy()
    for c in range(1, 5):
        ref, _, rc = O.solve_batch(ro.record())
        assert (rc == 0).all()
        ro.advance(ref[:, :12])
        r = ro.record()
        for k, v in b.items():
            if isinstance(v, np.ndarray):
                assert r[k].shape == v.shape and r[k].dtype == v.dtype, k
                assert np.isfinite(r[k].astype(np.float64)).all(), k
        g = r["gait"].reshape(B, h, 4)
        assert np.array_equal(g[:, 1:-1], prev[:, 2:])                # one MPC step later
        if periodic:
            assert np.array_equal(g[:, 0], prev[:, 1]) and np.array_equal(g[:, -1], prev[:, 0])   # gait period = horizon
        else:
            assert (g[:, 0] >= prev[:, 1]).all() and (g[:, 0].sum(1) >= 1).all()   # (a foot is put down where step 0 had none)
        # r = foot - body, axis-major (RobotState.cpp:25-27): stance feet have not moved in the world
        prev = g.copy()
    assert abs(float(np.linalg.norm(r["q"], axis=1).mean()) - 1.0) < 1e-5


def test_bench_whole_shard_parity_names_the_robots_over_the_flat_bound():
    """bench.py: parity_sample covers every robot of the shard, counts and names the robots over 1e-4 and compares them with
    the reference's own float evaluation-order spread (checked here with a stand-in for the GPU forces)."""
    import bench
    b = W.make_config(2, batch=96)
    ref, nw, bad = O.solve_packed(O.pack_updates(b), b)
    fake = ref[:, :12].astype(np.float32)
    fake[7] *= np.float32(1.0 + 3e-4)
    ps = bench.parity_whole_shard(b, fake, first=(ref[:32], nw[:32]))
    assert ps["robots"] == 96 and ps["whole_shard"] and ps["robots_over_1e-4"] == 1 and ps["worst_robots"][0]["robot"] == 7
    assert abs(ps["max_rel_grf_err"] - 3e-4) < 2e-5 and ps["spread_checked_on"] == 1
    assert ps["worst_checked_within_1.5x_reference_spread"] is False and ps["spread_unchecked"] == 0 and ps["all_over_1e-4_checked"]       # 3e-4 is not the reference's noise on that robot
    clean = bench.parity_whole_shard(b, ref[:, :12].astype(np.float32))
    assert clean["robots_over_1e-4"] == 0 and clean["worst_checked_within_1.5x_reference_spread"] is True and clean["spread_unchecked"] == 0


def test_bench_cpu_core_accounting():
    import bench
    cpus = bench.physical_core_cpus()
    assert 1 <= len(cpus) <= (os.cpu_count() or 1) and len(set(cpus)) == len(cpus)
    q = bench.cgroup_cpu_quota()
    assert q is None or q > 0
