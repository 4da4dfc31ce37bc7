"""CPU suite: the N>1 PLUMBING (independent robot shards, the result gather, max-over-ranks
timing) on world_size=2 with the gloo backend.

Sharding / collective arithmetic ONLY: there is no GPU here, so the per-rank "solve" is the
oracle standing in for the HIP solver (no product compute is exercised).  The product-side
N>1 path -- per-rank handles, RCCL all_gather_into_tensor of grf, per-rank timing -- runs in
the -m gpu suite (test_bench_two_ranks_dry_run, test_bench_two_ranks_gather) and on hardware
in the driver's SCALE run."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, outdir, use_gpu=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from quadruped_ctrl_amd import workloads as W
    full = W.make_config(2, batch=10)
    mine = W.shard(full, rank, world)
    if use_gpu:
        # the product: one handle per rank (the ranks share this box's GPU), the real HIP solver
        from quadruped_ctrl_amd.binding import BatchedConvexMPC
        m = BatchedConvexMPC(0, max_batch=mine["batch"], max_horizon=16)
        m.setup(mine["dt"], mine["horizon"], mine["mu"], mine["f_max"])
        q = m.solve(mine, full=True)["soln"]
        m.close()
    else:
        # stand-in for the GPU solve in the CPU suite: the checker itself
        from oracle import oracle as O
        q, nwsr, rc = O.solve_batch(mine)
    # the only collectives of the bench: barrier + MAX of elapsed + size gather
    dist.barrier()
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([mine["batch"]], dtype=torch.int64))
    # result collection as bench.py --gather does it: ONE all_gather_into_tensor of
    # the 12 first-step forces per robot; every rank ends up with every robot's forces
    grf = torch.from_numpy(np.ascontiguousarray(q[:, :12], np.float32))
    gathered = torch.empty((world * grf.shape[0], 12), dtype=torch.float32)
    dist.all_gather_into_tensor(gathered, grf)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), q=q, tmax=t.numpy(),
             sizes=np.array([int(s) for s in sizes]), gathered=gathered.numpy())
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process(tmp_path):
    from oracle import oracle as O
    if not O.have_ref():
        pytest.skip("no _ref")
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    from quadruped_ctrl_amd import workloads as W
    full = W.make_config(2, batch=10)
    qref, _, _ = O.solve_batch(full)
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert np.array_equal(np.concatenate([p["q"] for p in parts]), qref)
    assert all(p["tmax"][0] == 2.0 for p in parts)
    assert parts[0]["sizes"].tolist() == [5, 5]
    for p in parts:       # every rank holds the single-process result after the gather
        assert np.array_equal(p["gathered"], qref[:, :12].astype(np.float32))


@pytest.mark.gpu
def test_two_rank_sharding_real_solver(tmp_path):
    """The same two-rank plumbing with the PRODUCT on every rank: each rank solves its shard with
    the HIP solver, the all-gather collects the forces; the union equals a single-process solve of the whole batch bit
    for bit, and the oracle within north_star's tolerance."""
    from oracle import oracle as O
    from quadruped_ctrl_amd import workloads as W
    from quadruped_ctrl_amd.binding import BatchedConvexMPC
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path), True), nprocs=world, join=True)
    full = W.make_config(2, batch=10)
    m = BatchedConvexMPC(0, max_batch=10, max_horizon=16)
    m.setup(full["dt"], full["horizon"], full["mu"], full["f_max"])
    single = m.solve(full, full=True)["soln"]
    m.close()
    parts = [np.load(tmp_path / f"r{r}.npz") for r in range(world)]
    assert np.array_equal(np.concatenate([p["q"] for p in parts]), single)
    for p in parts:
        assert np.array_equal(p["gathered"], single[:, :12].astype(np.float32))
    qref, _, rc = O.solve_batch(full)
    assert (rc == 0).all()
    err = np.abs(single[:, :12] - qref[:, :12]).max(1) / np.maximum(np.abs(qref[:, :12]).max(1), 1.0)
    assert err.max() < 1e-4
