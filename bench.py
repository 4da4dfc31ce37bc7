#!/usr/bin/env python
"""bench.py -- convex-MPC QP solves/sec on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--config C | --workload standing --horizon H] [--batch B]

One "step" = one pass of the whole hot path (state -> linearisation -> condensed
QP -> exact QP solve -> first-step GRF) over one batch of synthetic robot states
already resident in HBM.  Default workload is BASELINE.json configs[1]
(batch=1024 robots, trot, horizon=10).  N>1 (launched by torch.distributed.run,
one rank per GPU) shards independent robots across ranks with no data-path
collective: every rank solves its own `batch` robots (weak scaling); RCCL carries
the timing barrier / MAX and, with --gather, ONE all_gather_into_tensor of the
48-byte result rows per step (consumers that need every force on
every GPU).

Timing: the K-step region of the contract (barrier + torch.cuda.synchronize on both sides, MAX over ranks) is run
`repeats` times (default 25, fewer when a region takes long); value / ms_per_step are the MEDIAN region's,
ms_per_step_min / _max the spread.  N > 1: `rank_devices` lists every rank's device name, PCI bus id and uuid,
`rccl_world_size` the size of the RCCL communicator that carried the barriers.

Contract fields (`value`, `ms_per_step`, `roofline`): the BASELINE config's static synthetic inputs, NO order hint
(qmpc_set_order_hint off) and the library's size order / proxy staging ON (its default since round 6: scheduling by what THIS
call's own records say -- contact-table size and a tracking-error proxy --, never by a previous solve; `size_order.off_same_inputs`
is the rate with it off, robot = workgroup index) -- the order hint of the library (scheduling by the previous call's iteration counts) would be
EXACT here, because every step re-solves the same inputs; it is reported beside the contract fields
(`order_hint.hinted_same_inputs`), and what a controller really gets from it -- the PREVIOUS MPC cycle's counts -- is measured
by the `closed_loop` leg: the config continued as a closed-loop rollout (workloads.ConfigRollout), >= 8 consecutive MPC cycles
pre-uploaded, the timed steps walking back and forth over them so that every call's hint is an adjacent cycle's.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      bound "fp64_valu": the kernel issues vector fp64 (no MFMA; on MI355X the dense
                fp64 MFMA peak equals the fp64 vector peak, 78.6 TFLOP/s).  `achieved` = the
                ALGORITHMIC flops of SURVEY.md 8d (what the reference's dense formulation needs)
                over the HIP-event kernel time; `executed_*` = what the kernel really issues, from
                the SQ_INSTS_VALU_*_F64 counters of the last PMC run of THIS kernel source
                (profiles/pmc_latest.json carries the source hash; stale -> null)
  roofline_hbm  algorithmic HBM bytes (728 B/QP at h=10) vs 8 TB/s; traffic from PMC likewise
  cpu_baseline  the oracle pipeline (C restatement of the reference assembly + the
                reference's own qpOASES, oracle/_ref): one host core, and one worker
                process per host core (core count + CPU model stated)
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP64_PEAK_TFLOPS = 78.6        # MI355X vector FP64 (= dense FP64 MFMA peak)
# how cpu_baseline.all_cores is taken (rounds 1-4: version 1 = one worker per LOGICAL cpu, 256 / 128 of them throttled onto the
# container's 16-CPU cgroup quota; since round 5: version 2) -- speedups derived from it are comparable within a version only
BASELINE_METHOD = {"version": 3, "all_cores": "one worker process per PHYSICAL core, pinned, at most the cgroup CPU quota (>= 1)",
                   "contract_regions": "no order hint (qmpc_set_order_hint off: nothing from a previous solve) since round 5 (rounds 1-4: exact "
                                       "hint); since round 6 (version 3) with the library's size order / proxy staging on -- scheduling by "
                                       "this call's own input records; size_order.off_same_inputs = the version-2 quantity"}
KERNEL_SOURCES = ["quadruped_ctrl_amd/csrc/qmpc_kernels.hip", "quadruped_ctrl_amd/csrc/qmpc_engine.hip",
                  "quadruped_ctrl_amd/csrc/qmpc_wave.h", "quadruped_ctrl_amd/csrc/qmpc_cmd.h",
                  "quadruped_ctrl_amd/csrc/qmpc_device.h"]


def kernel_source_hash():
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def alg_bytes_per_qp(h):
    """4*(26 + 12 + 12h + 2) + 4h + 48."""
    return 4 * (26 + 12 + 12 * h + 2) + 4 * h + 48


def alg_flops_per_qp(h, nr_mean, k_mean):
    """F_cond(h) + 2 F_fact(n) + K F_iter(n)."""
    f_cond = 3744 * h * (h + 1) * (h + 2) / 6 + 4056 * h + 338 * h + 312 * h * h
    f_fact = nr_mean ** 3 / 3 + nr_mean ** 2
    f_iter = 2 * nr_mean ** 2 + 40 * nr_mean
    return f_cond + 2 * f_fact + k_mean * f_iter


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def parity_whole_shard(b, gpu_grf, first=None, max_floors=48):
    """The run's parity check: EVERY robot of this GPU's shard against the oracle pipeline (float assembly restatement + the
    reference's qpOASES at nWSR = 100; 0.5 - 7 s of host time), not a sample.  Robots over north_star's flat 1e-4 are named and
    compared with the reference's own float evaluation-order spread on that robot (oracle/noise_floor.py; at most `max_floors`
    of them, the worst first).  `first` = (q_soln, nwsr) already computed for the first rows (the CPU-baseline probe)."""
    from oracle import oracle
    B = b["batch"]
    n0 = 0 if first is None else len(first[1])
    q = np.zeros((B, 12))
    nwsr = np.zeros(B, np.int32)
    if n0:
        q[:n0], nwsr[:n0] = np.asarray(first[0])[:, :12], first[1]
    if n0 < B:
        rest = oracle.solve_packed(oracle.pack_updates(b, range(n0, B)), b)
        q[n0:], nwsr[n0:] = rest[0][:, :12], rest[1]
    # robots on which the reference itself stops at its cap of nWSR = 100 working-set recalculations (SolverMPC.cpp:435; it
    # ignores init()'s return value and hands on a non-optimal point) are counted, not compared
    capped = nwsr >= 100
    err = np.abs(gpu_grf[:B].astype(np.float64) - q).max(1) / np.maximum(np.abs(q).max(1), 1.0)
    err[capped] = 0.0
    over = np.flatnonzero(err > 1e-4)
    worst = over[np.argsort(-err[over])][:max_floors]
    ratio, named = 0.0, []
    if worst.size:
        from oracle import noise_floor
        for i in worst:
            sp = noise_floor.robot_floor(b, int(i))["spread12"]
            ratio = max(ratio, err[i] / max(sp, 1e-300))
            named.append({"robot": int(i), "err": float(err[i]), "reference_float_order_spread": float(sp)})
    ok = err[~capped] if (~capped).any() else np.zeros(1)
    return {"robots": int((~capped).sum()), "whole_shard": True, "reference_hit_nwsr_cap": int(capped.sum()),
            "max_rel_grf_err": float(ok.max()), "median_rel_grf_err": float(np.median(ok)),
            "frac_over_1e-4": float((ok > 1e-4).mean()), "robots_over_1e-4": int(over.size),
            "worst_robots": named[:8], "max_err_over_reference_spread": (float(ratio) if worst.size else None),
            # (the spread costs six assemblies + six qpOASES solves per robot: it is evaluated for the WORST `max_floors` robots
            #  over 1e-4 only; spread_unchecked > 0 means the statement does not cover every robot over 1e-4)
            "worst_checked_within_1.5x_reference_spread": (bool(ratio < 1.5) if worst.size else True),
            "spread_checked_on": int(worst.size), "spread_unchecked": int(over.size - worst.size),
            "all_over_1e-4_checked": bool(over.size == worst.size),
            "note": "first-step GRF of the GPU vs the oracle pipeline (float assembly restatement + the reference's qpOASES), every "
                    "robot of the shard; the reference assembles in float with an operation order it leaves to Eigen, and its own "
                    "answers spread by more than 1e-4 on the robots named here (tests/test_gpu_parity.py::test_full_shard_vs_oracle "
                    "asserts err < max(1e-4, 1.5 x spread) on every robot)"}


def physical_core_cpus():
    """One logical CPU id per distinct (package, core) pair of /proc/cpuinfo -- SMT siblings counted once -- restricted to the
    CPUs this process may run on."""
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    try:
        first, cur = {}, {}
        for line in list(open("/proc/cpuinfo")) + ["\n"]:
            if ":" in line:
                k, v = line.split(":", 1)
                cur[k.strip()] = v.strip()
            elif not line.strip() and cur:
                cpu = int(cur.get("processor", -1))
                key = (cur.get("physical id"), cur.get("core id"))
                if cpu in allowed and None not in key:
                    first.setdefault(key, cpu)
                cur = {}
        cpus = sorted(first.values())
        return cpus or sorted(allowed)
    except (OSError, ValueError):
        return sorted(allowed)


def cgroup_cpu_quota():
    """CPUs' worth of time this container may use per period (cgroup v2 cpu.max / v1 cfs quota); None = unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline(b, spec, budget_s=8.0, all_cores=True, gpu_grf=None):
    """Reference-style CPU pipeline, bounded sample: one host core in-process, then one worker process per PHYSICAL host
    core (oracle/cpu_worker.py).  gpu_grf: the GPU's first-step forces of the same robots -- the oracle's answers double as
    the run's parity check, over the WHOLE shard (`parity_sample`: max relative GRF error, the fraction of robots over
    north_star's 1e-4, the robots concerned)."""
    try:
        from oracle import oracle
        if not oracle.have_ref():
            return None
        n = min(b["batch"], 1024)
        arr = oracle.pack_updates(b, range(n))
        t0 = time.perf_counter()
        first = oracle.solve_packed(arr, b)    # also the probe for the rate
        t1 = time.perf_counter() - t0
        parity = None
        try:
            if gpu_grf is not None:
                parity = parity_whole_shard(b, gpu_grf, first=(first[0][:n], first[1][:n]),
                                            max_floors=48 if b["horizon"] <= 16 else 4)   # (a float-order spread at h = 36 takes seconds)
        except Exception as e:
            parity = {"error": repr(e)}
        reps = max(1, int(budget_s / max(t1, 1e-6)) - 1)
        reps = min(reps, 50)
        t0 = time.perf_counter()
        for _ in range(reps):
            oracle.solve_packed(arr, b)
        dt = time.perf_counter() - t0
        res = {"value": n * reps / dt, "unit": "QP solves/s", "cores": 1, "kind": "port", "baseline_method": BASELINE_METHOD,
               "cpu_model": cpu_model(), "parity_sample": parity,
               "sample": f"{reps}x first {n} robots of the workload; C restatement of "
                         f"SolverMPC.cpp assembly (fp32 dense) + the reference's own "
                         f"qpOASES 3.2.0 build (oracle/_ref), single thread, {dt:.1f} s of CPU time"}
        if all_cores:
            procs = []
            try:
                logical = os.cpu_count() or 1
                cpus = physical_core_cpus()
                # the container may be allowed fewer CPUs than it sees (the GPU boxes: 256 logical CPUs visible, a cgroup quota of
                # 16): more workers than that are throttled, not parallel -- rounds 1-4 read the collapse as the CPU's
                quota = cgroup_cpu_quota()
                if quota is not None and max(1, int(quota)) < len(cpus):   # (a quota below one CPU: one worker)
                    nq = max(1, int(quota))
                    step = len(cpus) // nq
                    cpus = cpus[::step][:nq]
                cores = len(cpus)
                # one pass of a worker's robots ~1 s, so that every worker gets several passes into its window
                per_solve = dt / (n * reps)
                wspec = dict(spec, batch=int(min(256, max(8, 1.0 / max(per_solve, 1e-6)))))
                env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
                procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", json.dumps(wspec), str(budget_s), str(cpu)],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT, env=env)
                         for cpu in cpus]
                outs = [json.loads(p.communicate(timeout=180)[0].strip().splitlines()[-1]) for p in procs]
                total = sum(o["solved"] for o in outs)
                span = max(o["elapsed"] for o in outs)
                res["all_cores"] = {"value": total / span, "unit": "QP solves/s", "cores": cores, "physical_cores_visible": len(physical_core_cpus()),
                                    "baseline_method": BASELINE_METHOD,
                                    "logical_cores": logical, "cgroup_cpu_quota": quota, "per_core": total / span / cores, "cpu_model": cpu_model(),
                                    "sample": f"{cores} worker processes, each pinned to a PHYSICAL host core of its own -- as many as the "
                                              f"container's cgroup CPU quota allows ({quota}; {logical} logical CPUs visible; the "
                                              f"reference is single-threaded and non-reentrant) --, each looping over the first "
                                              f"{wspec['batch']} robots for {budget_s:.0f} s"}
            except Exception as e:
                res["all_cores"] = {"value": None, "error": repr(e)}
                for p in procs:
                    try:
                        p.kill()
                    except Exception:
                        pass
        return res
    except Exception as e:  # baseline is reporting only; never fail the bench
        return {"value": None, "error": repr(e)}


def cpu_baseline_sparse(b, spec, budget_s=8.0, all_cores=True, gpu_grf=None, n_problems=24):
    """north_star's "qpOASES/osqp path": the reference's SPARSE leg timed beside --model sparse.  SparseCMPC's QP of the first
    robots (oracle/sparse_model.py: a numpy restatement of SparseCMPC.cpp:31-73, built once, NOT timed -- the reference builds
    it with Eigen triplets) is solved by the reference's OWN vendored OSQP 0.5.0 (oracle/_ref/libosqp_ref.so, compiled
    unmodified; eps_abs = eps_rel = 1e-5, set-up + solve + clean-up per MPC cycle like OsqpTriples.cpp:57-142): one core
    in-process, then one pinned worker per physical core up to the cgroup quota."""
    try:
        from oracle import sparse_model as SM
        n = min(b["batch"], n_problems)
        qps, probs, rows = [], [], []
        for i in range(n):
            pr = SM.from_batch(b, i, weights=b["weights"][i].astype(np.float64), alpha=float(b["alpha"][i]), mu=b["mu"],
                               f_max=b["f_max"])
            if len(pr["blocks"]):
                qps.append(SM.osqp_prepare(pr["P"], pr["q"], pr["A"], pr["l"], pr["u"]))
                probs.append(pr)
                rows.append(i)
        if not qps:
            return None
        t0 = time.perf_counter()
        first = [SM.osqp_call(c) for c in qps]
        t1 = time.perf_counter() - t0
        its = [f[2] for f in first]
        ok = [f[1] == 1 for f in first]
        parity = None
        if gpu_grf is not None:
            worst = 0.0
            for c, pr, i in zip(qps, probs, rows):
                f = SM.first_step_forces(c["x"], pr)
                worst = max(worst, float(np.abs(gpu_grf[i].astype(np.float64) - f).max() / max(np.abs(f).max(), 1.0)))
            parity = {"robots": len(rows), "max_rel_grf_diff_gpu_exact_vs_osqp_at_reference_eps": worst,
                      "note": "the GPU returns the exact minimiser of SparseCMPC's QP (tests: <= 1e-5 against OSQP driven to 1e-10); "
                              "OSQP at the reference's eps = 1e-5 stops a few per cent from it on the small force components"}
        reps = max(1, min(int(budget_s / max(t1, 1e-6)) - 1, 2000))
        t0 = time.perf_counter()
        for _ in range(reps):
            for c in qps:
                SM.osqp_call(c)
        dt = time.perf_counter() - t0
        res = {"value": len(qps) * reps / dt, "unit": "QP solves/s", "cores": 1, "kind": "reference", "baseline_method": BASELINE_METHOD,
               "solver": "the reference's vendored OSQP 0.5.0 (oracle/_ref/libosqp_ref.so), eps 1e-5, cold start, osqp_setup + osqp_solve + "
                         "osqp_cleanup per solve (OsqpTriples.cpp:57-142); adaptive_rho_interval pinned to 50 (oracle/osqp_shim.c)",
               "osqp_iterations_mean": float(np.mean(its)), "osqp_iterations_max": int(max(its)), "all_solved": bool(all(ok)),
               "cpu_model": cpu_model(), "parity_sample": parity,
               "sample": f"{reps}x the sparse QPs of the first {len(qps)} robots (12 h states + 3 per stance foot-step as variables), "
                         f"single thread, {dt:.1f} s of CPU time; the QP BUILD (SparseCMPC::run's triplets) is not in the time"}
        if all_cores:
            procs = []
            try:
                cpus = physical_core_cpus()
                quota = cgroup_cpu_quota()
                if quota is not None and max(1, int(quota)) < len(cpus):
                    nq = max(1, int(quota))
                    cpus = cpus[::len(cpus) // nq][:nq]
                wspec = dict(spec, batch=n_problems, model="sparse", sparse_problems=n_problems)
                env = dict(os.environ, PYTHONPATH=ROOT, OMP_NUM_THREADS="1")
                procs = [subprocess.Popen([sys.executable, "-m", "oracle.cpu_worker", json.dumps(wspec), str(budget_s), str(cpu)],
                                          stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, cwd=ROOT, env=env)
                         for cpu in cpus]
                outs = [json.loads(p.communicate(timeout=240)[0].strip().splitlines()[-1]) for p in procs]
                total = sum(o["solved"] for o in outs)
                span = max(o["elapsed"] for o in outs)
                res["all_cores"] = {"value": total / span, "unit": "QP solves/s", "cores": len(cpus), "cgroup_cpu_quota": quota,
                                    "per_core": total / span / len(cpus), "baseline_method": BASELINE_METHOD,
                                    "sample": f"{len(cpus)} pinned worker processes, each looping over the sparse QPs of the first "
                                              f"{n_problems} robots for {budget_s:.0f} s"}
            except Exception as e:
                res["all_cores"] = {"value": None, "error": repr(e)}
                for p in procs:
                    try:
                        p.kill()
                    except Exception:
                        pass
        return res
    except Exception as e:
        return {"value": None, "error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=1, help="BASELINE.json configs index")
    ap.add_argument("--workload", choices=["config", "standing", "trot", "long-trot", "long-bound", "long-stand"], default="config",
                    help="'standing' = all four feet down for the whole horizon (the robot's default posture, the "
                         "reference's Standing gait, ConvexMPCLocomotion.cpp:35: n_r = 12 h); 'trot' = trot at --horizon")
    ap.add_argument("--horizon", type=int, default=10, help="for --workload standing / trot (reference: 10, 14, 16)")
    ap.add_argument("--batch", type=int, default=None, help="robots per GPU (override)")
    ap.add_argument("--repeats", type=int, default=25,
                    help="the K-step timed region (barrier + synchronize on both sides, MAX over ranks) is run R times; the "
                         "contract fields come from the MEDIAN repeat (min / max reported beside it).  R is cut down when "
                         "R x K steps would take more than ~10 s (never below 3)")
    ap.add_argument("--settle", type=float, default=0.3,
                    help="seconds of untimed load before the W warmup steps (GPU clock ramp); 0 = none")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-all-cores", action="store_true", help="cpu_baseline on one core only")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the extra two-stream measurement")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: after every solve, all_gather_into_tensor the grf[shard][12] rows over RCCL so that "
                         "every rank holds all forces (inside the timed region)")
    ap.add_argument("--caller-side", choices=["fused", "three-calls"], default=None,
                    help="time command -> record -> solve -> body-frame forces per step instead of the solve alone "
                         "(SURVEY row a12 on the GPU; not the headline configuration): 'fused' = one "
                         "qmpc_solve_commands launch, 'three-calls' = qmpc_pack + qmpc_solve + qmpc_forces_to_body")
    ap.add_argument("--order-hint", choices=["auto", "off"], default="off",
                    help="qmpc_set_order_hint DURING THE CONTRACT REGIONS: 'off' (default) = plain order, robot = workgroup index -- the "
                         "bench solves the SAME inputs every step, so the library's hint (the previous call's iteration counts) would be "
                         "exact there; 'auto' = the library default, for experiments (the line then says exact_in_contract_regions).  "
                         "Either way the hinted rate on the same inputs and the closed-loop rollout, where the hint is the previous MPC "
                         "cycle's, are measured beside the contract fields")
    ap.add_argument("--size-order", choices=["on", "off"], default="on",
                    help="qmpc_set_size_order during the contract regions: 'on' (default, the library's default) = multi-round launches take "
                         "the robots largest / highest proxy score first, one-round launches stage the sweep's priority per CU by the proxy "
                         "-- all from this call's own records; 'off' = robot = workgroup index.  The other setting is measured beside it")
    ap.add_argument("--no-extras", "--no-plain-order", dest="no_extras", action="store_true",
                    help="skip the extra regions (hinted same inputs, closed loop): profiling runs -- every launch of the trace is then "
                         "a launch of the contract loop")
    ap.add_argument("--no-closed-loop", action="store_true", help="skip the closed-loop leg only")
    ap.add_argument("--cl-cycles", type=int, default=8, help="consecutive MPC cycles of the closed-loop leg (>= 4)")
    ap.add_argument("--model", choices=["dense", "sparse"], default="dense",
                    help="'sparse' = the reference's SPARSE formulation (SparseCMPC.cpp:31-73, qmpc_set_model(QMPC_MODEL_SPARSE)) with "
                         "SparseCMPC's own parameters (mu 1, its weights, g = -9.81, ConvexMPCLocomotion.cpp:732-756); the CPU baseline is "
                         "then the reference's OSQP leg (and the dense qpOASES pipeline on the same states beside it)")
    ap.add_argument("--max-iter", type=int, default=None,
                    help="census runs (tools/census.sh): cap the active-set iterations (qmpc_settings); robots that need more stop "
                         "with QMPC_ST_MAXITER -- counters by cap give the cost of ONE iteration as a slope")
    ap.add_argument("--no-hint", action="store_true",
                    help="do not tell the solver the workload's max stance foot-steps (launch every size class)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if args.gpus > 1:
        if world != args.gpus:
            raise SystemExit("for --gpus N>1 launch with: python -m torch.distributed.run "
                             "--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # QMPC_BENCH_BACKEND=gloo + fewer GPUs than ranks: dry run of the N>1 path on a smaller box
        backend = os.environ.get("QMPC_BENCH_BACKEND", "nccl")
        local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    dev = local if args.gpus > 1 else 0
    torch.cuda.set_device(dev)

    from quadruped_ctrl_amd import workloads
    from quadruped_ctrl_amd.binding import BatchedConvexMPC

    # per-rank batch: the named config's batch on one GPU; rank r gets its own
    # independent robots (different seed stream via the config generator + rank)
    if args.workload == "config":
        cfg_batch = {0: 1, 1: 1024, 2: 4096, 3: 16384, 4: 65536}[args.config]
        per_gpu = args.batch or (cfg_batch if args.config in (0, 1, 2) else cfg_batch // (4 if args.config == 3 else 8))
        full = workloads.make_config(args.config, batch=per_gpu * world)
        spec = {"kind": "config", "config": args.config, "batch": per_gpu}
        wname = f"BASELINE.json configs[{args.config}]"
        wkey = f"config{args.config}"
    else:
        per_gpu = args.batch or 1024
        if args.workload in ("long-trot", "long-bound", "long-stand"):
            # horizons beyond the reference's own gaits, up to K_MAX_GAIT_SEGMENTS = 36 (the 192-row class)
            full = workloads.make_long_horizon(per_gpu * world, args.horizon, args.workload[5:])
        else:
            mk = workloads.make_standing if args.workload == "standing" else workloads.make_trot
            full = mk(per_gpu * world, args.horizon)
        spec = {"kind": args.workload, "horizon": args.horizon, "batch": per_gpu}
        wname = f"{args.workload} (all four feet in stance)" if args.workload == "standing" else args.workload
        wkey = f"{args.workload}_h{args.horizon}"
    b = workloads.shard(full, rank, world)
    h = b["horizon"]
    b_dense = b
    if args.model == "sparse":
        from oracle import sparse_model as _SM   # (constants only: ConvexMPCLocomotion::initSparseMPC's parameters)
        b = dict(b)
        b["mu"] = _SM.SPARSE_MU
        b["weights"] = np.tile(_SM.SPARSE_WEIGHTS.astype(np.float32), (b["batch"], 1))
        spec = dict(spec, model="sparse")
        wname += " -- SPARSE formulation (SparseCMPC's model and parameters)"
        wkey += "_sparse"

    mpc = BatchedConvexMPC(dev, max_batch=per_gpu, max_horizon=max(16, h))
    max_stance = int((b["gait"] != 0).sum(1).max())
    min_stance = int((b["gait"] != 0).sum(1).min())
    if not args.no_hint:                   # (before setup: qmpc_setup allocates the pools of the classes it can reach)
        mpc.set_max_stance(max_stance)     # the caller built the contact tables, it knows their bounds
        mpc.set_min_stance(min_stance)
    mpc.setup(b["dt"], h, b["mu"], b["f_max"])
    if args.model == "sparse":
        mpc.set_robot(9.0, (0.07, 0.26, 0.242), -9.81)   # SparseCMPC.cpp:40
        mpc.set_model(1)
    mpc.set_order_hint(1 if args.order_hint == "auto" else 0)
    mpc.set_size_order(args.size_order == "on")
    if args.max_iter is not None:
        mpc.settings(max_iter=args.max_iter)
    d = mpc.upload(b)
    o = mpc.alloc_outputs(per_gpu, full=False, iters=True)
    inp, out = mpc.make_args(d, o)
    stream = torch.cuda.current_stream(dev)
    if args.caller_side:
        # commands resident in HBM; the record is rebuilt on the GPU every step
        cmd = workloads.make_commands(per_gpu, horizon=h, seed=20260928 + rank, stand_fraction=0.0, calm=True)
        dcmd = mpc.upload_command(cmd)
        rec = mpc.alloc_record(per_gpu)
        inp, out = mpc.make_args(rec, o)
        f_ff = torch.empty_like(o["grf"])
        max_stance = int(max(np.minimum(cmd["gait_durations"], h).sum(1).max(), 1))
        if not args.no_hint:
            mpc.set_max_stance(max_stance)
        mpc.set_min_stance(0)
        solve_only = mpc.solve_async

        cs = mpc.make_command_args(dcmd)

        def step_all(n, inp_, out_, stream_):
            if args.caller_side == "fused":
                mpc.solve_commands_async(n, cs, out_, f_ff, stream_)
            else:
                mpc.pack_async(dcmd, rec, stream_)
                solve_only(n, inp_, out_, stream_)
                mpc.forces_to_body_async(n, dcmd["r_body"], o["grf"], f_ff, stream_)
        if args.caller_side == "fused":   # the record is never materialised: build it once for the statistics below
            mpc.pack_async(mpc.upload_command(cmd), rec, stream)
        mpc.solve_async = step_all

    gathered = None
    if args.gather and dist is not None:
        gathered = torch.empty((world * per_gpu, 12), dtype=torch.float32, device=f"cuda:{dev}")

    def one_step():
        mpc.solve_async(per_gpu, inp, out, stream)
        if gathered is not None:
            # torch's NCCL(=RCCL) work is enqueued on its own stream and ordered after `stream`
            dist.all_gather_into_tensor(gathered, o["grf"])

    def sync_all():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # clock settle (not part of W / K): the GPU ramps its clocks over the first ~100 ms of load
    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < args.settle:
        for _ in range(50):
            mpc.solve_async(per_gpu, inp, out, stream)
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        one_step()
    sync_all()
    # ---- the timed region: EXACTLY K steps between barrier + synchronize, R times over; every repeat is a complete
    # measurement by the contract's rules (MAX over ranks), the line reports the median one.  (One 20-step region at batch
    # 1024 is 0.9 ms: a single sample of it moves by several per cent from run to run.)
    def timed_region():
        sync_all()
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record(stream)
        for _ in range(args.steps):
            one_step()
        e1.record(stream)
        sync_all()
        return time.perf_counter() - t0, e0, e1

    first_el, e0, e1 = timed_region()
    regions = [(first_el, e0.elapsed_time(e1))]
    R = max(args.repeats, 1)
    if R > 3 and first_el * R > 10.0:
        R = max(3, int(10.0 / first_el))
    if dist is not None:   # every rank must run the same number of regions
        rt = torch.tensor([R], dtype=torch.int32, device=f"cuda:{dev}")
        dist.broadcast(rt, 0)
        R = int(rt.item())
    for _ in range(R - 1):
        el, e0, e1 = timed_region()
        regions.append((el, e0.elapsed_time(e1)))   # HIP events on the launch stream
    local_el = [r[0] for r in regions]
    per_rank = None
    ev_mat = None
    if dist is not None:
        t = torch.tensor(local_el, dtype=torch.float64, device=f"cuda:{dev}")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        mat = torch.stack(allt).cpu().numpy()          # [rank][repeat]
        region_el = mat.max(0)                          # MAX over ranks, per repeat
        # the same regions by every rank's HIP events on its launch stream (first launch -> last kernel done): the barrier and
        # the host-side synchronise that bracket the contract span are OUTSIDE this one
        t = torch.tensor([r[1] for r in regions], dtype=torch.float64, device=f"cuda:{dev}")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        ev_mat = torch.stack(allt).cpu().numpy()       # [rank][repeat], ms
    else:
        mat = None
        region_el = np.array(local_el)
    order = np.argsort(region_el)
    med = int(order[len(order) // 2])                   # the median repeat (an actual measurement, not an average)
    elapsed = float(region_el[med])
    ev_ms = regions[med][1]
    if mat is not None:
        per_rank = [float(x) for x in mat[:, med]]

    def median_region(n=3):
        rs = sorted(timed_region()[0] for _ in range(n))
        return rs[len(rs) // 2]

    # ---- extra: the same K steps WITH the library's order hint on the same inputs (exact: upper bound of what it gives)
    other_order = None
    if world == 1 and not args.caller_side and not args.no_extras:
        flip = 0 if args.order_hint == "auto" else 1
        mpc.set_order_hint(flip)
        for _ in range(max(args.warmup, 2)):
            one_step()
        pr = median_region()
        other_order = {"order_hint": "auto" if flip else "off", "value": per_gpu * args.steps / pr, "unit": "QP solves/s",
                       "ms_per_step": pr / args.steps * 1e3}
        mpc.set_order_hint(1 - flip)
        for _ in range(2):
            one_step()       # (the contract state again, for the statistics read below)
        torch.cuda.synchronize(dev)

    # ---- extra: the same K steps with the size order / proxy staging switched the other way (no hint either way)
    other_size = None
    if world == 1 and not args.caller_side and not args.no_extras:
        mpc.set_size_order(args.size_order != "on")
        for _ in range(max(args.warmup, 2)):
            one_step()
        pr = median_region()
        other_size = {"size_order": "off" if args.size_order == "on" else "on", "value": per_gpu * args.steps / pr, "unit": "QP solves/s",
                      "ms_per_step": pr / args.steps * 1e3}
        mpc.set_size_order(args.size_order == "on")
        for _ in range(2):
            one_step()
        torch.cuda.synchronize(dev)

    # ---- extra: CLOSED LOOP.  The config continued as a rollout (workloads.ConfigRollout: cycle 0 is the config itself, then
    # the contact table advances one step per cycle, the state is integrated with the forces the solver returned, pushes),
    # C consecutive MPC cycles solved once to generate the records, all C input sets resident in HBM, and the timed steps walk
    # 0, 1, .., C-1, C-2, .., 1, 0, 1, .. over them: every call's "previous call" is an ADJACENT MPC cycle of the same robots,
    # which is what a controller's order hint (and warm start) really sees.  Same K steps per region, median of 5.
    closed_loop = None
    if world == 1 and args.workload == "config" and args.config > 0 and not args.caller_side and not args.no_extras and not args.no_closed_loop:
        C_ = max(4, args.cl_cycles)
        ro = workloads.ConfigRollout(b, seed=rank, periodic=(args.config != 4))
        sets, stats = [], []
        o_cl = mpc.alloc_outputs(per_gpu, full=False, iters=True)
        prev_it = None
        for c in range(C_):
            rec_c = ro.record()
            d_c = mpc.upload(rec_c)
            i_c, o_c = mpc.make_args(d_c, o_cl)
            mpc.solve_async(per_gpu, i_c, o_c, stream)
            torch.cuda.synchronize(dev)
            it_c = o_cl["iters"].cpu().numpy()
            st_c = o_cl["status"].cpu().numpy()
            corr = (float(np.corrcoef(prev_it, it_c)[0, 1]) if prev_it is not None and it_c.std() > 0 and prev_it.std() > 0 else None)
            stats.append({"cycle": c, "iters_mean": float(it_c.mean()), "iters_max": int(it_c.max()),
                          "failed": int(((st_c & 47) != 0).sum()), "corr_with_previous_cycle": corr})
            prev_it = it_c.copy()
            sets.append((d_c, i_c, o_c))
            ro.advance(o_cl["grf"].cpu().numpy())
        walk = list(range(C_)) + list(range(C_ - 2, 0, -1))

        def cl_region(K):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for k in range(K):
                _, i_c, o_c = sets[walk[k % len(walk)]]
                mpc.solve_async(per_gpu, i_c, o_c, stream)
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        K = max(args.steps, 2 * len(walk))
        res_cl = {}
        ref_grf = None
        same = True
        for mode, name in ((0, "plain_order"), (1, "previous_cycle_hint")):
            mpc.set_order_hint(mode)
            cl_region(2 * len(walk))
            ts = sorted(cl_region(K) for _ in range(5))
            res_cl[name] = {"ms_per_cycle": ts[2] / K * 1e3, "value": per_gpu * K / ts[2], "unit": "QP solves/s"}
            _, i_c, o_c = sets[C_ - 1]
            mpc.solve_async(per_gpu, i_c, o_c, stream)
            torch.cuda.synchronize(dev)
            g = o_cl["grf"].cpu().numpy().copy()
            same = same and (ref_grf is None or np.array_equal(g, ref_grf))
            ref_grf = g
        mpc.set_order_hint(1 if args.order_hint == "auto" else 0)
        closed_loop = dict(res_cl, cycles=C_, steps_per_region=K, walk="0..C-1..0 (every call follows an adjacent MPC cycle)",
                           per_cycle=stats, results_identical_plain_vs_hinted=bool(same),
                           gain=res_cl["plain_order"]["ms_per_cycle"] / res_cl["previous_cycle_hint"]["ms_per_cycle"] - 1.0,
                           note="workloads.ConfigRollout: this config continued in closed loop (contact table + 1 step per cycle, "
                                "state integrated with the solver's own forces, pushes); inputs of all cycles resident in HBM")
        del sets
        # (back to the contract inputs for the statistics read below)
        mpc.solve_async(per_gpu, inp, out, stream)
        torch.cuda.synchronize(dev)

    # ---- extra (not part of the contract fields): the same K steps with two independent
    # batches in flight on two HIP streams.  At batch 1024 a launch is exactly one round of
    # workgroups and ends with its slowest robot (13 active-set iterations vs a median of 2),
    # so back-to-back launches on ONE stream leave most of the GPU idle during every tail;
    # a second stream lets the next batch start in the slots the early finishers free.
    pipelined = None
    if world == 1 and not args.caller_side and not args.no_pipelined:
        streams = [torch.cuda.Stream(dev) for _ in range(2)]
        ctx = [(mpc, inp, out)]
        mpc2 = BatchedConvexMPC(dev, max_batch=per_gpu, max_horizon=max(16, h))
        if not args.no_hint:
            mpc2.set_max_stance(max_stance)
            mpc2.set_min_stance(min_stance)
        mpc2.setup(b["dt"], h, b["mu"], b["f_max"])
        mpc2.set_order_hint(1 if args.order_hint == "auto" else 0)
        o2 = mpc2.alloc_outputs(per_gpu, full=False, iters=True)
        inp2, out2 = mpc2.make_args(d, o2)            # same resident inputs, its own outputs
        ctx.append((mpc2, inp2, out2))
        torch.cuda.synchronize(dev)
        for k in range(2 * max(args.warmup, 2)):
            m_, i_, o_ = ctx[k % 2]
            m_.solve_async(per_gpu, i_, o_, streams[k % 2])
        torch.cuda.synchronize(dev)
        tp = time.perf_counter()
        for k in range(args.steps):
            m_, i_, o_ = ctx[k % 2]
            m_.solve_async(per_gpu, i_, o_, streams[k % 2])
        torch.cuda.synchronize(dev)
        tp = time.perf_counter() - tp
        same = bool(torch.equal(o["grf"], o2["grf"]))
        pipelined = {"streams": 2, "value": per_gpu * args.steps / tp, "unit": "QP solves/s",
                     "ms_per_step": tp / args.steps * 1e3, "results_identical_to_single_stream": same,
                     "note": "two independent batches in flight on two HIP streams (K steps total); the "
                             "contract fields above are the single-stream run"}
        mpc2.close()

    # ---- extra for N > 1 (always, not only with --gather): the same K steps once more with ONE RCCL
    # all_gather_into_tensor of the 48-byte grf rows per step, so that the driver's default N-rank command
    # exercises an RCCL data collective and its cost is visible next to the collective-free contract number
    gather_obj = None
    if dist is not None:
        g_all = gathered if gathered is not None else torch.empty((world * per_gpu, 12), dtype=torch.float32, device=f"cuda:{dev}")

        def step_gather():
            mpc.solve_async(per_gpu, inp, out, stream)
            dist.all_gather_into_tensor(g_all, o["grf"])
        for _ in range(max(args.warmup, 2)):
            step_gather()
        sync_all()
        tg = time.perf_counter()
        for _ in range(args.steps):
            step_gather()
        sync_all()
        tg = time.perf_counter() - tg
        t = torch.tensor([tg], dtype=torch.float64, device=f"cuda:{dev}")
        allt = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(allt, t)
        tg_max = max(float(x.item()) for x in allt)
        ok = bool(torch.equal(g_all[rank * per_gpu:(rank + 1) * per_gpu], o["grf"]))
        okt = torch.tensor([1 if ok else 0], dtype=torch.int32, device=f"cuda:{dev}")
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        gather_obj = {"collective": "all_gather_into_tensor (RCCL over xGMI)" if os.environ.get("QMPC_BENCH_BACKEND", "nccl") == "nccl"
                      else "all_gather_into_tensor (gloo dry run)",
                      "bytes_per_step": int(world * per_gpu * 48), "value": per_gpu * world * args.steps / tg_max,
                      "unit": "QP solves/s", "ms_per_step": tg_max / args.steps * 1e3,
                      "gathered_rows_match_local_on_every_rank": bool(int(okt.item()) == 1),
                      "note": "K steps, each followed by one all-gather of every rank's grf[shard][12] rows (max over ranks); "
                              "the contract fields are the collective-free run unless --gather was given"}

    # who took part: one line per rank (device name, PCI bus id, uuid), so that a scaling record proves N distinct GPUs
    rank_devices = None
    if dist is not None:
        pr = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device_index": dev, "device_name": pr.name,
                "pci_bus_id": "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", -1) & 0xff,
                                                  getattr(pr, "pci_device_id", 0) & 0xff),
                "uuid": str(getattr(pr, "uuid", "")), "gcn_arch": getattr(pr, "gcnArchName", ""),
                "host": os.uname().nodename}
        allo = [None] * world
        dist.all_gather_object(allo, mine)
        rank_devices = allo

    # ---- extra: the TAIL of a one-round launch.  A launch of at most one workgroup per resident slot (1024
    # robots of the 64-row class) ends with its hardest robot: fixed part + iterations x cycles per iteration.  One stamped call
    # (shader-clock stamps per phase, csrc/qmpc_kernels.hip QMPC_TICK) of the full batch and one of the tail robot ALONE on a CU
    # (a batch of 64) give the launch's longest workgroup, the floor it could reach, and how close the launch is to it
    tail = None
    if world == 1 and args.workload == "config" and not args.caller_side and not args.no_extras and per_gpu <= 1024 and h <= 16:
        try:
            mpc.set_order_hint(0)
            clk = mpc.debug_clock(per_gpu)
            mpc.solve_async(per_gpu, inp, out, stream)
            torch.cuda.synchronize(dev)
            c_ = clk.cpu().numpy().astype(np.float64)
            it_ = o["iters"].cpu().numpy()
            ok_ = c_[:, 7] > c_[:, 0]
            tot = np.where(ok_, c_[:, 7] - c_[:, 0], 0.0)
            eng = np.where(ok_, c_[:, 6] - c_[:, 5], 0.0)
            hard = int(np.argmax(np.where(ok_, it_, -1)))
            # (the shader-clock counters of different XCDs are not synchronised: only differences inside one workgroup mean
            #  anything.  The launch is as long as its longest workgroup, plus the dispatch skew of ~0.2 - 1.4 us)
            launch_cycles = float(tot.max())
            # the same robot with a CU to itself: a batch of 64 that starts at its row (rows wrap: any 64 consecutive robots)
            lo = min(hard, per_gpu - 64) if per_gpu >= 64 else 0
            nb = min(64, per_gpu)
            sub = {k: (v[lo:lo + nb] if hasattr(v, "shape") and v.dim() >= 1 and v.shape[0] == per_gpu else v) for k, v in d.items()}
            sub["batch"] = nb
            o_s = mpc.alloc_outputs(nb, full=False, iters=True)
            i_s, u_s = mpc.make_args(sub, o_s)
            clk2 = mpc.debug_clock(nb)
            for _ in range(2):
                mpc.solve_async(nb, i_s, u_s, stream)
            torch.cuda.synchronize(dev)
            c2 = clk2.cpu().numpy().astype(np.float64)
            it2 = o_s["iters"].cpu().numpy()
            hr = hard - lo
            fixed_alone = float((c2[hr, 7] - c2[hr, 0]) - (c2[hr, 6] - c2[hr, 5]))
            sel = it2 >= 3
            per_iter_alone = float(np.median((c2[sel, 6] - c2[sel, 5]) / it2[sel])) if sel.any() else None
            mpc.debug_off()
            floor = (fixed_alone + int(it_[hard]) * per_iter_alone) if per_iter_alone else None
            tail = {"launch_max_iters": int(it_[hard]), "tail_robot": hard,
                    "tail_robot_cycles_in_this_launch": float(tot[hard]), "tail_robot_fixed_part_cycles": float(tot[hard] - eng[hard]),
                    "tail_robot_active_set_cycles": float(eng[hard]),
                    "median_robot_cycles": float(np.median(tot[ok_])), "median_robot_iters": float(np.median(it_[ok_])),
                    "longest_workgroup_cycles": launch_cycles, "longest_workgroup_iters": int(it_[int(np.argmax(tot))]),
                    "kernel_us_hip_events_unstamped": ev_ms / args.steps * 1e3,
                    "implied_shader_clock_ghz": launch_cycles / (ev_ms / args.steps * 1e6),
                    "fixed_part_cycles_alone_on_a_cu": fixed_alone, "cycles_per_iteration_alone": per_iter_alone,
                    "implied_floor_cycles": floor, "achieved_over_floor": (launch_cycles / floor if floor else None),
                    "floor_over_achieved": (floor / launch_cycles if floor else None),
                    "note": "one-round launch: every robot starts at once and the launch ends with the robot that iterates longest.  Floor = "
                            "that robot's fixed part when it has a CU to itself (stamped in a 64-robot launch) + its iterations x the "
                            "cycles per iteration measured there; shader-clock cycles (s_memtime), stamped calls outside the timed regions.  "
                            "The same arithmetic reaches reference_equivalent_frac ~0.68 at 16 384 robots (profiles/r06_zz_bench_cfg1_b16384.json), "
                            "where the tail is amortised over 13 rounds"}
            mpc.set_order_hint(1 if args.order_hint == "auto" else 0)
            for _ in range(2):
                mpc.solve_async(per_gpu, inp, out, stream)   # (the contract state again, without stamps)
            torch.cuda.synchronize(dev)
        except Exception as e:
            tail = {"error": repr(e)}
            try:
                mpc.debug_off()
            except Exception:
                pass

    status = o["status"].cpu().numpy()
    iters = o["iters"].cpu().numpy()
    grf_host = o["grf"].cpu().numpy()
    nst = ((rec["gait"].cpu().numpy() if args.caller_side else b["gait"]) != 0).sum(1)
    n_fail = int(((status & 47) != 0).sum())   # QMPC_ST_ERROR_MASK
    gather_ok = None
    if gathered is not None:
        torch.cuda.synchronize(dev)
        mine = gathered[rank * per_gpu:(rank + 1) * per_gpu]
        gather_ok = bool(torch.equal(mine, o["grf"]))

    if rank == 0:
        total_qp = per_gpu * world * args.steps
        value = total_qp / elapsed
        step_ms_ev = ev_ms / args.steps
        abytes = alg_bytes_per_qp(h) * per_gpu
        ach = abytes / (step_ms_ev * 1e-3) / 1e9
        # counters of the last PMC run (tools/pmc.sh -> tools/pmc_to_latest.py), valid only for the
        # kernel source they were collected on
        traffic = executed = None
        pmc_extra = {}
        pmc_note = "no PMC entry for this workload in profiles/pmc_latest.json"
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                ent = json.load(open(pmc)).get(wkey)
                if ent and ent.get("batch") == per_gpu:
                    if ent.get("kernel_source_sha") == kernel_source_hash():
                        traffic = ent.get("hbm_bytes_per_launch")       # (per step: summed over the step's kernels)
                        executed = ent.get("fp64_flops_per_launch")
                        pmc_extra = {"pmc_measured_in_this_run": False,
                                     "pmc_fields": "traffic, executed_*, lds_bank_conflict_rate, wave_cycle_shares, rocprof: copied from "
                                                   "the committed rocprofv3 collection named in pmc_profile (same kernel source hash); "
                                                   "kernel_ms_hip_events, achieved, frac: measured in this run",
                                     "lds_bank_conflict_rate": ent.get("lds_bank_conflict_rate"),
                                     "wave_cycle_shares": ent.get("wave_cycle_shares"),
                                     "rocprof": ent.get("rocprof"), "pmc_profile": ent.get("profile")}
                        pmc_note = f"PMC counters from {ent.get('profile')} (same kernel source)"
                    else:
                        pmc_note = "profiles/pmc_latest.json was collected on a different kernel source: dropped"
            except Exception:
                pass
        nr_mean = 3.0 * nst.mean()
        flops = alg_flops_per_qp(h, nr_mean, float(iters.mean())) * per_gpu
        nr_max = 3 * int(nst.max())
        kclass = 1 if nr_max <= 64 else (4 if nr_max <= 96 else (2 if nr_max <= 128 else 3))
        # template arguments: <size class, command mode, warm start, list-consuming>; the dominant kernel of a uniform
        # workload is the first of its chain (not list-consuming) -- the name rocprofv3 --kernel-trace --stats reports
        cm = 'true' if args.caller_side == 'fused' else 'false'
        if kclass == 1 and per_gpu >= 2048 and not args.no_hint:
            kclass = 6   # the 64-row class's five-workgroups-per-CU instantiation (qmpc_set_dense, automatic from 2048 robots per handle)
        kname = f"qmpc_solve_kernel<{kclass}, {cm}, false, false>"
        if kclass in (2, 3) and os.environ.get("QMPC_NO_SPLIT", "0") != "1":
            # decoupled path: the step is the sweep kernel, the engine kernel and the (normally empty) hand-back launch
            kname = (f"qmpc_sweep_kernel<{kclass}, {cm}, false> + qmpc_engine_kernel<{kclass}, ...> "
                     f"(+ qmpc_solve_kernel<{kclass}, {cm}, false, true> on handed-back robots)")
        t_s = step_ms_ev * 1e-3
        res = {
            "metric": "convex-MPC QP solves/sec (horizon=%d, 4-leg)" % h,
            "value": value, "unit": "QP solves/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": int(len(region_el)),
            "ms_per_step_min": float(region_el.min()) / args.steps * 1e3,
            "ms_per_step_median": elapsed / args.steps * 1e3,
            "ms_per_step_max": float(region_el.max()) / args.steps * 1e3,
            "timing": "value / ms_per_step = the MEDIAN of `repeats` timed regions of exactly `steps` steps each (barrier + "
                      "synchronize on both sides, MAX over ranks per region)",
            "order_hint": {"mode_in_contract_regions": args.order_hint,
                           "what": "qmpc_set_order_hint (on by default in the library): scheduling by the iteration counts the handle's "
                                   "previous call left -- launches of several rounds take the robots hardest first, one-round launches "
                                   "keep the hard robots at the highest issue priority; results are bit-identical to the plain order (tests)",
                           "exact_in_this_bench": False if args.order_hint == "off" else True,
                           "note": "the contract fields are measured in PLAIN order: every step re-solves the same inputs, so the "
                                   "previous call's counts would be exact (hinted_same_inputs, an upper bound); what a controller gets "
                                   "-- the previous MPC cycle's counts -- is the closed_loop object",
                           ("hinted_same_inputs" if args.order_hint == "off" else "plain_order"): other_order},
            "size_order": {"mode_in_contract_regions": args.size_order,
                           "what": "qmpc_set_size_order (on by default in the library): without a usable hint, launches of several rounds take the "
                                   "robots that fit the first class largest first and, among equals, highest proxy score first (robots handed on: by "
                                   "the score, among their own places), the permutation built inside the launch; launches of one round with full CUs "
                                   "stage the sweep's issue priority per CU by the score.  Everything from THIS call's input records (contact "
                                   "tables, tracking error of the coasting state, support asymmetry): nothing from a previous solve.  Results "
                                   "are bit-identical either way (tests)",
                           ("off_same_inputs" if args.size_order == "on" else "on_same_inputs"): other_size},
            "closed_loop": closed_loop,
            "tail": tail,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"caller-side pipeline ({args.caller_side}: command -> record -> solve -> body-frame forces), " if args.caller_side else "") +
                                   f"{wname}: batch={per_gpu} robots/GPU, "
                                   f"horizon={h}, mean reduced QP size {nr_mean:.1f} vars",
                       "batch_per_gpu": per_gpu, "horizon": h, "sharding": f"independent robots x{world}",
                       "mean_active_set_iters": float(iters.mean()), "max_active_set_iters": int(iters.max()),
                       "failed": n_fail, "max_stance_hint": (0 if args.no_hint else max_stance),
                       "result_gather": ("rccl all_gather_into_tensor of grf per step" if gathered is not None else "none")},
            # The path is compute-shaped, not HBM-shaped: the binding roof is the fp64
            # VALU rate.  The kernel issues vector fp64 (DPP fmac), no MFMA; on MI355X the dense fp64
            # MFMA peak is the same 78.6 TFLOP/s.
            "roofline": dict({"bound": "fp64_valu",
                         # primary figure: what the kernel EXECUTES (fp64 wave-instructions from the PMC counters of this
                         # kernel source) over the HIP-event time; without counters for this source: the SURVEY 8d figure
                         "achieved": (executed if executed else flops) / t_s / 1e12,
                         "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": (executed if executed else flops) / t_s / 1e12 / FP64_PEAK_TFLOPS,
                         "frac_is": "executed" if executed else "reference_equivalent (no PMC counters for this kernel source)",
                         "traffic": traffic,
                         "kernel": kname, "kernel_ms_hip_events": step_ms_ev,
                         "executed_flops_per_qp": (executed / per_gpu if executed else None),
                         "executed_tflops": (executed / t_s / 1e12 if executed else None),
                         "executed_frac": (executed / t_s / 1e12 / FP64_PEAK_TFLOPS if executed else None),
                         # SURVEY.md 8d's ALGORITHMIC flops F_alg(h, n_r, K) -- what the reference's dense formulation needs
                         "reference_equivalent_flops_per_qp": flops / per_gpu,
                         "reference_equivalent_tflops": flops / t_s / 1e12,
                         "reference_equivalent_frac": flops / t_s / 1e12 / FP64_PEAK_TFLOPS,
                         "note": "frac = executed fp64 flops (64 lanes x (2 FMA + MUL + ADD) wave-instructions, PMC) over the "
                                 "HIP-event kernel time against the 78.6 TFLOP/s fp64 peak; reference_equivalent_*: SURVEY.md 8d's "
                                 "algorithmic F_alg(h, n_r, K), 84 % of which is the dense condensation the closed-form assembly "
                                 "never executes. " + pmc_note}, **pmc_extra),
            "roofline_hbm": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                             # what the memory system really moved (PMC bytes per step over the HIP-event time): the large-problem
                             # producer streams its 1.5 MB work items through HBM once per block step and IS bound here
                             "traffic_rate_gbs": (traffic / t_s / 1e9 if traffic else None),
                             "traffic_frac_of_peak": (traffic / t_s / 1e9 / HBM_PEAK_GBS if traffic else None),
                             "alg_bytes_per_qp": alg_bytes_per_qp(h),
                             "note": "728 B in + 48 B out per robot at h=10: tiny by construction"},
        }
        if nr_max > 192:
            # the large-problem path (192 < n_r <= 432): its producer streams the robot's 1.5 MB work item through HBM once per
            # block step -- THAT is the roof it runs against; the fp64 figures (vector instructions only:
            # the rank-16 updates run on the matrix cores and are not in those counters) move to roofline_fp64_valu
            res["roofline_fp64_valu"] = res["roofline"]
            rate = (traffic / t_s / 1e9) if traffic else ach
            res["roofline"] = {"bound": "hbm", "achieved": rate, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rate / HBM_PEAK_GBS,
                               "frac_is": ("measured HBM traffic (PMC FETCH_SIZE x 2 + WRITE_SIZE per step) over the HIP-event time" if traffic
                                           else "algorithmic bytes (no PMC counters for this kernel source)"),
                               "traffic": traffic, "alg_bytes_per_step": abytes,
                               "kernel": "qmpc_big_kernel<%s> (+ qmpc_engine_kernel<7, ...>)" % cm, "kernel_ms_hip_events": step_ms_ev,
                               "pmc_measured_in_this_run": False if traffic else None, "pmc_profile": pmc_extra.get("pmc_profile"),
                               "note": "traffic well above the algorithmic bytes is inherent to inverting a 432 x 432 matrix that fits no "
                                       "on-chip memory: 27 block steps x the lower triangle read and written once. " + pmc_note}
        if rank_devices is not None:
            res["rccl_world_size"] = world if os.environ.get("QMPC_BENCH_BACKEND", "nccl") == "nccl" else None
            res["world_size"] = world
            res["backend"] = os.environ.get("QMPC_BENCH_BACKEND", "nccl")
            res["rank_devices"] = rank_devices
            res["distinct_devices"] = len({(r.get("pci_bus_id"), r.get("uuid")) for r in rank_devices})
        if per_rank is not None:
            res["per_rank"] = {"elapsed_s": per_rank,
                               "qp_per_s": [per_gpu * args.steps / t for t in per_rank],
                               "gathered_rows_match_local": gather_ok}
        if ev_mat is not None:
            # N > 1: every timed region of the contract ends with ONE collective barrier after the stream work (sync_all); at
            # configs[1] size a 20-step region is 0.9 ms, so a 30 - 60 us 8-rank barrier reads as 3 - 7 % "sub-linear scaling".
            # value_event_timed takes the same median region by the ranks' own HIP events (MAX over ranks): barrier outside the span
            ev_max_ms = float(ev_mat[:, med].max())
            res["value_event_timed"] = per_gpu * world * args.steps / (ev_max_ms * 1e-3)
            res["event_timed"] = {"unit": "QP solves/s", "ms_per_step": ev_max_ms / args.steps,
                                  "per_rank_ms_per_step": [float(x) / args.steps for x in ev_mat[:, med]],
                                  "barrier_bias_measured": float(1.0 - ev_max_ms * 1e-3 / elapsed),
                                  "barrier_bias_expected": "one collective barrier + host synchronise per timed region of `steps` steps: "
                                                           "~30 - 60 us at 8 ranks = %.1f - %.1f %% of this region" % (
                                                               3e-5 / elapsed * 100.0, 6e-5 / elapsed * 100.0),
                                  "note": "value (the contract field) = robots x steps / MAX over ranks of the host-timed region incl. the "
                                          "barrier; value_event_timed = the same region by HIP events on each rank's launch stream, MAX over "
                                          "ranks -- the number to read scaling efficiency from when regions are short"}
        if gather_obj is not None:
            res["gather"] = gather_obj
        if pipelined is not None:
            res["pipelined"] = pipelined
        if not args.no_cpu_baseline and world == 1 and args.model == "sparse":
            # north_star: "the reference qpOASES/osqp path timed on the same box": the OSQP leg is THIS model's reference path;
            # the dense qpOASES pipeline on the same states stands beside it
            res["cpu_baseline"] = cpu_baseline_sparse(b, spec, all_cores=not args.no_cpu_all_cores, gpu_grf=grf_host)
            res["cpu_baseline_dense_qpoases"] = cpu_baseline(b_dense, {k: v for k, v in spec.items() if k != "model"},
                                                             all_cores=not args.no_cpu_all_cores, gpu_grf=None)
        elif not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(b, spec, all_cores=not args.no_cpu_all_cores,
                                               gpu_grf=None if args.caller_side else grf_host)
            ps = (res["cpu_baseline"] or {}).get("parity_sample")
            if ps and "max_rel_grf_err" in ps:
                # the fraction of robots over north_star's flat 1e-4 and the maximum, next to the workload they belong to
                res["config"]["parity_sample"] = {k: ps[k] for k in ("robots", "whole_shard", "reference_hit_nwsr_cap", "max_rel_grf_err", "frac_over_1e-4",
                                                                     "robots_over_1e-4", "max_err_over_reference_spread",
                                                                     "worst_checked_within_1.5x_reference_spread", "spread_checked_on",
                                                                     "spread_unchecked", "all_over_1e-4_checked")}
        print(json.dumps(res))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
