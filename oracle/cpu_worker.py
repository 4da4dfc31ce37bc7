"""One worker of the all-cores CPU baseline -- TEST / BENCH INFRASTRUCTURE ONLY.

    python -m oracle.cpu_worker <workload-json> <seconds> [cpu]

Runs the reference-style pipeline (C restatement of the SolverMPC.cpp assembly + the
reference's own qpOASES, oracle/_ref) over its robots in a loop for about <seconds> seconds of
wall time and prints {"solved": n, "elapsed": s}.  bench.py starts one of these per PHYSICAL host core, pinned to it ([cpu])
(the reference itself is single-threaded and non-reentrant: file-scope globals,
convexMPC_interface.cpp:13-20; one process per core is how a user would scale it).
"""
import json
import sys
import time

import numpy as np


def make_workload(spec):
    from quadruped_ctrl_amd import workloads as W
    kind = spec["kind"]
    if kind == "config":
        return W.make_config(spec["config"], batch=spec["batch"])
    if kind == "standing":
        return W.make_standing(spec["batch"], spec["horizon"])
    if kind == "trot":
        return W.make_trot(spec["batch"], spec["horizon"])
    raise ValueError(kind)


def main():
    spec = json.loads(sys.argv[1])
    seconds = float(sys.argv[2])
    if len(sys.argv) > 3:
        try:
            import os
            os.sched_setaffinity(0, {int(sys.argv[3])})
        except (AttributeError, OSError, ValueError):
            pass
    from oracle import oracle as O
    b = make_workload(spec)
    arr = O.pack_updates(b)
    O.solve_packed(arr, b)                      # warm (page in, malloc arenas)
    solved = 0
    t0 = time.perf_counter()
    while True:
        O.solve_packed(arr, b)
        solved += b["batch"]
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    print(json.dumps({"solved": solved, "elapsed": el}))


if __name__ == "__main__":
    main()
