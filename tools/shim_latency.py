#!/usr/bin/env python
"""Batch-of-one latency through the reference's six-symbol interface (libconvexmpc_shim.so)
next to the reference-style CPU pipeline (oracle assembly + the reference's qpOASES) on the
same robots.  Run on the GPU box:  python tools/shim_latency.py > gpurun_out/shim_latency.json"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from quadruped_ctrl_amd import workloads as W  # noqa: E402


def write_records(b, path):
    h, B = b["horizon"], b["batch"]
    with open(path, "wb") as f:
        f.write(np.array([h, B], np.int32).tobytes())
        f.write(np.array([b["dt"], b["mu"], b["f_max"]], np.float32).tobytes())
        for i in range(B):
            rec = np.concatenate([b["p"][i], b["v"][i], b["q"][i], b["w"][i], b["r"][i], [b["yaw"][i]],
                                  b["weights"][i], b["traj"][i], [b["alpha"][i]]]).astype(np.float32)
            f.write(rec.tobytes())
            f.write(b["gait"][i].astype(np.int32).tobytes())


def main():
    exe = os.path.join(ROOT, "tools", "shim_latency")
    pkg = os.path.join(ROOT, "quadruped_ctrl_amd")
    subprocess.run(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tools", "shim_latency.cpp"),
                    "-I" + os.path.join(ROOT, "include"), "-L" + pkg, "-lconvexmpc_shim", "-lqmpc",
                    "-Wl,-rpath," + pkg, "-o", exe], check=True)
    out = []
    fams = [("trot_h10", W.make_config(1, batch=64)), ("trot_h14", W.make_trot(64, 14)),
            ("trot_h16", W.make_config(3, batch=64)), ("standing_h10", W.make_standing(64, 10)),
            ("standing_h14", W.make_standing(32, 14)), ("standing_h16", W.make_standing(32, 16))]
    for name, b in fams:
        path = f"/tmp/shim_{name}.bin"
        write_records(b, path)
        r = subprocess.run([exe, path, "2000"], capture_output=True, text=True, check=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        d["family"] = name
        try:
            from oracle import oracle as O
            arr = O.pack_updates(b)
            O.solve_packed(arr, b)
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                O.solve_packed(arr, b)
            d["cpu_reference_style_us_per_solve"] = (time.perf_counter() - t0) / (reps * b["batch"]) * 1e6
        except Exception as e:
            d["cpu_error"] = repr(e)
        out.append(d)
    print(json.dumps({"shim_latency": out}, indent=1))


if __name__ == "__main__":
    main()
