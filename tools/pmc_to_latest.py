"""Fold a tools/pmc.sh run into profiles/pmc_latest.json (what bench.py's roofline.traffic and
roofline.executed_* read), stamped with the hash of the kernel source it was collected on.

    python tools/pmc_to_latest.py <pmc_summary.json | its directory> <workload-key> <batch> <profile-name> [kernel_stats.csv]
e.g. python tools/pmc_to_latest.py gpurun_out/r02_pmc_cfg1 config1 1024 profiles/r02_a_pmc_summary.json profiles/r02_a_kernel_stats.csv
(the optional rocprofv3 --kernel-trace --stats csv of the same command: the dominant kernel's average duration is
recorded next to the counters, with the file name, so that the bench line can cite it)
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash  # noqa: E402

src, key, batch, prof = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
summ = json.load(open(src if os.path.isfile(src) else os.path.join(src, "pmc_summary.json")))
# every solve-path kernel of a step (one-kernel classes: qmpc_solve_kernel; decoupled classes: qmpc_sweep_kernel +
# qmpc_engine_kernel + the hand-back launch): counters are per dispatch, a step dispatches each of them once, so the
# per-step figure is the sum weighted by dispatches / steps; `name` = the one with the most wave cycles
ks = {k: v for k, v in summ.items() if "qmpc_" in k and ("solve_kernel" in k or "sweep_kernel" in k or "engine_kernel" in k or "admm_kernel" in k or "big_kernel" in k)}
name = max(ks, key=lambda k: ks[k].get("SQ_WAVE_CYCLES", 0) * ks[k].get("dispatches", 1))
steps = ks[name].get("dispatches", 1)   # (the dominant kernel runs once per step; a hand-back kernel may run twice)
d = {}
for k, v in ks.items():
    wgt = v.get("dispatches", steps) / steps
    for c, x in v.items():
        if c != "dispatches" and isinstance(x, (int, float)):
            d[c] = d.get(c, 0.0) + wgt * x
fetch, write = d.get("FETCH_SIZE", 0.0), d.get("WRITE_SIZE", 0.0)     # KiB per step
flops = 64.0 * (2.0 * d.get("SQ_INSTS_VALU_FMA_F64", 0.0) + d.get("SQ_INSTS_VALU_MUL_F64", 0.0) + d.get("SQ_INSTS_VALU_ADD_F64", 0.0))
path = os.path.join(ROOT, "profiles", "pmc_latest.json")
latest = json.load(open(path)) if os.path.exists(path) else {}
latest[key] = {
    "kernel": name, "kernels": sorted(ks), "batch": batch, "profile": prof, "kernel_source_sha": kernel_source_hash(),
    # MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are in KiB... on gfx950 the read side is
    # under-reported by 2x for 64-byte requests: doubled here (upper bound for this narrow-access kernel)
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
    "fetch_size_kib": fetch, "write_size_kib": write,
    "fp64_flops_per_launch": flops,
    "fp64_wave_insts": {k: d.get(k) for k in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU")},
    "wave_cycle_shares": {k: (d.get(k, 0.0) / d["SQ_WAVE_CYCLES"] if d.get("SQ_WAVE_CYCLES") else None)
                          for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY")},
}
ent = latest[key]
# LDS bank-conflict rate of the solver kernel = conflict cycles / LDS-active cycles
ent["lds_bank_conflict_rate"] = (d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"]
                                 if d.get("SQ_LDS_IDX_ACTIVE") else None)
if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
    rows = [r for r in csv.DictReader(open(sys.argv[5])) if "qmpc_" in r["Name"]]
    if rows:
        top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
        nstep = max(int(r["Calls"]) for r in rows)
        ent["rocprof"] = {"file": os.path.relpath(sys.argv[5], ROOT) if os.path.isabs(sys.argv[5]) else sys.argv[5],
                          "kernel": top["Name"], "calls": int(top["Calls"]), "avg_us": float(top["AverageNs"]) / 1e3,
                          "all_qmpc_kernels_us_per_step": sum(float(r["TotalDurationNs"]) for r in rows) / nstep / 1e3,
                          "all_qmpc_kernels_avg_us": {r["Name"]: float(r["AverageNs"]) / 1e3 for r in rows}}
json.dump(latest, open(path, "w"), indent=1)
print(json.dumps(latest[key], indent=1))
