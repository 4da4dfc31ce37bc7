"""tools/census.sh's summaries -> the per-stage instruction census (markdown).  Wave-instructions per ROBOT (all waves of its
workgroup), differences of PMC counters between consecutive census builds; one engine iteration = the slope over the
iteration cap."""
import glob
import json
import os
import sys

out = sys.argv[1]
CNT = ["SQ_INSTS_VALU", "FP64", "NONFP64", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR",
       "SQ_INSTS_BRANCH", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU"]
HEAD = ["VALU", "fp64 (FMA+MUL+ADD)", "other VALU", "SALU", "LDS", "SMEM", "VMEM rd", "VMEM wr", "branch", "wave-cycles x4", "waiting x4",
        "VALU busy x4"]


def load(name, cls):
    f = os.path.join(out, f"census_{name}.json")
    if not os.path.exists(f):
        return None
    d = json.load(open(f))
    for k, v in d.items():
        if k.startswith(f"void qmpc_solve_kernel<{cls},"):
            v = dict(v)
            v["FP64"] = v.get("SQ_INSTS_VALU_FMA_F64", 0) + v.get("SQ_INSTS_VALU_MUL_F64", 0) + v.get("SQ_INSTS_VALU_ADD_F64", 0)
            v["NONFP64"] = v.get("SQ_INSTS_VALU", 0) - v["FP64"]
            return v
    return None


def bench_line(name):
    for f in sorted(glob.glob(os.path.join(out, f"{name}.p*.log"))):
        for line in open(f, errors="replace"):
            if line.startswith("{"):
                try:
                    return json.loads(line)
                except ValueError:
                    pass
    return None


for key, cls, batch, title in (("c1", 1, 1024, "configs[1]: 1024 robots, trot, the four-per-CU instantiation (class 1)"),
                               ("c2", 6, 4096, "configs[2]: 4096 robots, mixed gaits, the five-per-CU instantiation (class 6)")):
    st = {v: load(f"{key}_{v}", cls) for v in ("stopm1", "stop0", "stop1", "stop2", "stop3", "stop4")}
    it = {m: load(f"{key}_iter{m}", cls) for m in (0, 1, 2, 3, 1000)}
    itn = {}
    for m in it:
        b = bench_line(f"{key}_iter{m}")
        itn[m] = b["config"]["mean_active_set_iters"] if b else None
    if any(v is None for v in st.values()) or it[1000] is None:
        print(f"### {title}\n\nincomplete: {[k for k, v in st.items() if v is None]} {[m for m, v in it.items() if v is None]}\n")
        continue
    rows = []

    def diff(a, b):
        return [(a.get(c, 0) - (b.get(c, 0) if b else 0)) / batch for c in CNT]
    rows.append(("census dump + kernel entry (subtracted below)", diff(st["stopm1"], None)))
    rows.append(("stage 0: loads, stance list, M_b / N_b, error rows", diff(st["stop0"], st["stopm1"])))
    rows.append(("stage 1: E_00 / E_11, moment scan", diff(st["stop1"], st["stop0"])))
    rows.append(("stage 2: g, H rows into registers", diff(st["stop2"], st["stop1"])))
    rows.append(("stage 3: Gauss-Jordan sweep", diff(st["stop3"], st["stop2"])))
    rows.append(("stage 4: x_u, packed inverse -> LDS, diag", diff(st["stop4"], st["stop3"])))
    full = it[1000]
    # engine: the full kernel minus (stop4 - dump); first search + outputs = cap 0 minus that
    base4 = [(st["stop4"].get(c, 0) - st["stopm1"].get(c, 0)) / batch for c in CNT]
    kernel_entry = [0.0] * len(CNT)
    if it[0] is not None:
        rows.append(("stage 5 at cap 0: first search, outputs", [it[0].get(c, 0) / batch - b for c, b in zip(CNT, base4)]))
    ms = [m for m in (0, 1, 2, 3) if it[m] is not None and itn[m] is not None]
    if len(ms) >= 2:
        lo, hi = ms[0], ms[-1]
        dn = (itn[hi] - itn[lo])
        rows.append((f"one engine iteration (slope, cap {lo} -> {hi}: {dn:.2f} iterations / robot)",
                     [(it[hi].get(c, 0) - it[lo].get(c, 0)) / batch / dn for c in CNT]))
    rows.append((f"whole kernel ({itn[1000]:.2f} iterations / robot)", [full.get(c, 0) / batch for c in CNT]))
    print(f"### {title}\n")
    print("| per robot (wave-instructions, all waves) | " + " | ".join(HEAD) + " |")
    print("|---|" + "---|" * len(HEAD))
    for name, v in rows:
        print(f"| {name} | " + " | ".join(f"{x:,.0f}" for x in v) + " |")
    print()
