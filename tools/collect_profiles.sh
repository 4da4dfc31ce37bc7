#!/bin/bash
# Copy one tools/profile_round.sh output set from gpurun_out/<tag>/ into profiles/<tag>_* and restamp
# profiles/pmc_latest.json:   bash tools/collect_profiles.sh r02_z
set -eu
T=$1
cd "$(dirname "$0")/.."
for n in cfg1 standing_h10 standing_h14 standing_h16 trot_h16; do
  cp gpurun_out/$T/bench_$n.json profiles/${T}_bench_$n.json
  cp gpurun_out/$T/kernel_stats_$n.csv profiles/${T}_kernel_stats_$n.csv
  cp gpurun_out/$T/pmc_summary_$n.json profiles/${T}_pmc_summary_$n.json
done
for n in cfg2 cfg4; do   # (round 4: counters and kernel stats for configs[2] / [4] as well)
  for f in bench_$n.json kernel_stats_$n.csv pmc_summary_$n.json; do [ -f gpurun_out/$T/$f ] && cp gpurun_out/$T/$f profiles/${T}_$f; done
done
cp gpurun_out/$T/shim_latency.json profiles/${T}_shim_latency.json
cp gpurun_out/$T/warm_rollout.json profiles/${T}_warm_rollout.json
cp gpurun_out/$T/class_stats.txt profiles/${T}_class_stats.txt 2>/dev/null || true
for f in large_trot_h36 large_stand_h36; do cp gpurun_out/$T/kernel_stats_$f.csv profiles/${T}_kernel_stats_$f.csv 2>/dev/null || true; cp gpurun_out/$T/pmc_summary_$f.json profiles/${T}_pmc_summary_$f.json 2>/dev/null || true; done
for f in split_check engine_phases sweep_phases jcqp_long dense_threshold; do cp gpurun_out/$T/$f.txt profiles/${T}_$f.txt 2>/dev/null || true; done
for f in long_trot_h24 long_bound_h36 large_trot_h36 large_standing_h24 large_stand_h36 standing_h10_one_kernel standing_h14_one_kernel standing_h16_one_kernel; do
  cp gpurun_out/$T/bench_$f.json profiles/${T}_bench_$f.json 2>/dev/null || true
done
python - "$T" <<'PY'
import glob, json, os, sys
T = sys.argv[1]
out = {}
for f in sorted(glob.glob(f"gpurun_out/{T}/bench_cfg*.json") + glob.glob(f"gpurun_out/{T}/bench_caller*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    k = os.path.basename(f)[6:-5]
    out[k] = {kk: d[kk] for kk in ("value", "ms_per_step", "config") if kk in d}
    if "pipelined" in d:
        out[k]["pipelined"] = d["pipelined"]["value"]
json.dump(out, open(f"profiles/{T}_bench_other_configs.json", "w"), indent=1)
for n in ("cfg1", "standing_h10", "standing_h14", "standing_h16", "trot_h16"):
    d = json.loads(open(f"profiles/{T}_bench_{n}.json").read().strip().splitlines()[-1])
    c = d.get("cpu_baseline") or {}
    r = d["roofline"]
    print(n, "QP/s %.3g" % d["value"], "kernel ms %.4f" % r["kernel_ms_hip_events"], "alg frac %.3f" % r["frac"],
          "iters %.1f/%d" % (d["config"]["mean_active_set_iters"], d["config"]["max_active_set_iters"]),
          "fail", d["config"]["failed"], "pipelined %.3g" % d["pipelined"]["value"], "cpu1 %.0f" % (c.get("value") or 0),
          "cpuall", (c.get("all_cores") or {}).get("value"))
PY
rm -f profiles/pmc_latest.json
for pair in "cfg1 config1 1024" "standing_h10 standing_h10 1024" "standing_h14 standing_h14 1024" "standing_h16 standing_h16 1024" "trot_h16 config3 4096"; do
  set -- $pair
  python tools/pmc_to_latest.py profiles/${T}_pmc_summary_$1.json $2 $3 profiles/${T}_pmc_summary_$1.json profiles/${T}_kernel_stats_$1.csv > /dev/null
done
for pair in "large_stand_h36 long-stand_h36 1024" "large_trot_h36 long-trot_h36 1024" "cfg2 config2 4096" "cfg4 config4 8192"; do
  set -- $pair
  [ -f profiles/${T}_pmc_summary_$1.json ] && python tools/pmc_to_latest.py profiles/${T}_pmc_summary_$1.json $2 $3 profiles/${T}_pmc_summary_$1.json profiles/${T}_kernel_stats_$1.csv > /dev/null
done
python -c "
import json; d=json.load(open('profiles/pmc_latest.json')); [print(k, int(v['hbm_bytes_per_launch']), '%.3g'%v['fp64_flops_per_launch'], v['kernel_source_sha'], {a:round(b,2) for a,b in v['wave_cycle_shares'].items()}) for k,v in d.items()]"
