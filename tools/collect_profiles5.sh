#!/bin/bash
# gpurun_out/<tag>/ (tools/profile_round5.sh) -> profiles/<tag>_*  and profiles/pmc_latest.json
set -eu
T=$1
cd "$(dirname "$0")/.."
for f in gpurun_out/$T/bench_*.json gpurun_out/$T/kernel_stats_*.csv gpurun_out/$T/pmc_summary_*.json gpurun_out/$T/shim_latency.json; do
  [ -s "$f" ] && cp "$f" profiles/${T}_$(basename "$f")
done
cp gpurun_out/$T/pmc_latest.json profiles/pmc_latest.json
python - "$T" <<'PY'
import glob, json, os, sys
T = sys.argv[1]
for f in sorted(glob.glob(f"profiles/{T}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    r = d["roofline"]; cl = d.get("closed_loop") or {}
    hs = (d.get("order_hint") or {}).get("hinted_same_inputs") or {}
    print(f"{os.path.basename(f)[len(T)+7:-5]:22s} value {d['value']:.4g}  hinted-same-inputs {hs.get('value', float('nan')):.4g}  closed loop plain/hint "
          f"{(cl.get('plain_order') or {}).get('value', float('nan')):.4g}/{(cl.get('previous_cycle_hint') or {}).get('value', float('nan')):.4g}  "
          f"frac {r['frac']:.3f} ({r['frac_is'][:8]})  ref-equiv {r.get('reference_equivalent_frac', float('nan')):.3f}  kernel ms {r.get('kernel_ms_hip_events', float('nan')):.4f}  "
          f"parity {(d['config'].get('parity_sample') or {}).get('max_rel_grf_err')}")
PY
