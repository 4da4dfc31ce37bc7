#!/bin/bash
# Quick before/after set (GPU box): bench lines of the BASELINE configs and the standing / large workloads, no CPU baseline.
# usage: gpurun -- 'bash tools/quick_bench.sh <tag> [extra bench args]'
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
q() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    p=d.get("pipelined") or {}
    print(f"{sys.argv[1].split('/')[-1]:34s} {d['value']:.4e} QP/s  ms/step med {d['ms_per_step']:.4f} min {d.get('ms_per_step_min',0):.4f} max {d.get('ms_per_step_max',0):.4f} R={d.get('repeats')} iters {d['config']['mean_active_set_iters']:.2f}/{d['config']['max_active_set_iters']} fail {d['config']['failed']} 2-stream {p.get('value',0):.3e}")
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
}
b() { local n=$1; shift; python $R/bench.py --no-cpu-baseline "$@" > $OUT/$n.json 2> $OUT/$n.err; q $OUT/$n.json; }
b cfg1 --steps 200 "$@"
b cfg1_b16384 --steps 50 --batch 16384 --no-pipelined "$@"
b cfg2 --steps 100 --config 2 "$@"
b cfg3 --steps 50 --config 3 "$@"
b cfg4 --steps 50 --config 4 "$@"
b standing_h10 --steps 100 --workload standing --horizon 10 "$@"
b standing_h14 --steps 50 --workload standing --horizon 14 "$@"
b large_stand_h36 --steps 10 --warmup 2 --workload long-stand --horizon 36 --no-pipelined "$@"
b large_trot_h36 --steps 20 --warmup 2 --workload long-trot --horizon 36 --no-pipelined "$@"
