#!/bin/bash
# Bench a list of variant builds (variants/<name>/libqmpc.so) on a few workloads.  usage: tools/variant_bench.sh "<names>" [workload keys]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
NAMES=$1; shift
KEYS=${@:-cfg1 cfg3_512 cfg3 cfg4 s10 s14}
declare -A ARGS
ARGS[cfg1]="--steps 200"
ARGS[cfg1_16k]="--steps 50 --batch 16384 --no-pipelined"
ARGS[cfg2]="--steps 100 --config 2"
ARGS[cfg3_512]="--steps 100 --config 3 --batch 512 --no-pipelined"
ARGS[cfg3]="--steps 50 --config 3 --no-pipelined"
ARGS[cfg4]="--steps 50 --config 4 --no-pipelined"
ARGS[s10]="--steps 100 --workload standing --horizon 10 --no-pipelined"
ARGS[s14]="--steps 50 --workload standing --horizon 14 --no-pipelined"
ARGS[l36]="--steps 10 --warmup 2 --workload long-stand --horizon 36 --no-pipelined"
for k in $KEYS; do
  for n in $NAMES; do
    QMPC_LIB=$R/variants/$n/libqmpc.so python $R/bench.py --no-cpu-baseline --repeats 11 ${ARGS[$k]} 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(f'$k $n: {d[\"value\"]:.4e} QP/s  {d[\"ms_per_step\"]:.4f} ms (min {d[\"ms_per_step_min\"]:.4f}) fail {d[\"config\"][\"failed\"]}')
"
  done
done
