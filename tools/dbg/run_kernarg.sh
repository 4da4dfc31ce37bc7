#!/bin/bash
# where do the kernel arguments live?  the contract loop of configs[1] with the kernarg segment forced to device / host memory
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for v in unset 0 1; do
  for k in "--steps 200" "--steps 100 --config 2"; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    python $R/bench.py --no-cpu-baseline --repeats 11 $k 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(f'HIP_FORCE_DEV_KERNARG=$v $k: {d[\"value\"]:.4e} QP/s  {d[\"ms_per_step\"]:.4f} ms (min {d[\"ms_per_step_min\"]:.4f})')
"
  done
done
