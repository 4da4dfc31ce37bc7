#!/bin/bash
# per-kernel durations of the decoupled path on one workload:  bash tools/split_prof.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/st -o s --output-format csv -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-pipelined "$@" > $OUT/bench.log 2>&1
find $OUT/st -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
rm -rf $OUT/st
cat $OUT/kernel_stats.csv | cut -c1-160
tail -1 $OUT/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['mean_active_set_iters'], d['config']['max_active_set_iters'], d['config']['failed'])"
