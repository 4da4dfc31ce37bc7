#!/bin/bash
# instruction-cache counters per kernel:  bash tools/icache_pmc.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH SQ_INSTS_VALU SQ_BUSY_CYCLES -d $OUT/p -o p --output-format csv -- python $R/bench.py --steps 10 --warmup 3 --settle 0 --no-cpu-baseline --no-pipelined "$@" > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, d in acc.items():
    print(k, n[k], {c: round(v / max(n[k], 1)) for c, v in d.items()})
PY
tail -3 $OUT/log.txt | cut -c1-300
