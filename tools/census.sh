#!/bin/bash
# Per-stage instruction census of the one-kernel path, on the GPU box:
#     gpurun -- 'bash tools/census.sh r06_e'
# needs the census builds variants/stop{m1,0,1,2,3,4}/libqmpc.so (tools/census_build.sh, on the build host) next to the
# production library.  For the 64-row class as configs[1] runs it (class 1, 1024 robots) and as configs[2] runs it (class 6,
# 4096 robots): three PMC passes per build (instruction mix, fp64 mix, cycles), then the production library with the
# iteration cap at 0 / 1 / 2 / 3 / 1000 (one engine iteration = the slope).  tools/census_table.py turns the summaries into the table.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name, lib ("" = production), bench args...
  local name=$1 lib=$2; shift 2
  local i=0
  for grp in \
    "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES" \
    "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" \
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC" ; do
    i=$((i+1))
    QMPC_LIB=${lib:+$R/variants/$lib/libqmpc.so} rocprofv3 --kernel-trace --pmc $grp -d $OUT/$name/p$i -o p --output-format csv -- \
      python $R/bench.py --steps 10 --warmup 3 --settle 0 --repeats 1 --no-cpu-baseline --no-extras --no-pipelined "$@" > $OUT/$name.p$i.log 2>&1
  done
  python $R/tools/pmc_summary.py $OUT/$name > /dev/null 2>&1
  cp $OUT/$name/pmc_summary.json $OUT/census_$name.json 2>/dev/null
  rm -rf $OUT/$name
}
for w in "c1 --config 1" "c2 --config 2"; do
  set -- $w; key=$1; shift
  for v in stopm1 stop0 stop1 stop2 stop3 stop4; do pass ${key}_$v $v "$@"; done
  for m in 0 1 2 3 1000; do pass ${key}_iter$m "" "$@" --max-iter $m; done
done
python $R/tools/census_table.py $OUT > $OUT/census_table.md 2>&1
cat $OUT/census_table.md
