#!/bin/bash
# PMC passes for the solve kernel (run on the GPU box via gpurun).  Each counter
# group in its own rocprofv3 run (--kernel-trace only, as gpurun requires).
# usage: tools/pmc.sh <outdir-under-gpurun_out> [bench args...]
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU" \
  "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE" \
  "FETCH_SIZE" \
  "WRITE_SIZE" \
  "GRBM_GUI_ACTIVE" ; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p --output-format csv -- python $R/bench.py --steps 20 --warmup 5 --settle 0 --repeats 1 --no-cpu-baseline --no-extras "$@" > $OUT/p$i.log 2>&1
done
python $R/tools/pmc_summary.py $OUT
