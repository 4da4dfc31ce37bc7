#!/bin/bash
# Per-kernel register / spill / scratch / occupancy table of qmpc_kernels.hip (compile only, no GPU needed).
# usage: tools/kernel_resources.sh [class ...]   (default: all four, compiled in parallel)
R=$(cd "$(dirname "$0")/.." && pwd)
CLASSES=${@:-1 6 4 2 3}
for rb in $CLASSES; do
  (cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value -DQMPC_RB=$rb -c $R/quadruped_ctrl_amd/csrc/qmpc_kernels.hip \
    -Rpass-analysis=kernel-resource-usage -o /tmp/qmpc_k$rb.o 2>&1 \
    | grep -E "Function Name|  VGPRs:|SGPRs Spill|ScratchSize|Occupancy" | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass.*//' | paste - - - - - \
    | sed -e 's/Function Name: //' | sort > /tmp/qmpc_res$rb.txt) &
done
wait
for rb in $CLASSES; do cat /tmp/qmpc_res$rb.txt; done
