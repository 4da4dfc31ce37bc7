#!/bin/bash
# The census builds of tools/census.sh: variants/stop{m1,0,1,2,3,4}/libqmpc.so (the kernel leaves right after stage k).
R=$(cd "$(dirname "$0")/.." && pwd)
for k in -1 0 1 2 3 4; do
  n=stop${k/-/m}
  bash $R/tools/build_variant.sh $n "-DQMPC_STOP_AFTER=$k" || exit 1
done
ls -la $R/variants/*/libqmpc.so
