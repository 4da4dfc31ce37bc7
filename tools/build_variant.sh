#!/bin/bash
# Build a variant of libqmpc.so with extra compile flags into variants/<name>/ (development: select it with QMPC_LIB=...).
# usage: tools/build_variant.sh <name> "<hipcc flags>"
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R && QMPC_EXTRA_HIPFLAGS="$2" python -c "import __graft_entry__ as g; print(g.build(out_dir='$R/variants/$1'))" && rm -rf $R/variants/$1/obj
