R=$GRAFT_REPO_ROOT
for n in shipped st_early st_nb st_both shipped; do
  echo "== $n"
  if [ $n = shipped ]; then unset QMPC_LIB; else export QMPC_LIB=$R/variants/$n/libqmpc.so; fi
  timeout 300 python $R/tools/prio_proxy_ab.py 2>&1 >/dev/null | grep -v amdgpu
done
