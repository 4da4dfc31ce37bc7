for tr in 3 5 8; do
  QMPC_SO_TAIL_ROUNDS=$tr timeout 400 python tools/size_order_ab.py 50 > gpurun_out/size_order_ab_tr$tr.json 2>/dev/null
done
timeout 300 python tools/prio_proxy_ab.py > gpurun_out/prio_proxy_ab.json 2> gpurun_out/prio_proxy_ab.err
