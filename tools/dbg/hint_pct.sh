for pct in 0 25 50 0 50; do
  echo "pct $pct"; QMPC_SO_FIRST_PCT=$pct timeout 300 python tools/order_hint.py --quick --static-only 2>&1 | grep "^#"
done
