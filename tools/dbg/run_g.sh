mkdir -p gpurun_out/$1
timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/$1/gpu_tests.txt 2>&1
tail -3 gpurun_out/$1/gpu_tests.txt
for a in "--steps 200" "--config 2 --steps 100" "--config 3 --steps 50" "--config 4 --steps 50" "--batch 16384 --steps 50"; do python bench.py $a --no-cpu-baseline --no-pipelined --no-closed-loop 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$a', '%.4e QP/s %.4f ms' % (d['value'], d['ms_per_step']), 'hinted', (d['order_hint'].get('hinted_same_inputs') or {}).get('value'), 'fail', d['config']['failed'], 'tail', (d.get('tail') or {}).get('longest_workgroup_cycles'), (d.get('tail') or {}).get('implied_floor_cycles'))
"; done > gpurun_out/$1/bench_quick.txt 2>&1
cat gpurun_out/$1/bench_quick.txt
