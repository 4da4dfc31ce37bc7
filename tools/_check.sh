cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/queue
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python tools/gpu_phases.py s14 256 | sed -n 7,10p
python tools/gpu_phases.py s10 256 | sed -n 7,10p
for c in 1 3 4; do timeout 300 python bench.py --steps 200 --config $c --no-cpu-baseline --no-pipelined > gpurun_out/queue/cfg$c.json 2>/dev/null; done
for h in 10 14 16; do timeout 300 python bench.py --steps 100 --workload standing --horizon $h --no-cpu-baseline --no-pipelined > gpurun_out/queue/st$h.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/queue/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%d['value'], '%.4f'%d['ms_per_step'], d['config'].get('failed'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 300 python tools/shim_latency.py 2>/dev/null | python -c "import sys,json; d=json.load(sys.stdin); print([(x['family'], x['median_us']) for x in d['shim_latency']])"
