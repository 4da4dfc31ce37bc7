cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/queue
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
for h in 14 16; do python tools/class_stats.py --workload standing --horizon $h --batch 1024; done
for c in 1 2 3 4; do python bench.py --steps 200 --config $c --no-cpu-baseline --no-pipelined > gpurun_out/queue/cfg$c.json 2>/dev/null; done
for h in 10 14 16; do python bench.py --steps 100 --workload standing --horizon $h --no-cpu-baseline --no-pipelined > gpurun_out/queue/st$h.json 2>/dev/null; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/queue/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, '%.3e'%d['value'], '%.4f'%d['ms_per_step'], d['config'].get('failed'))
    except Exception as e: print(f, 'ERR', e)
PY
