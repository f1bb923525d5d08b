C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],1), 'us/step', round(d['value']/1e6,1), 'M edges/s depth', d['config']['pipeline_depth'])
"; }
for rep in 1 2; do
python bench.py $C 2>/dev/null | show "replica (two issuing threads)"
GNNFLOW_ENQUEUE_LANES=1 python bench.py $C 2>/dev/null | show "replica (one issuing thread)"
python bench.py $C --partition hash 2>/dev/null | show "hash one rank (two threads)"
GNNFLOW_ENQUEUE_LANES=1 python bench.py $C --partition hash 2>/dev/null | show "hash one rank (one thread)"
GNNFLOW_ENQUEUE_LANES=1 GNNFLOW_PART_LANES=2 python bench.py $C --partition hash 2>/dev/null | show "hash one rank (one thread) lanes env 2 (ignored)"
done
