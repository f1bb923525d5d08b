C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0 --partition hash --always-exchange"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],1), 'us/step', round(d['value']/1e6,1), 'M edges/s depth', d['config']['pipeline_depth'])
"; }
for L in 2 3 4; do for D in 4 6 8; do
GNNFLOW_ENQUEUE_LANES=1 python bench.py $C --part-lanes $L --pipeline-depth $D 2>/dev/null | show "one-enqueue-thread lanes$L depth$D"
done; done
python bench.py $C --part-lanes 3 --pipeline-depth 6 2>/dev/null | show "two-threads lanes3 depth6"
python bench.py $C --part-lanes 4 --pipeline-depth 8 2>/dev/null | show "two-threads lanes4 depth8"
