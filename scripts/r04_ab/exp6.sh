C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],1), 'us/step', round(d['value']/1e6,1), 'M edges/s depth', d['config']['pipeline_depth'])
"; }
python bench.py $C 2>/dev/null | show "replica"
for FF in 0 1; do for OT in 0 1; do for L in 1 2 4; do
GNNFLOW_PIPELINE_FETCH_FIRST=$FF GNNFLOW_PART_OWN_THREAD=$OT python bench.py $C --partition hash --always-exchange --part-lanes $L 2>/dev/null | show "pairs lanes$L fetch_first=$FF own_thread=$OT"
done; done; done
for D in 2 3 6 8; do
python bench.py $C --partition hash --always-exchange --part-lanes 2 --pipeline-depth $D 2>/dev/null | show "pairs lanes2 depth$D"
done
python bench.py $C 2>/dev/null | show "replica"
