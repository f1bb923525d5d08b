C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],1), 'us/step', round(d['value']/1e6,1), 'M edges/s depth', d['config']['pipeline_depth'], 'gather us', round(d.get('roofline',{}).get('avg_launch_us',0),2))
"; }
for rep in 1 2; do
GNNFLOW_SAMPLER_LANES=1 python bench.py $C 2>/dev/null | show "replica 1 lane"
python bench.py $C 2>/dev/null | show "replica 2 lanes"
python bench.py $C --pipeline-depth 3 2>/dev/null | show "replica 2 lanes depth3"
GNNFLOW_SAMPLER_LANES=3 python bench.py $C --pipeline-depth 3 2>/dev/null | show "replica 3 lanes depth3"
GNNFLOW_PIPELINE_FETCH_FIRST=1 python bench.py $C 2>/dev/null | show "replica 2 lanes fetch-first"
for L in 3 4; do for D in 4 8; do
GNNFLOW_PIPELINE_FETCH_FIRST=0 python bench.py $C --partition hash --always-exchange --part-lanes $L --pipeline-depth $D 2>/dev/null | show "hash lanes$L depth$D sample-first"
python bench.py $C --partition hash --always-exchange --part-lanes $L --pipeline-depth $D 2>/dev/null | show "hash lanes$L depth$D fetch-first"
done; done
done
