timeout -k 10 600 python -m pytest tests/test_gpu_cache.py tests/test_gpu_pipeline_parity.py -x -q 2>&1 | tail -2
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],2), 'us/step', round(d['value']/1e6,1), 'M edges/s gather', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3))
"; }
for rep in 1 2 3; do
python bench.py $C 2>/dev/null | show "default (floor 16)"
GNNFLOW_GATHER_TILE_ROWS=8 python bench.py $C 2>/dev/null | show "forced 8"
done
python scripts/sweep.py gather 2>/dev/null | tail -12
