C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],2), 'us/step', round(d['value']/1e6,1), 'M edges/s gather', round(d['roofline']['avg_launch_us'],2), 'frac', round(d['roofline']['frac'],3))
"; }
for rep in 1 2; do
python bench.py $C 2>/dev/null | show "default"
for I in 6 8 11 16; do GNNFLOW_GATHER_INFLIGHT=$I python bench.py $C 2>/dev/null | show "inflight=$I"; done
GNNFLOW_GATHER_TILE_ROWS=12 python bench.py $C 2>/dev/null | show "tile 12"
GNNFLOW_GATHER_TILE_ROWS=20 python bench.py $C 2>/dev/null | show "tile 20"
GNNFLOW_GATHER_TILE_ROWS=24 python bench.py $C 2>/dev/null | show "tile 24"
done
