timeout -k 10 600 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py tests/test_gpu_configs_4_5.py -x -q 2>&1 | tail -12
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],1), 'us/step', round(d['value']/1e6,1), 'M edges/s depth', d['config']['pipeline_depth'], d['config'].get('exchange','')[-90:])
"; }
for rep in 1 2; do
python bench.py $C 2>/dev/null | show "replica"
for L in 2 3 4; do
python bench.py $C --partition hash --always-exchange --part-lanes $L 2>/dev/null | show "hash pairs lanes$L"
done
GNNFLOW_PART_PAIR=0 python bench.py $C --partition hash --always-exchange --part-lanes 4 2>/dev/null | show "hash singles lanes4"
done
for a in "--lanes 2" "--lanes 4"; do timeout -k 10 120 python scripts/host_overhead_hash.py $a 2>/dev/null | tail -2; done
