timeout -k 10 600 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py -x -q 2>&1 | tail -3
for a in "--lanes 1" "--lanes 2" "--lanes 3 --depth 6" "--lanes 4 --depth 8"; do timeout -k 10 120 python scripts/host_overhead_hash.py $a 2>/dev/null | tail -2; done
GNNFLOW_PART_FUSED_MERGE=0 timeout -k 10 120 python scripts/host_overhead_hash.py --lanes 3 --depth 6 2>/dev/null | tail -2
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0 --partition hash --always-exchange"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],1), 'us/step', round(d['value']/1e6,1), 'M edges/s depth', d['config']['pipeline_depth'])
"; }
for L in 2 3 4; do for D in 4 6 8; do
python bench.py $C --part-lanes $L --pipeline-depth $D 2>/dev/null | show "lanes$L depth$D"
done; done
