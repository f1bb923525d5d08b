# emit kernel: four slots per thread and trip (memory-level parallelism) vs one (GNNFLOW_EMIT_UNROLL=1)
timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_sampler_parity.py -x -q 2>&1 | tail -2
for U in 1 0 1 0; do echo "GNNFLOW_EMIT_UNROLL=$U (1 = one slot per trip)"; GNNFLOW_EMIT_UNROLL=$U python scripts/config3_bench.py --batches 6000,60000,600000 --policies uniform,recent --reps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{') and '\"policy\"' in l:
        d=json.loads(l); print(' ', d['policy'], d['batch'], 'wall', round(d['wall_us']), 'search', round(d['search_us']), 'emit', round(d['emit_us']), 'emit frac', round(d['emit_frac'],3), 'all', round(d['all_frac'],3))"; done
