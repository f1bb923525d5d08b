# lane-per-root search pass: four chunks' table entries requested per trip
timeout -k 10 600 python -m pytest tests/test_gpu_config3.py tests/test_gpu_sampler_parity.py -x -q 2>&1 | tail -2
for i in 1 2; do python scripts/config3_bench.py --batches 60000,600000 --policies uniform,recent --reps 5 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{') and '\"policy\"' in l:
        d=json.loads(l); print(' ', d['policy'], d['batch'], 'wall', round(d['wall_us']), 'search', round(d['search_us']), 'emit', round(d['emit_us']), 'search frac', round(d['search_frac'],3), 'all', round(d['all_frac'],3))"; done
