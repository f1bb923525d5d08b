# install kernel: block rows per row workgroup (GNNFLOW_INST_ROWS, temporary override)
timeout -k 10 600 python -m pytest tests/test_gpu_cache.py tests/test_gpu_pipeline_parity.py -x -q 2>&1 | tail -2
GNNFLOW_INST_ROWS=64 timeout -k 10 600 python -m pytest tests/test_gpu_cache.py -x -q 2>&1 | tail -2
for rep in 1 2; do for R in 256 128 64 512; do
GNNFLOW_INST_ROWS=$R python bench.py --no-config3 --min-seconds 1 --breakdown 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('rows', $R, 'us/step', round(1e3*d['ms_per_step'],2), {k:v for k,v in d.items() if 'breakdown' in k or 'kernel' in k})"
done; done
