# publish by event (no publish kernel) again: three more pairs on another box + the sampler / pipeline tests under it
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2 3 4; do
echo "default"; one
echo "publish by event"; GNNFLOW_PUBLISH_EVENT=1 one
done
GNNFLOW_PUBLISH_EVENT=1 timeout -k 10 600 python -m pytest tests/test_gpu_sampler_parity.py tests/test_gpu_pipeline_parity.py tests/test_gpu_golden.py tests/test_gpu_fuzz.py tests/test_gpu_config3.py -x -q 2>&1 | tail -2
