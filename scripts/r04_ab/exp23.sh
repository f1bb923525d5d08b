# torch's current stream by raw handle (torch._C._cuda_getCurrentRawStream) instead of torch.cuda.current_stream()
timeout -k 10 600 python -m pytest tests/test_gpu_cache.py tests/test_gpu_pipeline_parity.py tests/test_gpu_block_ops.py tests/test_memory.py -x -q 2>&1 | tail -2
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.5"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2 3; do
echo "raw handle: replica, hash chain 4"; one; one --partition hash --always-exchange
echo "Stream object: replica, hash chain 4"; GNNFLOW_AB_SLOW_STREAM=1 one; GNNFLOW_AB_SLOW_STREAM=1 one --partition hash --always-exchange
done
