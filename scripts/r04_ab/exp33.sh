# gather in two launches (probe, then copy || list scan) against the fused gather + scan
timeout -k 10 900 python -m pytest tests/test_gpu_cache.py tests/test_gpu_pipeline_parity.py tests/test_gpu_harness.py tests/test_gpu_dist_features.py tests/test_gpu_golden.py -x -q 2>&1 | tail -3
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ', round(1e3*d['ms_per_step'],2), 'us/step; gather', round(r['avg_launch_us'],2), 'us frac', round(r['frac'],3))"; }
for rep in 1 2 3; do
echo "split"; one
echo "fused"; GNNFLOW_GATHER_SPLIT=0 one
done
echo "split, breakdown"; python bench.py $C --breakdown 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:(round(1e3*v['total_ms']/max(v['intervals'],1),2), v['intervals']) for k,v in d['kernel_breakdown_200_steps'].items()})"
