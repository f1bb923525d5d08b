timeout -k 10 600 python -m pytest tests/test_gpu_cache.py tests/test_gpu_pipeline_parity.py -x -q 2>&1 | tail -3
export TMPDIR=/tmp; mkdir -p gpurun_out/prof
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-config3 --no-second-leg --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(1e3*d['ms_per_step'],2), 'us/step', round(d['value']/1e6,1), 'M edges/s gather', round(d['roofline']['avg_launch_us'],2))"; done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r04_g -- python3 bench.py --no-cpu-baseline --no-config3 --no-second-leg --steps 20 --warmup 5 --min-seconds 0.3 > gpurun_out/prof/r04_g_bench.log 2>&1
rm -f gpurun_out/prof/r04_g_kernel_trace.csv
head -8 gpurun_out/prof/r04_g_kernel_stats.csv | cut -d, -f1-4 | cut -c1-120
