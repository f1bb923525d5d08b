# group width of the serve + own-share launch by the layer's real roots (16 lanes) instead of the slot rows
timeout -k 10 600 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py -x -q 2>&1 | tail -2
bash scripts/r04_trace_hash.sh 2 6 2>&1 | grep -v "^  queue" | head -16
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
for rep in 1 2 3; do python bench.py $C --partition hash --always-exchange 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('hash pairs', round(1e3*d['ms_per_step'],1), 'us/step')"; done
