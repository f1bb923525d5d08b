# one rank, nothing to exchange: shared chains of four without the all-to-alls
timeout -k 10 900 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py tests/test_gpu_configs_4_5.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -3
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0 --steps 1121 --warmup 20"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step depth', d['config']['pipeline_depth'])"; }
for rep in 1 2; do
echo replica; one
echo "hash, one rank, no exchange: chains of 4 / 2 / 1"; one --partition hash; one --partition hash --part-chain 2; one --partition hash --part-chain 1
echo "hash over RCCL chains of 4"; one --partition hash --always-exchange
done
