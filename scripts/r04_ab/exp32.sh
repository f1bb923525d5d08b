# 12-byte reply slots in the shared chains (ids that fit 32 bits) against the 24-byte ones
timeout -k 10 900 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py tests/test_gpu_configs_4_5.py tests/test_gpu_dist_features.py -x -q 2>&1 | tail -3
timeout -k 10 300 python scripts/fuzz_partitioned.py --seeds 800 --first 5000 2>&1 | tail -2
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0 --steps 1121 --warmup 20"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2 3; do
echo "hash over RCCL, chains of 4: 12-byte / 24-byte reply slots"; one --partition hash --always-exchange; GNNFLOW_PART_NARROW=0 one --partition hash --always-exchange
echo "hash, no exchange: 12 / 24"; one --partition hash; GNNFLOW_PART_NARROW=0 one --partition hash
done
echo replica; one
