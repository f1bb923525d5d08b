# shared chains of up to 4 samples (gf_sampler_sample_partitioned_comm_group): parity, then one
# rank over RCCL at chain 1 / 2 / 3 / 4 (2 lanes), several depths, same box
timeout -k 10 900 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py tests/test_gpu_dist_features.py tests/test_gpu_configs_4_5.py -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 600 python scripts/fuzz_partitioned.py --seeds 150 2>&1 | tail -4 || exit 1
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],1), 'us/step depth', d['config']['pipeline_depth'])"; }
for rep in 1 2; do
echo replica; one
for K in 1 2 3 4; do echo "hash over RCCL (one rank), chain $K, 2 lanes"; one --partition hash --always-exchange --part-chain $K; done
echo "chain 4, 2 lanes, depth 8"; one --partition hash --always-exchange --part-chain 4 --pipeline-depth 8
echo "chain 4, 2 lanes, depth 16"; one --partition hash --always-exchange --part-chain 4 --pipeline-depth 16
echo "chain 4, 1 lane"; one --partition hash --always-exchange --part-chain 4 --part-lanes 1
echo "chain 4, 1 lane, depth 8"; one --partition hash --always-exchange --part-chain 4 --part-lanes 1 --pipeline-depth 8
echo "chain 4, 4 lanes"; one --partition hash --always-exchange --part-chain 4 --part-lanes 4
done
