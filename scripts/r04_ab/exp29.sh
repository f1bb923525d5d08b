timeout -k 10 900 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py tests/test_gpu_dist_features.py tests/test_gpu_configs_4_5.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -3 || exit 1
timeout -k 10 900 python scripts/fuzz_partitioned.py --seeds 600 2>&1 | tail -3 || exit 1
