for V in "GNNFLOW_PART_CHAIN=1" "GNNFLOW_PART_CHAIN=2" "GNNFLOW_PART_CHAIN=3 GNNFLOW_PART_NARROW=0" "GNNFLOW_PART_LANES=1" "GNNFLOW_PART_OWN_THREAD=1" "GNNFLOW_ENQUEUE_LANES=1" "GNNFLOW_PART_CHAIN_WIDTH=16 GNNFLOW_PART_FUSED_MERGE=0"; do
echo "== $V"
env $V timeout -k 10 600 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py tests/test_gpu_configs_4_5.py tests/test_gpu_dist_features.py tests/test_gpu_bench_contract.py -x -q -k "not hanging and not single_gpu_line and not dying" 2>&1 | tail -2
done
