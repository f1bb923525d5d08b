run() { env $1 python bench.py --no-config3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$1: replica %.2f us  hash(no exchange) %.2f us  hash over rccl one rank %.2f us' % (d['ms_per_step']*1e3, d['hash_partition']['ms_per_step']*1e3, d['hash_partition_over_rccl_one_rank']['ms_per_step']*1e3))"; }
run A=1
run A=2
timeout -k 10 600 python -m pytest tests/test_gpu_partitioned.py tests/test_gpu_loopback_world8.py -x -q -m gpu 2>&1 | tail -2
