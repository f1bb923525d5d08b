python scripts/host_overhead_hash.py --lanes 2 --chain 4 --depth 12 2>&1 | grep "per step" | tail -2
python scripts/host_overhead_hash.py --replica --depth 2 2>&1 | grep "per step" | tail -1
