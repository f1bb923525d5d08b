export GNNFLOW_PART_CHAIN_SMALL=0
python scripts/host_overhead_hash.py --lanes 2 --chain 4 --depth 12 2>&1 | grep "per step\|issuing" | tail -2
python scripts/host_overhead_hash.py --lanes 2 --chain 4 --depth 8 2>&1 | grep "per step\|issuing" | tail -2
python scripts/host_overhead_hash.py --lanes 1 --chain 4 --depth 8 2>&1 | grep "per step\|issuing" | tail -2
python scripts/host_overhead_hash.py --lanes 3 --chain 4 --depth 16 2>&1 | grep "per step\|issuing" | tail -2
