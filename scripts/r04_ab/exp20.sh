# main-thread time per step: the replica loop next to shared chains of 4 (one rank over RCCL)
python scripts/host_overhead_hash.py --replica --depth 2 2>&1 | grep "per step" 
python scripts/host_overhead_hash.py --replica --depth 2 --profile 2>&1 | grep -A26 "Ordered by"
python scripts/host_overhead_hash.py --lanes 2 --chain 4 --depth 12 2>&1 | grep "per step"
