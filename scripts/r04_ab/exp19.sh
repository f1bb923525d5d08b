# where does the step go with shared chains of 1 / 2 / 4 samples (one rank over RCCL)
for cfg in "2 1 6" "2 2 6" "2 4 12" "1 4 8" "2 4 8"; do set -- $cfg
python scripts/host_overhead_hash.py --lanes $1 --chain $2 --depth $3 2>&1 | grep -v amdgpu.ids | tail -4
done
