C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0 --steps 1121 --warmup 20"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2; do
for T in 0 4096 32768; do echo "chain 4, width 2 from $T roots on"; GNNFLOW_PART_CHAIN_SMALL=$T one --partition hash --always-exchange; done
echo "chain 4, width 4 from 0 roots on"; GNNFLOW_PART_CHAIN_WIDTH=4 GNNFLOW_PART_CHAIN_SMALL=0 one --partition hash --always-exchange
for G in 16 8 4; do echo "replica, search group $G"; GNNFLOW_SEARCH_GROUP=$G one; done
done
export GNNFLOW_PART_CHAIN_SMALL=4096
bash scripts/r04_trace_hash.sh 2 12 2>&1 | head -16
rm -f gpurun_out/prof/r04_hash2_kernel_trace.csv
