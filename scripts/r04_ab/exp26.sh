C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0 --steps 1121 --warmup 20"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2; do
for K in 2 3 4; do for W in 2 4; do echo "chain $K, width $W"; GNNFLOW_PART_CHAIN_WIDTH=$W one --partition hash --always-exchange --part-chain $K; done; done
for T in 4096 16384; do echo "chain 4, width 2 from $T roots on"; GNNFLOW_PART_CHAIN_SMALL=$T one --partition hash --always-exchange; done
echo "chain 4, width 2, 1 lane depth 8"; one --partition hash --always-exchange --part-lanes 1 --pipeline-depth 8
echo "chain 4, width 2, 2 lanes depth 8"; one --partition hash --always-exchange --pipeline-depth 8
echo "replica"; one
done
