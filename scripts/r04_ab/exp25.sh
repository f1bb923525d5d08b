# lanes per root of the shared chain's serve + own-share launch when m samples' layers together exceed 32 768 roots
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0 --steps 1121 --warmup 20"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2; do for W in 16 8 4 2; do echo "chain 4, width $W"; GNNFLOW_PART_CHAIN_WIDTH=$W one --partition hash --always-exchange; done
echo "chain 2, width 16 / 4";  GNNFLOW_PART_CHAIN_WIDTH=16 one --partition hash --always-exchange --part-chain 2; GNNFLOW_PART_CHAIN_WIDTH=4 one --partition hash --always-exchange --part-chain 2
done
