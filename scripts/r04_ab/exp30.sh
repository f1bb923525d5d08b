C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0 --steps 1121 --warmup 20"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2 3; do
echo "chains on the fetch thread"; one --partition hash --always-exchange
echo "chains on their own thread"; GNNFLOW_PART_OWN_THREAD=1 one --partition hash --always-exchange
echo "chains on their own thread, 3 lanes depth 16"; GNNFLOW_PART_OWN_THREAD=1 one --partition hash --always-exchange --part-lanes 3 --pipeline-depth 16
echo "fetch thread, 3 lanes depth 16"; one --partition hash --always-exchange --part-lanes 3 --pipeline-depth 16
done
