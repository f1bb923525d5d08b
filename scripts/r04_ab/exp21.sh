# asynchronous fetch submissions allowed with the enqueue thread (GNNFLOW_FETCH_QUEUED): 2 / 4 / 6
for Q in 2 4 6; do echo "queued $Q"
GNNFLOW_FETCH_QUEUED=$Q python scripts/host_overhead_hash.py --lanes 2 --chain 4 --depth 12 2>&1 | grep "per step" | tail -2
GNNFLOW_FETCH_QUEUED=$Q python scripts/host_overhead_hash.py --lanes 2 --chain 2 --depth 6 2>&1 | grep "per step" | tail -1
GNNFLOW_FETCH_QUEUED=$Q python scripts/host_overhead_hash.py --replica --depth 2 2>&1 | grep "per step" | tail -1
done
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],1), 'us/step depth', d['config']['pipeline_depth'])"; }
for Q in 2 4; do export GNNFLOW_FETCH_QUEUED=$Q; echo "bench, queued $Q: replica, hash chain 4, hash chain 2"; one; one --partition hash --always-exchange --part-chain 4;  one --partition hash --always-exchange --part-chain 2; done
