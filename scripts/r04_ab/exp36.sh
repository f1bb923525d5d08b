# slow-host boxes: is the SAMPLING stream launch-bound?  (worker busy time, pipeline depth, one launch less per sample)
python scripts/host_overhead_hash.py --replica --depth 2 2>&1 | grep "per step" | tail -1 | cut -c1-250
python scripts/host_overhead_hash.py --replica --depth 3 2>&1 | grep "per step" | tail -1 | cut -c1-250
GNNFLOW_PUBLISH_EVENT=1 python scripts/host_overhead_hash.py --replica --depth 2 2>&1 | grep "per step" | tail -1 | cut -c1-250
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"; }
for rep in 1 2 3; do
echo "default"; one
echo "publish by event"; GNNFLOW_PUBLISH_EVENT=1 one
echo "depth 3"; one --pipeline-depth 3
done
