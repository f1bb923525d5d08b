for rep in 1 2 3; do
echo "old"; (cd _ab_old && python scripts/host_overhead_hash.py --replica --depth 2 2>&1 | grep "per step" | tail -1 | cut -c1-260)
echo "new"; python scripts/host_overhead_hash.py --replica --depth 2 2>&1 | grep "per step" | tail -1 | cut -c1-260
done
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0"
for rep in 1 2 3; do
echo "old bench"; (cd _ab_old && python bench.py $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')")
echo "new bench"; python bench.py $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('  ', round(1e3*d['ms_per_step'],2), 'us/step')"
done
