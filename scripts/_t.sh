for st in 1 5 17 64; do
python bench.py --no-hash-leg --no-config3 --no-cpu-baseline --event-stride $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('stride $st: %.2f us/step gather %.2f us (%d timed) lru %.2f us' % (d['ms_per_step']*1e3, r['avg_launch_us'], r['launches_timed'], d['roofline_lru']['avg_launch_us']))"
done
