C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 1.0"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ', round(1e3*d['ms_per_step'],2), 'us/step; gather', round(r['avg_launch_us'],2), 'us frac', round(r['frac'],3))"; }
F=ffffffff
for rep in 1 2; do
echo "no mask"; one
echo "first 64 CUs"; GNNFLOW_AB_SIDE_CU_MASK=$F,$F one
echo "first 128 CUs"; GNNFLOW_AB_SIDE_CU_MASK=$F,$F,$F,$F one
echo "every 4th CU (64)"; GNNFLOW_AB_SIDE_CU_MASK=11111111,11111111,11111111,11111111,11111111,11111111,11111111,11111111 one
echo "every 2nd CU (128)"; GNNFLOW_AB_SIDE_CU_MASK=55555555,55555555,55555555,55555555,55555555,55555555,55555555,55555555 one
echo "last 64 CUs"; GNNFLOW_AB_SIDE_CU_MASK=0,0,0,0,0,0,$F,$F one
done
