# TEMPORARY experiment hooks: the fused copy || scan launch with one role switched off (timing only, results invalid), and copy workgroups with fewer active waves
C="--no-cpu-baseline --no-second-leg --no-config3 --min-seconds 0.6"
one() { python bench.py $C "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('  ', round(1e3*d['ms_per_step'],2), 'us/step; gather', round(r['avg_launch_us'],2), 'us')"; }
echo both; one
echo "copy only"; GNNFLOW_AB_ROLE=1 one
echo "scan only"; GNNFLOW_AB_ROLE=2 one
for W in 8 4 2; do echo "copy waves per WG $W"; GNNFLOW_AB_COPY_WAVES=$W one; done
echo both; one
