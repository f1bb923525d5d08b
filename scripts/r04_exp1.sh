C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0 --partition hash --always-exchange"
show() { python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1', round(1e3*d['ms_per_step'],1), 'us/step', round(d['value']/1e6,1), 'M edges/s depth', d['config']['pipeline_depth'])
"; }
python bench.py $C --part-lanes 4 2>/dev/null | show "lanes4"
GPU_MAX_HW_QUEUES=8 python bench.py $C --part-lanes 4 2>/dev/null | show "lanes4 hwq8"
GPU_MAX_HW_QUEUES=16 python bench.py $C --part-lanes 4 2>/dev/null | show "lanes4 hwq16"
GNNFLOW_PART_LANE_PRIORITY=-1 python bench.py $C --part-lanes 4 2>/dev/null | show "lanes4 prio-1"
python bench.py $C --part-lanes 4 --pipeline-depth 8 2>/dev/null | show "lanes4 depth8"
python bench.py $C --part-lanes 4 --pipeline-depth 4 2>/dev/null | show "lanes4 depth4"
GPU_MAX_HW_QUEUES=16 python bench.py $C --part-lanes 4 --pipeline-depth 8 2>/dev/null | show "lanes4 hwq16 depth8"
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r04_hash4 -- python3 bench.py --no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 0.3 --min-replays 1 --partition hash --always-exchange --part-lanes 4 > gpurun_out/prof/r04_hash4_bench.log 2>&1
python3 scripts/analyze_trace.py r04_hash4 | head -40
