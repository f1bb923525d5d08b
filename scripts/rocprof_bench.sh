# rocprofv3 kernel trace + stats of the bench workload (run on the GPU box via gpurun).
#   bash scripts/rocprof_bench.sh <tag> [bench args]     default args: --steps 300 --warmup 10
#   bash scripts/rocprof_bench.sh <tag> default          the exact default bench command
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r1}
shift
ARGS="$*"
if [ -z "$ARGS" ]; then ARGS="--steps 300 --warmup 10"; fi
if [ "$ARGS" = "default" ]; then ARGS=""; fi
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --no-cpu-baseline --no-placement-legs $ARGS ${BENCH_ARGS} > gpurun_out/prof/${TAG}_bench.log 2>&1
tail -1 gpurun_out/prof/${TAG}_bench.log | cut -c1-400
python3 scripts/analyze_trace.py $TAG
