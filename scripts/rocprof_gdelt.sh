# rocprofv3 kernel stats of the full-size GDELT-shaped step (config 4 on one GPU), run via gpurun:
#   bash scripts/rocprof_gdelt.sh <tag> [gdelt_scale_bench.py args]
# Keeps only the per-kernel stats (the trace of 2000 steps is large) under gpurun_out/gdelt/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-gdelt}
shift
ARGS="$*"
if [ -z "$ARGS" ]; then ARGS="--batches 2000 --warmup 800"; fi
mkdir -p gpurun_out/gdelt /tmp/gdelt_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/gdelt_prof -o $TAG -- python3 scripts/gdelt_scale_bench.py $ARGS > gpurun_out/gdelt/${TAG}.log 2>&1
tail -1 gpurun_out/gdelt/${TAG}.log | cut -c1-600
find /tmp/gdelt_prof -name "${TAG}_kernel_stats.csv" -exec cp {} gpurun_out/gdelt/ \;
head -25 gpurun_out/gdelt/${TAG}_kernel_stats.csv | cut -c1-200
