# kernel trace of the hash-partitioned loop, one rank over RCCL, N lanes: which streams / queues
# the chains run on and where the step's time goes
L=${1:-4}; D=${2:-8}
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r04_hash$L -- python3 bench.py --no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 0.3 --min-replays 1 --partition hash --always-exchange --part-lanes $L --pipeline-depth $D > gpurun_out/prof/r04_hash${L}_bench.log 2>&1
tail -1 gpurun_out/prof/r04_hash${L}_bench.log | cut -c1-300
python3 scripts/analyze_trace.py r04_hash$L | head -40
