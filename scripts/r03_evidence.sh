#!/bin/bash
# Round-3 evidence in one box visit (same box for everything): ingest build, gather sweep,
# config-3 search under rocprofv3 next to its own event timing, PMC traffic of the gather,
# kernel stats of the driver's command.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
set -o pipefail
timeout -k 10 300 python scripts/ingest_200M.py > gpurun_out/r03_ingest_200M.jsonl 2> gpurun_out/r03_ingest_200M.err && tail -1 gpurun_out/r03_ingest_200M.jsonl &&
SWEEP_ROWS=20000,198000,1000000,4000000 timeout -k 10 300 python scripts/sweep.py gather > gpurun_out/r03_gather_sweep.jsonl 2> gpurun_out/r03_gather_sweep.err && tail -2 gpurun_out/r03_gather_sweep.jsonl | cut -c1-300 &&
mkdir -p gpurun_out/prof &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r03_c3 -- python3 scripts/config3_bench.py --batches 300000 --policies uniform --reps 7 > gpurun_out/prof/r03_c3_bench.log 2>&1 && grep '"config3"' gpurun_out/prof/r03_c3_bench.log | cut -c1-400 && head -12 gpurun_out/prof/r03_c3_kernel_stats.csv | cut -c1-120 &&
timeout -k 10 400 bash scripts/rocprof_pmc.sh r03 2>&1 | tail -12 | cut -c1-200 &&
timeout -k 10 300 bash scripts/rocprof_bench.sh r03_f --steps 20 --warmup 5 --no-second-leg 2>&1 | grep -E "lru_|gather_rows|sample_|queue " | cut -c1-200
