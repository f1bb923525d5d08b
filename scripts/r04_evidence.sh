#!/bin/bash
# Round-4 evidence on one MI355X box (run through gpurun; everything lands in gpurun_out/):
#   1. the driver's command, plain                     -> r04_bench_<tag>.json
#   2. rocprofv3 kernel trace + stats of the same      -> prof/r04_<tag>_* , step timeline
#   3. PMC passes (FETCH_SIZE / WRITE_SIZE) of the same -> pmc/r04_<tag>_gather_traffic.json
#   4. config 3: PMC passes over scripts/config3_bench.py -> pmc/r04_<tag>_c3_traffic.json
#   5. the TGN-shaped epoch (config 5 stand-in)         -> r04_tgn_epoch_<tag>.json
TAG=${1:-a}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_bench_$TAG.json 2> gpurun_out/r04_bench_$TAG.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r04_bench_$TAG.json
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r04_$TAG -- python3 bench.py --no-cpu-baseline --no-config3 --steps 20 --warmup 5 --min-seconds 0.3 > gpurun_out/prof/r04_${TAG}_bench.log 2>&1; echo "trace rc=$?"
python3 scripts/analyze_trace.py r04_$TAG > gpurun_out/prof/r04_${TAG}_analysis.txt; head -24 gpurun_out/prof/r04_${TAG}_analysis.txt
python3 scripts/step_timeline.py r04_$TAG 20 4000 --summary gpurun_out/prof/r04_${TAG}_step_timeline.txt > gpurun_out/prof/r04_${TAG}_step_timeline.csv
rm -f gpurun_out/prof/r04_${TAG}_kernel_trace.csv     # tens of MB: only the summaries travel back
timeout -k 10 500 bash scripts/rocprof_pmc.sh r04_$TAG > gpurun_out/pmc/r04_${TAG}_pmc.log 2>&1; echo "pmc rc=$?"; tail -3 gpurun_out/pmc/r04_${TAG}_pmc.log
timeout -k 10 600 bash scripts/rocprof_config3.sh r04_$TAG > gpurun_out/pmc/r04_${TAG}_pmc_c3.log 2>&1; echo "pmc c3 rc=$?"
rm -f gpurun_out/pmc/*_counter_collection.csv gpurun_out/pmc/*_agent_info.csv
du -sh gpurun_out
timeout -k 10 400 python examples/tgn_epoch.py > gpurun_out/r04_tgn_epoch_$TAG.json 2> gpurun_out/r04_tgn_epoch_$TAG.err; echo "tgn rc=$?"; cut -c1-1500 gpurun_out/r04_tgn_epoch_$TAG.json; tail -3 gpurun_out/r04_tgn_epoch_$TAG.err
