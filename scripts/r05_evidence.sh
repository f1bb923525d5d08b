#!/bin/bash
# Round-5 evidence on one MI355X box (run through gpurun; everything lands in gpurun_out/):
#   1. the driver's command, plain                         -> r05_bench_<tag>.json
#   2. kernel stats of the REPLICA leg only (one leg per stats file) + step timeline
#   3. PMC passes (FETCH_SIZE / WRITE_SIZE) of the gather's launch
#   4. config 3, one row per dispatch keyed (batch, layer, kernel)
#   5. wire bytes per sample by world size; the TGN-shaped epoch
TAG=${1:-a}
export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05_bench_$TAG.json 2> gpurun_out/r05_bench_$TAG.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r05_bench_$TAG.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r05_$TAG -- python3 bench.py --no-hash-leg --no-config3 --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/prof/r05_${TAG}_bench.log 2>&1; echo "trace rc=$?"
python3 scripts/step_timeline.py r05_$TAG 20 4000 --summary gpurun_out/prof/r05_${TAG}_step_timeline.txt > gpurun_out/prof/r05_${TAG}_step_timeline.csv; cat gpurun_out/prof/r05_${TAG}_step_timeline.txt
rm -f gpurun_out/prof/r05_${TAG}_kernel_trace.csv
timeout -k 10 400 bash scripts/rocprof_pmc.sh r05_$TAG > gpurun_out/pmc/r05_${TAG}_pmc.log 2>&1; echo "pmc rc=$?"; tail -2 gpurun_out/pmc/r05_${TAG}_pmc.log
rm -f gpurun_out/pmc/*_counter_collection.csv gpurun_out/pmc/*_agent_info.csv
timeout -k 10 900 bash scripts/rocprof_config3_per_dispatch.sh r05_$TAG > gpurun_out/pmc/r05_${TAG}_c3d.log 2>&1; echo "c3 per dispatch rc=$?"; tail -14 gpurun_out/pmc/r05_${TAG}_c3d.log | cut -c1-400
timeout -k 10 120 python scripts/wire_bytes.py > gpurun_out/r05_wire_bytes_$TAG.json 2>/dev/null; echo "wire rc=$?"
timeout -k 10 400 python examples/tgn_epoch.py > gpurun_out/r05_tgn_epoch_$TAG.json 2> gpurun_out/r05_tgn_epoch_$TAG.err; echo "tgn rc=$?"
du -sh gpurun_out
