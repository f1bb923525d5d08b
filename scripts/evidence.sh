#!/bin/bash
# A round's evidence on ONE MI355X box (through gpurun; everything lands in gpurun_out/<round>/):
#   bash scripts/evidence.sh ROUND TAG [steps: bench trace pmc c3d wire tgn]      default: all
#   bench  the driver's command, plain                              -> bench_<tag>.json
#   trace  kernel stats of the REPLICA leg only + step timeline     -> prof_<tag>_*
#   pmc    FETCH_SIZE / WRITE_SIZE passes of the gather's launch    -> pmc_<tag>_*
#   c3d    config 3, one row per dispatch keyed (batch, layer, kernel)
#   wire   wire bytes per sample by world size
#   tgn    the TGN-shaped epoch + its kernel-family account (scripts/rocprof_tgn_epoch.sh)
# (rounds 3-5 had one such script each; this is the round-5 one with the round as an argument)
R=${1:-r06}; TAG=${2:-a}; shift 2 2>/dev/null
STEPS="${*:-bench trace pmc c3d wire tgn}"
export TMPDIR=/tmp
O=gpurun_out/$R; mkdir -p $O /tmp/prof gpurun_out/prof gpurun_out/pmc
for s in $STEPS; do case $s in
bench) timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_$TAG.json 2> $O/bench_$TAG.err; echo "bench rc=$?"; cut -c1-300 $O/bench_$TAG.json;;
trace) timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o ${R}_$TAG -- python3 bench.py --no-hash-leg --no-config3 --no-cpu-baseline --no-placement-legs --steps 20 --warmup 5 > gpurun_out/prof/${R}_${TAG}_bench.log 2>&1; echo "trace rc=$?"
   python3 scripts/step_timeline.py ${R}_$TAG 20 4000 --summary $O/prof_${TAG}_step_timeline.txt > $O/prof_${TAG}_step_timeline.csv; cat $O/prof_${TAG}_step_timeline.txt
   cp gpurun_out/prof/${R}_${TAG}_kernel_stats.csv $O/prof_${TAG}_kernel_stats_replica_only.csv; rm -f gpurun_out/prof/${R}_${TAG}_kernel_trace.csv;;
pmc) timeout -k 10 400 bash scripts/rocprof_pmc.sh ${R}_$TAG > $O/pmc_${TAG}.log 2>&1; echo "pmc rc=$?"; tail -2 $O/pmc_${TAG}.log; rm -f gpurun_out/pmc/*_counter_collection.csv gpurun_out/pmc/*_agent_info.csv;;
c3d) timeout -k 10 900 bash scripts/rocprof_config3_per_dispatch.sh ${R}_$TAG > $O/c3d_${TAG}.log 2>&1; echo "c3 per dispatch rc=$?"; tail -14 $O/c3d_${TAG}.log | cut -c1-400;;
wire) timeout -k 10 120 python scripts/wire_bytes.py > $O/wire_bytes_$TAG.json 2>/dev/null; echo "wire rc=$?";;
tgn) timeout -k 10 600 bash scripts/rocprof_tgn_epoch.sh $R > $O/tgn_$TAG.log 2>&1; echo "tgn rc=$?"; tail -5 $O/tgn_$TAG.log | cut -c1-300;;
esac; done
du -sh gpurun_out
