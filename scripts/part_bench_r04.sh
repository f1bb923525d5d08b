#!/bin/bash
# Round 4: the partitioned sample as a throughput pipeline.  One rank over RCCL (every message to
# the rank itself: --always-exchange) next to the replica loop and the exchange-free hash chain,
# on ONE box, every configuration twice (the run-to-run spread on a shared box is +-6 us):
# pairs (two samples per chain) at 1 / 2 / 4 lanes, single chains at 2 / 4 lanes, round 3's
# arrangement (1 lane, single chains, two-launch merge, own issuing thread).  Writes
# gpurun_out/r04_part_bench.jsonl (one bench.py line each, tagged) and the kernel stats of the
# default arrangement (pairs, 2 lanes).
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
out=gpurun_out/r04_part_bench.jsonl
: > $out
run() { tag="$1"; shift; echo "# $tag: $*" >&2
  "$@" 2>>gpurun_out/r04_part_bench.err | grep '^{' | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); d['tag']='$tag'; print(json.dumps(d))" >> $out || echo '{"error": "'"$tag"'"}' >> $out; }
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
H="$C --partition hash --always-exchange"
for rep in 1 2; do
run "replica" python bench.py $C
run "hash, no exchange (one rank)" python bench.py $C --partition hash
for L in 1 2 4; do run "pairs, $L lanes" python bench.py $H --part-lanes $L; done
for L in 2 4; do GNNFLOW_PART_PAIR=0 run "single chains, $L lanes" python bench.py $H --part-lanes $L; done
GNNFLOW_PART_PAIR=0 GNNFLOW_PART_FUSED_MERGE=0 GNNFLOW_PART_OWN_THREAD=1 run "round-3 arrangement (1 lane, single chains, 2-launch merge, own thread)" python bench.py $H --part-lanes 1
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r04_hashpairs -- python3 bench.py $H --min-seconds 0.3 --min-replays 1 > gpurun_out/prof/r04_hashpairs_bench.log 2>&1
rm -f gpurun_out/prof/r04_hashpairs_kernel_trace.csv
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_part_bench.jsonl"):
    d = json.loads(l)
    if "error" in d: print(d); continue
    c = d["config"]
    print("{:75s} {:7.1f} us/step {:7.1f} M edges/s depth {}".format(d["tag"], 1e3*d["ms_per_step"], d["value"]/1e6, c["pipeline_depth"]))
PY
head -14 gpurun_out/prof/r04_hashpairs_kernel_stats.csv | cut -c1-150
