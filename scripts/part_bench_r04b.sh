#!/bin/bash
# Round 4, final arrangement of the partitioned sample: ONE box, every configuration twice.
# One rank over RCCL (every message to the rank itself: --always-exchange) next to the replica
# loop and the exchange-free hash chain: shared chains of 1 / 2 / 4 samples (2 lanes), chains of 4
# with the 16-lane search the single chains use, 1 and 4 lanes.  Writes
# gpurun_out/r04_part_bench_final.jsonl (one bench.py line each, tagged) and the kernel stats of
# the default arrangement (chains of 4, 2 lanes).
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
out=gpurun_out/r04_part_bench_final.jsonl
: > $out
run() { tag="$1"; shift; echo "# $tag: $*" >&2
  "$@" 2>>gpurun_out/r04_part_bench_final.err | grep '^{' | python3 -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); d['tag']='$tag'; print(json.dumps(d))" >> $out || echo '{"error": "'"$tag"'"}' >> $out; }
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
H="$C --partition hash --always-exchange"
for rep in 1 2; do
run "replica" python bench.py $C
run "hash, no exchange (one rank)" python bench.py $C --partition hash
for K in 1 2 4; do run "chains of $K, 2 lanes" python bench.py $H --part-chain $K; done
GNNFLOW_PART_CHAIN_WIDTH=16 run "chains of 4, 2 lanes, 16 lanes per root in the serve launch" python bench.py $H
for L in 1 4; do run "chains of 4, $L lanes" python bench.py $H --part-lanes $L; done
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o r04_hashchain4 -- python3 bench.py $H --min-seconds 0.3 --min-replays 1 > gpurun_out/prof/r04_hashchain4_bench.log 2>&1
rm -f gpurun_out/prof/r04_hashchain4_kernel_trace.csv
python3 - <<'PY'
import json
for l in open("gpurun_out/r04_part_bench_final.jsonl"):
    d = json.loads(l)
    if "error" in d: print(d); continue
    c = d["config"]
    print("{:75s} {:7.1f} us/step {:7.1f} M edges/s depth {}".format(d["tag"], 1e3*d["ms_per_step"], d["value"]/1e6, c["pipeline_depth"]))
PY
head -12 gpurun_out/prof/r04_hashchain4_kernel_stats.csv | cut -c1-150
