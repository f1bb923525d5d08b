#!/bin/bash
# Round 4: the partitioned sample as a throughput pipeline.  One rank over RCCL (every message to
# the rank itself: --always-exchange), lanes 1 / 2 / 3 / 4, next to the replica loop and the
# exchange-free hash chain.  Writes gpurun_out/r04_part_bench.jsonl (one bench.py line each) and
# prints a summary with the enqueue threads' busy time per step (gf_worker_stats).
set -o pipefail
out=gpurun_out/r04_part_bench.jsonl
: > $out
run() { echo "# $*" >&2; "$@" 2>>gpurun_out/r04_part_bench.err | grep '^{' >> $out || echo '{"error": "'"$*"'"}' >> $out; }
C="--no-cpu-baseline --no-second-leg --no-config3 --steps 1121 --warmup 20 --min-seconds 1.0"
run python bench.py $C
run python bench.py $C --partition hash
for L in 1 2 3 4; do
  run python bench.py $C --partition hash --always-exchange --part-lanes $L
done
run python bench.py $C --partition hash --always-exchange --part-lanes 3 --pipeline-depth 6
python - <<'PY'
import json
for l in open("gpurun_out/r04_part_bench.jsonl"):
    d = json.loads(l)
    if "error" in d: print(d); continue
    c = d["config"]
    print("{:12s} n={} {:8.1f} M edges/s {:7.1f} us/step depth {} | {}".format(c["parallelism"], d["n_gpus"], d["value"]/1e6, 1e3*d["ms_per_step"], c["pipeline_depth"], c.get("exchange", "")[:150]))
PY
