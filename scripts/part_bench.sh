#!/bin/bash
# The partitioned path on a ONE-GPU box, as far as it can be measured there (no box with more
# than one GPU has been available): the native chain alone, the chain with the RCCL calls
# (one rank: every message empty) through the library's communicator and through
# torch.distributed, and 2 / 4 ranks sharing the card with the exchange staged through gloo +
# host memory.  Writes gpurun_out/r03_part_bench.jsonl (one bench.py line each).
set -o pipefail
out=gpurun_out/r03_part_bench.jsonl
: > $out
run() { echo "# $*" >&2; "$@" 2>>gpurun_out/r03_part_bench.err | grep '^{' >> $out || echo '{"error": "'"$*"'"}' >> $out; }
C="--no-cpu-baseline --no-second-leg --steps 1121 --warmup 20"
run python bench.py $C
run python bench.py $C --partition hash
run python bench.py $C --partition hash --always-exchange
GNNFLOW_PART_OVERLAP=0 run python bench.py $C --partition hash --always-exchange
GNNFLOW_PART_TRANSPORT=torch run python bench.py $C --partition hash --always-exchange
run python bench.py $C --partition hash --always-exchange --part-slack 0
run python bench.py $C --partition hash --shard-features
run python bench.py $C --partition hash --shard-features --always-exchange
if [ "$1" = "multi" ]; then
for n in 2 4; do
  GNNFLOW_BENCH_DEVICE=0 GNNFLOW_BENCH_BACKEND=gloo run python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2954$n bench.py --gpus $n $C --min-replays 1
  GNNFLOW_BENCH_DEVICE=0 GNNFLOW_BENCH_BACKEND=gloo run python -m torch.distributed.run --nnodes=1 \
    --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2955$n bench.py --gpus $n $C --min-replays 1 --partition replica
done
fi
python - <<'PY'
import json
for l in open("gpurun_out/r03_part_bench.jsonl"):
    d = json.loads(l)
    if "error" in d: print(d); continue
    c = d["config"]
    print("{:12s} n={} {:8.1f} M edges/s {:7.1f} us/step  {} | features: {}".format(c["parallelism"], d["n_gpus"], d["value"]/1e6, 1e3*d["ms_per_step"], c.get("exchange", ""), c.get("features", "replica")))
PY
