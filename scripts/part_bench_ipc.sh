#!/bin/bash
# 1 / 2 / 4 ranks SHARING the one GPU of the test box, the hash-partitioned replay with the
# native one-call chain over the hipIpc transport (GNNFLOW_PART_TRANSPORT=ipc; gloo carries the
# bootstrap and bench.py's bookkeeping collectives), replicated and sharded features.
# Writes gpurun_out/r03_part_bench_ipc.jsonl.
set -o pipefail
out=gpurun_out/r03_part_bench_ipc.jsonl
: > $out
run() { echo "# $*" >&2; "$@" 2>>gpurun_out/r03_part_bench_ipc.err | grep '^{' >> $out || echo '{"error": "'"$*"'"}' >> $out; }
C="--no-cpu-baseline --no-second-leg --steps 1121 --warmup 20 --min-replays 2"
export GNNFLOW_BENCH_DEVICE=0 GNNFLOW_BENCH_BACKEND=gloo GNNFLOW_PART_TRANSPORT=ipc
run python bench.py $C --partition hash
for n in 2 4; do
  run python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2956$n bench.py --gpus $n $C
  run python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2957$n bench.py --gpus $n $C --shard-features
  run python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2958$n bench.py --gpus $n $C --partition replica
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_part_bench_ipc.jsonl"):
    d = json.loads(l)
    if "error" in d: print(d); continue
    c = d["config"]
    print("{:12s} n={} {:8.1f} M edges/s {:7.1f} us/step | {} | features: {}".format(c["parallelism"], d["n_gpus"], d["value"]/1e6, 1e3*d["ms_per_step"], c.get("exchange", "")[:60], c.get("features", "replica")[:40]))
PY
