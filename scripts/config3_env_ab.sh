#!/bin/bash
# Config 3's sampler sweep under two environments on ONE box:
#   bash scripts/config3_env_ab.sh "ENV_A" "ENV_B" [batches=6000,60000,300000] [policies=uniform,recent]
# e.g. "GNNFLOW_SEARCH_FENCES=1" "GNNFLOW_SEARCH_FENCES=0" (round 3: the timestamp fences),
#      "GNNFLOW_SEARCH_LAST_TS=1" "GNNFLOW_SEARCH_LAST_TS=0" (the newest-edge shortcut),
#      "GNNFLOW_LANE_SEARCH_MIN_ROOTS=65536" "GNNFLOW_LANE_SEARCH_MIN_ROOTS=1048576" (lane pass),
#      "GNNFLOW_EMIT_UNROLL=1" "" (round 6: emit slots per thread).
A=$1; B=$2; BATCHES=${3:-6000,60000,300000}; POL=${4:-uniform,recent}
mkdir -p gpurun_out/c3ab
for side in A B; do
  if [ $side = A ]; then E="$A"; else E="$B"; fi
  echo "== $side [$E]"
  env $E python scripts/config3_bench.py --batches $BATCHES --policies $POL --reps 5 > gpurun_out/c3ab/$side.jsonl 2> gpurun_out/c3ab/$side.err
  python - gpurun_out/c3ab/$side.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        if d.get("sweep") == "config3":
            print(d["policy"], d["batch"], "search %.0f emit %.0f scan %.0f wall %.0f us  search %.0f GB/s emit %.0f GB/s all %.0f" % (
                d["search_us"], d["emit_us"], d["scan_us"], d["wall_us"], d["search_GBps"], d["emit_GBps"], d["all_GBps"]))
PY
done
