#!/bin/bash
# config 3, same box: the newest-edge shortcut of the node entries on / off
show() { python - "$1" <<PY
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l)
        if d.get("sweep")=="config3": print(d["policy"], d["batch"], "search %.0f emit %.0f scan %.0f wall %.0f us  search %.0f GB/s all %.0f" % (d["search_us"], d["emit_us"], d["scan_us"], d["wall_us"], d["search_GBps"], d["all_GBps"]))
PY
}
for f in 1 0; do echo "last_ts=$f"; GNNFLOW_SEARCH_LAST_TS=$f python scripts/config3_bench.py --batches 6000,60000,300000 --policies uniform,recent --reps 5 > gpurun_out/r03_c3_last$f.jsonl 2>&1; show gpurun_out/r03_c3_last$f.jsonl; done
