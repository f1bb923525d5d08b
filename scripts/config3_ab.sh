show() { python - "$1" <<PY
import json,sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d=json.loads(l)
        if d.get("sweep")=="config3": print(d["policy"], d["batch"], "search %.0f emit %.0f scan %.0f wall %.0f us" % (d["search_us"], d["emit_us"], d["scan_us"], d["wall_us"]))
PY
}
for cfg in "1 1048576" "0 1048576" "1 65536" "0 65536" "0 262144"; do set -- $cfg; echo "fences=$1 lane_min=$2"; GNNFLOW_SEARCH_FENCES=$1 GNNFLOW_LANE_SEARCH_MIN_ROOTS=$2 python scripts/config3_bench.py --batches 6000,60000,300000 --policies uniform,recent --reps 5 > gpurun_out/r03_c3_ab_$1_$2.jsonl 2>&1; show gpurun_out/r03_c3_ab_$1_$2.jsonl; done
