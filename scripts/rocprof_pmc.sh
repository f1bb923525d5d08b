# HBM traffic of the gather kernel from PMC counters, one counter per pass (TCC has 4
# slots: FETCH_SIZE costs 3, WRITE_SIZE 2), per /opt/skills/guides/MI355X_MICROARCH.md.
#   bash scripts/rocprof_pmc.sh <tag> [bench args]     default: the driver's command line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r1}
shift
ARGS="${*:---steps 20 --warmup 5}"
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc -o ${TAG}_$C -- python3 bench.py --no-cpu-baseline --no-hash-leg --no-config3 --no-placement-legs --min-seconds 0.2 $ARGS > gpurun_out/pmc/${TAG}_$C.log 2>&1
done
ls gpurun_out/pmc | head
python3 - <<PY
import csv, collections, re, glob, json
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one counter per pass) -- python3 bench.py "
                  "--no-cpu-baseline --no-hash-leg --no-config3 --no-placement-legs --min-seconds 0.2 $ARGS",
       "kernel": "gather_rows_kernel", "all_kernels": {}}
for line in open("gpurun_out/pmc/${TAG}_FETCH_SIZE.log"):
    if line.startswith("{"):
        out["bench_args"] = json.loads(line)["config"]["workload_key"]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc/${TAG}_%s_counter_collection.csv" % c)
    if not f:
        print("missing", c); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in rows:
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"][:30]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    out["all_kernels"][c] = {k: {"dispatches": n, "sum_KB": v, "avg_KB_per_dispatch": v / n}
                             for k, (n, v) in agg.items() if k.endswith("_kernel") and not k.startswith("vectorized")
                             and "elementwise" not in k and "reduce" not in k}
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(c, "%-30s dispatches %5d sum %.1f avg/dispatch %.3f" % (k, n, v, v / n))
try:
    g_f = out["all_kernels"]["FETCH_SIZE"]["gather_rows_kernel"]
    g_w = out["all_kernels"]["WRITE_SIZE"]["gather_rows_kernel"]
    out["dispatches"] = g_f["dispatches"]
    out["FETCH_SIZE_KB_per_dispatch_raw"] = g_f["avg_KB_per_dispatch"]
    # gfx950: FETCH_SIZE counts half of the bytes of wide coalesced reads (MI355X_MICROARCH.md,
    # HBM section) -> doubled; both counters are in KiB
    out["FETCH_bytes_per_dispatch_x2_gfx950_correction"] = 2 * 1024 * g_f["avg_KB_per_dispatch"]
    out["WRITE_bytes_per_dispatch"] = 1024 * g_w["avg_KB_per_dispatch"]
    out["hbm_traffic_bytes_per_dispatch"] = (out["FETCH_bytes_per_dispatch_x2_gfx950_correction"]
                                             + out["WRITE_bytes_per_dispatch"])
    json.dump(out, open("gpurun_out/pmc/${TAG}_gather_traffic.json", "w"), indent=1)
    print("hbm traffic per gather dispatch: %.2f MB" % (out["hbm_traffic_bytes_per_dispatch"] / 1e6))
except KeyError as e:
    print("incomplete counters:", e)
PY
