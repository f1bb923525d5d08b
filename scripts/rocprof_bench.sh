# rocprofv3 kernel trace + stats of the default bench workload (run on the GPU box via gpurun)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r1}
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o $TAG -- python3 bench.py --steps 300 --warmup 10 --no-cpu-baseline ${BENCH_ARGS} > gpurun_out/prof/${TAG}_bench.log 2>&1
tail -1 gpurun_out/prof/${TAG}_bench.log | cut -c1-400
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("gpurun_out/prof/${TAG}_kernel_trace.csv")))
by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"]
    short = name.split("(")[0].split("::")[-1][:40]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by[(short, r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(dur)
agg = collections.defaultdict(list)
for (short, grid), v in by.items():
    agg[short].append((len(v), sum(v) / len(v), grid))
for short, lst in sorted(agg.items(), key=lambda kv: -sum(n * a for n, a, _ in kv[1])):
    tot = sum(n * a for n, a, _ in lst)
    n = sum(n for n, _, _ in lst)
    print("%-42s calls %6d total %9.1f us avg %7.2f us" % (short, n, tot, tot / n))
PY
