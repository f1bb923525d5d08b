# HBM traffic of the gather kernel from PMC counters, one counter per pass (TCC has 4
# slots: FETCH_SIZE costs 3, WRITE_SIZE 2), per /opt/skills/guides/MI355X_MICROARCH.md.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r1}
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc -o ${TAG}_$C -- python3 bench.py --no-cpu-baseline --no-pipeline > gpurun_out/pmc/${TAG}_$C.log 2>&1
done
ls gpurun_out/pmc | head
python3 - <<PY
import csv, collections, re, glob
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
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(c, "%-30s dispatches %5d sum %.1f avg/dispatch %.3f" % (k, n, v, v / n))
PY
