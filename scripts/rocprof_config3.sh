# PMC traffic of the sampler kernels on BASELINE config 3 (10 M nodes / 200 M edges): one
# counter per pass (FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2), kernel-trace only.
#   bash scripts/rocprof_config3.sh <tag> [config3_bench.py args]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r02}
shift
ARGS="${*:---batches 60000,600000 --policies uniform --reps 2}"
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc -o ${TAG}_c3_$C -- python3 scripts/config3_bench.py $ARGS > gpurun_out/pmc/${TAG}_c3_$C.log 2>&1
  tail -3 gpurun_out/pmc/${TAG}_c3_$C.log | cut -c1-300
done
python3 - <<PY
import csv, collections, re, glob, json
out = {"command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE (one per pass) -- python3 scripts/config3_bench.py $ARGS",
       "note": "counter values in KiB as rocprofv3 reports them; FETCH_SIZE = TCC_EA0_RDREQ x 64 B "
               "(MI355X_MICROARCH.md: exact x2 only for wide coalesced reads; these kernels issue 4 B probes "
               "and 16 B pair loads, so the raw figure is a LOWER bound of the sector traffic and 2x an upper bound)",
       "per_dispatch": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("gpurun_out/pmc/${TAG}_c3_%s_counter_collection.csv" % c)
    if not f:
        continue
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        m = re.search(r"(sample_\w+_kernel)", r["Kernel_Name"])
        if m:
            per[m.group(1)].append((int(r["Grid_Size"]) if "Grid_Size" in r else 0, float(r["Counter_Value"])))
    for k, rows in per.items():
        # group dispatches by grid size: each (batch, layer) has its own grid
        by = collections.defaultdict(list)
        for gsz, v in rows:
            by[gsz].append(v)
        out["per_dispatch"].setdefault(k, {})[c] = {
            str(gsz): {"dispatches": len(v), "avg_KiB": sum(v) / len(v)} for gsz, v in sorted(by.items())}
json.dump(out, open("gpurun_out/pmc/${TAG}_c3_traffic.json", "w"), indent=1)
print(json.dumps(out["per_dispatch"], indent=1)[:3000])
PY
