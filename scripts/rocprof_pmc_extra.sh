# Extra PMC passes over the default bench (one counter group per pass, kernel-trace/stats only):
# L2 hit / miss per kernel, waves launched, LDS instruction mix.  Writes
# gpurun_out/pmc/<tag>_extra.json.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r1}
mkdir -p gpurun_out/pmc
for C in "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES_sum" "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_INSTS_LDS_ATOMIC"; do
  N=$(echo $C | tr ' ' '_')
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc -o ${TAG}_$N -- python3 bench.py --no-cpu-baseline --no-placement-legs --no-pipeline --steps 400 > gpurun_out/pmc/${TAG}_$N.log 2>&1
done
python3 - <<PY
import csv, collections, re, glob, json
out = {"command": "rocprofv3 --pmc <group> -- python3 bench.py --no-cpu-baseline --no-pipeline --steps 400 "
                  "(one group per pass)", "per_dispatch_average": {}}
for f in sorted(glob.glob("gpurun_out/pmc/${TAG}_*_counter_collection.csv")):
    if "FETCH_SIZE" in f or "WRITE_SIZE" in f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
        if not m or not m.group(1).startswith(("gather", "lru_", "sample_")):
            continue
        a = agg[m.group(1)][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
    for k, cs in agg.items():
        for c, (n, v) in cs.items():
            out["per_dispatch_average"].setdefault(k, {})[c] = v / n
for k, cs in out["per_dispatch_average"].items():
    if "TCC_HIT_sum" in cs and "TCC_MISS_sum" in cs:
        cs["L2_hit_rate"] = cs["TCC_HIT_sum"] / max(cs["TCC_HIT_sum"] + cs["TCC_MISS_sum"], 1)
json.dump(out, open("gpurun_out/pmc/${TAG}_extra.json", "w"), indent=1)
for k, cs in sorted(out["per_dispatch_average"].items()):
    print(k, {c: round(v, 3) for c, v in cs.items()})
PY
