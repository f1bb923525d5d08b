# BASELINE config 3 (10 M nodes / 200 M edges, uniform), ONE row per dispatch of the sampling
# kernels of ONE sample() per batch size, keyed (batch, layer, kernel): duration from a
# --kernel-trace pass, FETCH_SIZE and WRITE_SIZE from a --pmc pass each (one counter per pass,
# MI355X_MICROARCH.md HBM section), and the algorithmic bytes of that layer beside them, so that
# traffic / algorithmic is a column.  Every pass is a fresh process with the same seeds: the
# same roots, the same Philox draws, the same dispatches.
#   bash scripts/rocprof_config3_per_dispatch.sh <tag> [batches="60000 600000"]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-r05}
BATCHES=${2:-"60000 600000"}
mkdir -p gpurun_out/pmc
for B in $BATCHES; do
  A="--batches $B --policies uniform --reps 1"
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pmc -o ${TAG}_c3d_${B}_TIME -- python3 scripts/config3_bench.py $A > gpurun_out/pmc/${TAG}_c3d_${B}_TIME.log 2>&1
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc -o ${TAG}_c3d_${B}_$C -- python3 scripts/config3_bench.py $A > gpurun_out/pmc/${TAG}_c3d_${B}_$C.log 2>&1
  done
done
python3 - <<PY
import csv, glob, json, re
rows_out = []
for B in "$BATCHES".split():
    alg = None
    for line in open("gpurun_out/pmc/${TAG}_c3d_%s_TIME.log" % B):
        if line.startswith("{") and '"layers"' in line:
            alg = json.loads(line)
    def last_sample(path, value):
        f = glob.glob(path)
        if not f:
            return []
        seq = []
        rows = list(csv.DictReader(open(f[0])))
        key = "Dispatch_Id" if "Dispatch_Id" in rows[0] else None
        if key:
            rows.sort(key=lambda r: int(r[key]))
        for r in rows:
            m = re.search(r"(sample_\w+_kernel)", r["Kernel_Name"])
            if m:
                seq.append((m.group(1), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0), value(r)))
        n = len(seq) // 3                    # 2 warm-up samples + 1 timed: identical sequences
        return seq[2 * n:]
    t = last_sample("gpurun_out/pmc/${TAG}_c3d_%s_TIME_kernel_trace.csv" % B,
                    lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    fe = last_sample("gpurun_out/pmc/${TAG}_c3d_%s_FETCH_SIZE_counter_collection.csv" % B,
                     lambda r: float(r["Counter_Value"]))
    wr = last_sample("gpurun_out/pmc/${TAG}_c3d_%s_WRITE_SIZE_counter_collection.csv" % B,
                     lambda r: float(r["Counter_Value"]))
    layer, seen_emit = 0, False
    for i, (name, grid, us) in enumerate(t):
        if name.startswith("sample_search") and seen_emit:
            layer, seen_emit = layer + 1, False
        if "emit" in name:
            seen_emit = True
        row = {"batch": int(B), "layer": layer, "kernel": name, "grid": grid, "duration_us": us}
        if i < len(fe) and fe[i][0] == name:
            row["FETCH_SIZE_KiB_raw"] = fe[i][2]
        if i < len(wr) and wr[i][0] == name:
            row["WRITE_SIZE_KiB"] = wr[i][2]
        if alg and layer < len(alg["layers"]):
            L = alg["layers"][layer]
            kind = "search" if "search" in name else ("emit" if "emit" in name else None)
            row["layer_roots"], row["layer_edges"] = L["roots"], L["edges"]
            if kind:
                row["algorithmic_MB_of_the_layer_%s" % kind] = L[kind + "_alg_MB"]
                a = L[kind + "_alg_MB"] * 1e6
                if "FETCH_SIZE_KiB_raw" in row and "WRITE_SIZE_KiB" in row and a > 0:
                    f, w = row["FETCH_SIZE_KiB_raw"] * 1024, row["WRITE_SIZE_KiB"] * 1024
                    # gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads; scattered
                    # 4-32 B accesses are counted in full: raw = lower bound, x2 = upper bound
                    row["traffic_over_algorithmic_low"] = (f + w) / a
                    row["traffic_over_algorithmic_high"] = (2 * f + w) / a
                    row["traffic_TBps_low"] = (f + w) / (us * 1e-6) / 1e12
                    row["traffic_TBps_high"] = (2 * f + w) / (us * 1e-6) / 1e12
        rows_out.append(row)
json.dump({"command": "bash scripts/rocprof_config3_per_dispatch.sh ${TAG} '$BATCHES'",
           "what": "one row per dispatch of the last sample() of scripts/config3_bench.py --policies "
                   "uniform --reps 1 per batch size; kernels of one name may split a layer's bytes "
                   "among them (lane pass + group pass of the search): the layer's algorithmic "
                   "bytes are the same figure on each of their rows",
           "rows": rows_out}, open("gpurun_out/pmc/${TAG}_c3_per_dispatch.json", "w"), indent=1)
for r in rows_out:
    print({k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items()})
PY
rm -f gpurun_out/pmc/${TAG}_c3d_*_kernel_trace.csv gpurun_out/pmc/${TAG}_c3d_*_counter_collection.csv gpurun_out/pmc/*_agent_info.csv
