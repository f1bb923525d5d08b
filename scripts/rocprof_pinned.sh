# Kernel stats + a window of the kernel timeline of the pinned-placement loop (tables in pinned host
# memory, LRU 0.2, staging ring), through gpurun:   bash scripts/rocprof_pinned.sh [tag]
export TMPDIR=/tmp
TAG=${1:-r06}
mkdir -p /tmp/prof gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o pin -- python3 bench.py --no-cpu-baseline --no-config3 --no-hash-leg --no-placement-legs --feature-placement pinned --steps 20 --warmup 5 > /tmp/prof/pin.log 2>&1
grep "^{" /tmp/prof/pin.log | tail -1 > gpurun_out/$TAG/pinned_bench_profiled.json
cp /tmp/prof/pin_kernel_stats.csv gpurun_out/$TAG/pinned_kernel_stats.csv
python3 scripts/kernel_stats_summary.py gpurun_out/$TAG/pinned_kernel_stats.csv --top 8
python3 - /tmp/prof/pin_kernel_trace.csv <<'PY' > gpurun_out/$TAG/pinned_step_timeline.txt
import csv, re, sys
ev = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    if m and m.group(1).startswith(("gather_rows", "lru_", "sample_", "stage_")):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "q" + r.get("Queue_Id", "?"), m.group(1)))
ev.sort()
g = [i for i, e in enumerate(ev) if e[3].startswith("gather_rows")]
lo, hi = ev[g[20000]][0], ev[g[20008]][0]
print("eight steps of the pinned-placement loop under rocprofv3 (us from the first gather; the tracer")
print("slows the host, the kernels' durations are what to read):")
for s, e, q, n in ev:
    if lo <= s < hi:
        print("%8.1f %6.1f %-4s %s" % ((s - lo) / 1e3, (e - s) / 1e3, q, n))
print("period %.1f us" % ((hi - lo) / 8e3))
PY
tail -3 gpurun_out/$TAG/pinned_step_timeline.txt
python3 bench.py --no-cpu-baseline --no-config3 --no-hash-leg --no-placement-legs --feature-placement pinned --steps 20 --warmup 5 2>/dev/null > gpurun_out/$TAG/pinned_bench.json
cut -c1-200 gpurun_out/$TAG/pinned_bench.json
