"""A window of consecutive steps out of a rocprofv3 kernel trace of bench.py, both GPU streams:
one CSV row per dispatch (stream, kernel, start and duration in us relative to the window's
first dispatch, gap to the previous dispatch on the same stream) and a summary — per step and
stream: kernel time, gaps between dependent kernels, and the step's length — that says which
chain binds.
    python scripts/step_timeline.py <tag> [steps=20] [skip_steps=2000] > profiles/<..>.csv
reads gpurun_out/prof/<tag>_kernel_trace.csv; the summary goes to stderr and to <out>.txt when
--summary PATH is given."""
import collections
import csv
import re
import sys

tag = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
summary_path = None
if "--summary" in sys.argv:
    summary_path = sys.argv[sys.argv.index("--summary") + 1]

rows = list(csv.DictReader(open("gpurun_out/prof/%s_kernel_trace.csv" % tag)))
ev = []
for r in rows:
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    if not m or not m.group(1).startswith(("gather_rows", "lru_", "sample_")):
        continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Stream_Id"], m.group(1)))
ev.sort()
# a step = one gather launch on the main stream; the window starts at the skip-th gather
gathers = [i for i, e in enumerate(ev) if e[3] == "gather_rows_kernel"]
lo = ev[gathers[skip]][0]
hi = ev[gathers[skip + steps]][0]
win = [e for e in ev if lo <= e[0] < hi]
main_stream = ev[gathers[skip]][2]
t0 = win[0][0]
last_end = {}
out = csv.writer(sys.stdout)
out.writerow(["stream", "kernel", "start_us", "duration_us", "gap_after_previous_on_stream_us"])
per_stream = collections.defaultdict(lambda: {"busy": 0.0, "gaps": 0.0, "n": 0})
for s, e, st, name in win:
    gap = (s - last_end[st]) / 1e3 if st in last_end else 0.0
    last_end[st] = e
    role = "main (fetch)" if st == main_stream else "side (sampling)"
    out.writerow([role, name, "%.2f" % ((s - t0) / 1e3), "%.2f" % ((e - s) / 1e3), "%.2f" % gap])
    ps = per_stream[role]
    ps["busy"] += (e - s) / 1e3
    ps["gaps"] += gap
    ps["n"] += 1
span = (hi - lo) / 1e3
lines = ["window: %d consecutive steps, %.1f us per step (rocprofv3 kernel trace, tag %s)" % (
    steps, span / steps, tag)]
for role, ps in sorted(per_stream.items()):
    lines.append("%-16s %5.2f launches/step, kernels %6.2f us/step, gaps between its kernels %6.2f "
                 "us/step, idle %6.2f us/step" % (role, ps["n"] / steps, ps["busy"] / steps,
                                                  ps["gaps"] / steps,
                                                  (span - ps["busy"]) / steps))
by = collections.defaultdict(list)
for s, e, st, name in win:
    by[name].append((e - s) / 1e3)
for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    lines.append("  %-28s %5.2f per step, avg %6.2f us" % (name, len(v) / steps, sum(v) / len(v)))
text = "\n".join(lines)
sys.stderr.write(text + "\n")
if summary_path:
    open(summary_path, "w").write(text + "\n")
