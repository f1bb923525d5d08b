"""Concurrency analysis of a rocprofv3 kernel trace: busy time per stream/queue, union
coverage, and per-kernel averages (reads gpurun_out/prof/<tag>_kernel_trace.csv)."""
import collections, csv, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
rows = list(csv.DictReader(open("gpurun_out/prof/%s_kernel_trace.csv" % tag)))
ev = []
for r in rows:
    m = re.search(r"(\w+_kernel|__amd_rocclr_\w+)", r["Kernel_Name"])
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Stream_Id"],
               m.group(1) if m else r["Kernel_Name"][:30]))
ev.sort()
# restrict to the steady-state region: after the last torch rand / fill kernels
t_first = next(s for s, e, q, st, n in ev if n == "gather_rows_kernel")
ev = [x for x in ev if x[0] >= t_first]
span = (max(e for s, e, *_ in ev) - ev[0][0]) / 1e3
busy_by_q = collections.defaultdict(float)
for s, e, q, st, n in ev:
    busy_by_q[(q, st)] += (e - s) / 1e3
union, cur_s, cur_e = 0.0, None, None
for s, e, *_ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += (cur_e - cur_s) / 1e3
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += (cur_e - cur_s) / 1e3
total = sum(busy_by_q.values())
print("span %.0f us, sum of kernel time %.0f us, union (any kernel running) %.0f us => GPU idle %.0f%%, overlap factor %.2f" % (
    span, total, union, 100 * (1 - union / span), total / union))
for k, v in sorted(busy_by_q.items(), key=lambda kv: -kv[1]):
    print("  queue %s stream %s busy %.0f us (%.0f%% of span)" % (k[0], k[1], v, 100 * v / span))
agg = collections.defaultdict(list)
for s, e, q, st, n in ev:
    agg[n].append((e - s) / 1e3)
for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print("  %-32s calls %5d avg %6.2f us total %8.0f us" % (n, len(v), sum(v) / len(v), sum(v)))
