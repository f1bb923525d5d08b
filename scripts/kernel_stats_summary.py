"""Prints the top kernels of a rocprofv3 *_kernel_stats.csv: calls, average / min / max us and
us per step (total / --steps).   python scripts/kernel_stats_summary.py FILE [--steps N] [--top K]"""
import argparse
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("file")
ap.add_argument("--steps", type=int, default=0)
ap.add_argument("--top", type=int, default=14)
a = ap.parse_args()
for r in list(csv.DictReader(open(a.file)))[:a.top]:
    name = re.sub(r"\(.*", "", r["Name"].replace("gf::(anonymous namespace)::", "").replace("void ", ""))[:44]
    per = "%9.1f" % (float(r["TotalDurationNs"]) / a.steps / 1e3) if a.steps else ""
    print("%-44s %7s avg %9.1f min %8.1f max %9.1f %s" % (
        name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, per))
