"""The 10 M-node / 200 M-edge graph of BASELINE config 3 built with add_edges in 10^7-edge
chunks: one JSON line per chunk (host time of the call, edges/s, the library's own per-phase
milliseconds from GNNFLOW_INGEST_PROFILE=1) and a summary line.  Run on the GPU box:
    python scripts/ingest_200M.py > gpurun_out/r03_ingest_200M.jsonl
(the per-phase figures are printed by the library on stderr; this script re-reads them)."""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if os.environ.get("GNNFLOW_INGEST_CHILD") != "1":
    # run the build in a child whose stderr (the phase timings) is captured
    env = dict(os.environ, GNNFLOW_INGEST_CHILD="1", GNNFLOW_INGEST_PROFILE="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                       capture_output=True, text=True)
    phases = []
    for line in p.stderr.splitlines():
        if line.startswith("add_edges") or "ms" in line and ":" in line and "n=" in line:
            phases.append(line.strip())
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    chunk_rows = [r for r in rows if r.get("record") == "chunk"]
    timing = [l for l in phases if l.startswith("add_edges")]
    for r, l in zip(chunk_rows, timing[-len(chunk_rows):] if timing else []):
        r["phases_ms"] = {k: float(v) for k, v in re.findall(r"(\w+) ([0-9.]+)", l.split(":", 1)[-1])}
    for r in rows:
        print(json.dumps(r))
    if p.returncode:
        sys.stderr.write(p.stderr[-3000:])
    sys.exit(p.returncode)

import numpy as np
import torch

import gnnflow_amd
from gnnflow_amd import synthetic
from gnnflow_amd.utils import bind_to_device_cpus

dev = torch.device("cuda", 0)
bind_to_device_cpus(0)   # host arrays and ingest threads on the GPU's NUMA node
N, E, CH = 10_000_000, 200_000_000, 10_000_000
g = synthetic.powerlaw_device(N, E, dev, seed=42)
graph = gnnflow_amd.DynamicGraph(1 << 30, 64 << 30, "cuda", 16, 1024, "insert")
t_all = time.perf_counter()
for lo in range(0, E, CH):
    hi = lo + CH
    t0 = time.perf_counter()
    graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
    dt = time.perf_counter() - t0
    print(json.dumps({"record": "chunk", "first_edge": lo, "edges": CH, "ms": round(1e3 * dt, 2),
                      "Medges_per_s": round(CH / dt / 1e6, 1)}), flush=True)
total = time.perf_counter() - t_all
print(json.dumps({"record": "build", "nodes": N, "edges": E, "chunk": CH, "seconds": round(total, 3),
                  "Medges_per_s": round(E / total / 1e6, 1), "num_edges": graph.num_edges(),
                  "num_vertices": graph.num_vertices(),
                  "graph_memory_MB": round(graph.get_graph_memory_usage() / 1e6, 1)}))
