import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import gnnflow_amd
from gnnflow_amd import synthetic
N, E = 2000000, 40000000
g = synthetic.powerlaw(N, E, seed=42)
graph = gnnflow_amd.DynamicGraph(1 << 30, 64 << 30, "cuda", 64, 1024, "insert")
t0 = time.time()
for lo in range(0, E, 10000000):
    hi = lo + 10000000
    graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
print("total %.2f s, %.1f M edges/s" % (time.time() - t0, E / (time.time() - t0) / 1e6))
