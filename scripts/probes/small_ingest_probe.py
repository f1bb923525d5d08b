import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import gnnflow_amd
from gnnflow_amd import synthetic
g = synthetic.reddit_like(seed=42)
for chunk in (600, 6000, 60000):
    graph = gnnflow_amd.DynamicGraph(20 << 20, 1000 << 20, "cuda", 62, 1024, "insert")
    E = g["num_edges"]
    # warm: first half in big chunks
    half = E // 2
    graph.add_edges(g["src"][:half], g["dst"][:half], g["ts"][:half], g["eid"][:half])
    t0 = time.perf_counter(); n = 0
    for lo in range(half, min(E, half + chunk * 200), chunk):
        hi = lo + chunk
        graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi]); n += 1
    dt = time.perf_counter() - t0
    print("chunk %6d: %.1f us per add_edges call, %.2f M edges/s" % (chunk, 1e6 * dt / n, chunk * n / dt / 1e6))
