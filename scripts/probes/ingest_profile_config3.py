import os, sys, time
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import gnnflow_amd
from gnnflow_amd import synthetic
dev = torch.device("cuda", 0)
from gnnflow_amd.utils import bind_to_device_cpus
bind_to_device_cpus(0)   # host arrays and ingest threads on the GPU's NUMA node
N, E = 10_000_000, 60_000_000
g = synthetic.powerlaw_device(N, E, dev, seed=42)
graph = gnnflow_amd.DynamicGraph(1 << 30, 64 << 30, "cuda", 16, 1024, "insert")
t0 = time.time()
for lo in range(0, E, 10_000_000):
    hi = lo + 10_000_000
    graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
print("ingest %.2f s  %.1f M edges/s" % (time.time() - t0, E / (time.time() - t0) / 1e6))
