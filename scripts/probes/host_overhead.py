"""Where does the host time of one step go?  (diagnostic, not part of the bench)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import gnnflow_amd
from gnnflow_amd import _capi, synthetic
from gnnflow_amd.cache import LRUCache

dev = torch.device("cuda", 0)
g = synthetic.reddit_like(seed=42)
graph = gnnflow_amd.DynamicGraph(20 << 20, 1000 << 20, "cuda", 62, 1024, "insert")
for lo in range(0, g["num_edges"], 100000):
    graph.add_edges(g["src"][lo:lo+100000], g["dst"][lo:lo+100000], g["ts"][lo:lo+100000], g["eid"][lo:lo+100000])
sampler = gnnflow_amd.TemporalSampler(graph, [10, 10])
ef = torch.rand((g["num_edges"], 172), device=dev); nf = torch.rand((g["num_nodes"], 172), device=dev)
cache = LRUCache(0.2, 0.2, g["num_nodes"], g["num_edges"], dev, nf, ef, 172, 172); cache.init_cache()
batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(e).to(dev))
           for r, t, e in list(synthetic.replay_batches(g, 600))[500:900]]
lib = _capi.load()
orig_begin, orig_end, orig_fetch = lib.gf_sampler_sample_begin, lib.gf_sampler_sample_end, lib.gf_cache_fetch_blocks
T = {"begin": 0.0, "end": 0.0, "fetchC": 0.0, "sample": 0.0, "fetch": 0.0}
class Wrap:
    def __init__(self, fn, key): self.fn, self.key = fn, key
    def __call__(self, *a):
        t0 = time.perf_counter(); r = self.fn(*a); T[self.key] += time.perf_counter() - t0; return r
lib.gf_sampler_sample_begin = Wrap(orig_begin, "begin")
lib.gf_sampler_sample_end = Wrap(orig_end, "end")
lib.gf_cache_fetch_blocks = Wrap(orig_fetch, "fetchC")
for rep in range(2):
    for k in T: T[k] = 0.0
    torch.cuda.synchronize(); t00 = time.perf_counter()
    for r, t, e in batches:
        t0 = time.perf_counter(); m = sampler.sample(r, t); T["sample"] += time.perf_counter() - t0
        t0 = time.perf_counter(); cache.fetch_feature(m, e); T["fetch"] += time.perf_counter() - t0
    torch.cuda.synchronize(); tot = time.perf_counter() - t00
n = len(batches)
print("per step us: total %.1f | sample %.1f (C begin %.1f, C end(wait) %.1f, python %.1f) | fetch %.1f (C %.1f, python %.1f)" % (
    1e6*tot/n, 1e6*T["sample"]/n, 1e6*T["begin"]/n, 1e6*T["end"]/n, 1e6*(T["sample"]-T["begin"]-T["end"])/n,
    1e6*T["fetch"]/n, 1e6*T["fetchC"]/n, 1e6*(T["fetch"]-T["fetchC"])/n))
