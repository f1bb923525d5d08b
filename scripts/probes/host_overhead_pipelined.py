"""Host-time breakdown of the pipelined bench loop (diagnostic)."""
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
           for r, t, e in list(synthetic.replay_batches(g, 600))]
side = torch.cuda.Stream(device=dev); main = torch.cuda.current_stream(dev)
from collections import deque
T = dict(wait=0.0, begin=0.0, rec=0.0, fetch=0.0, wq=0.0)
orig_wait = cache.wait_enqueued
def timed_wait(upto=None):
    t0 = time.perf_counter(); orig_wait(upto); T["wq"] += time.perf_counter() - t0
cache.wait_enqueued = timed_wait
DEPTH = 2
for rep in range(3):
    for k in T: T[k] = 0.0
    cache.init_cache()
    torch.cuda.synchronize(); t00 = time.perf_counter()
    pending = deque(); nxt = 0
    while nxt < len(batches) and len(pending) < DEPTH:
        pending.append(sampler.sample_async(batches[nxt][0], batches[nxt][1], stream=side, worker_enqueue=True)); nxt += 1
    for i in range(len(batches)):
        t0 = time.perf_counter(); mfgs = pending.popleft().wait(); T["wait"] += time.perf_counter() - t0
        if nxt < len(batches):
            t0 = time.perf_counter(); pending.append(sampler.sample_async(batches[nxt][0], batches[nxt][1], stream=side, worker_enqueue=True)); nxt += 1; T["begin"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        for mfg in mfgs:
            for b in mfg: b.record_stream(main)
        T["rec"] += time.perf_counter() - t0
        t0 = time.perf_counter(); cache.fetch_feature(mfgs, batches[i][2], async_enqueue=True); T["fetch"] += time.perf_counter() - t0
    cache.wait_enqueued(); torch.cuda.synchronize(); tot = time.perf_counter() - t00
    n = len(batches)
    print("per step us: total %.1f | sample.wait %.1f | sample_async %.1f | record_stream %.1f | fetch_feature %.1f (of which wait_enqueued %.1f)" % (
        1e6*tot/n, 1e6*T["wait"]/n, 1e6*T["begin"]/n, 1e6*T["rec"]/n, 1e6*T["fetch"]/n, 1e6*T["wq"]/n))
import ctypes as C
lib = _capi.load(); bu, jb = C.c_double(0), C.c_uint64(0); lib.gf_worker_stats(C.byref(bu), C.byref(jb))
print("workers: %.1f us busy per job over %d jobs (all reps, both lanes)" % (bu.value / max(jb.value, 1), jb.value))
