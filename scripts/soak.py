"""Soak run of the pipelined step loop: throughput and memory (HBM via torch, host RSS) every
10 000 steps — leaks in the event pool, the slab allocator or the lazy views would show here."""
import os, sys, time, resource
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnnflow_amd
from gnnflow_amd import synthetic
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
           for r, t, e in synthetic.replay_batches(g, 600)]
side = torch.cuda.Stream(device=dev); main = torch.cuda.current_stream(dev)
total = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
nb = len(batches)
pending = sampler.sample_async(batches[0][0], batches[0][1], stream=side, worker_enqueue=True)
t0 = time.time(); edges = 0
for i in range(total):
    mfgs = pending.wait()
    r, t, _ = batches[(i + 1) % nb]
    pending = sampler.sample_async(r, t, stream=side, worker_enqueue=True)
    for mfg in mfgs:
        for b in mfg:
            b.record_stream(main)
    cache.fetch_feature(mfgs, batches[i % nb][2], async_enqueue=True)
    edges += mfgs[0][0].num_edges() + mfgs[1][0].num_edges()
    if i % 997 == 0:                      # touch the lazy views now and then
        _ = mfgs[0][0].srcdata["h"].shape, mfgs[0][0].edata["f"].shape, cache.target_edge_features.shape
    if (i + 1) % 10000 == 0:
        torch.cuda.synchronize()
        dt = time.time() - t0
        print("step %7d  %.1f M edges/s  hbm alloc %.0f MB reserved %.0f MB  rss %.0f MB  hit %.3f" % (
            i + 1, edges / dt / 1e6, torch.cuda.memory_allocated() / 1e6,
            torch.cuda.memory_reserved() / 1e6, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e3,
            float(cache.cache_edge_ratio)), flush=True)
        t0 = time.time(); edges = 0
pending.wait(); cache.wait_enqueued(); torch.cuda.synchronize()
print("done")
