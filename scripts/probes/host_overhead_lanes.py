"""Where the main Python thread of the pipelined replay loop (ReplayPipeline.run, two sampling
lanes) spends a step: perf_counter_ns around each call of the loop body, device placement, with
and without the cache.   python scripts/probes/host_overhead_lanes.py"""
import os
import sys
import time
from collections import deque

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gnnflow_amd  # noqa: E402
from gnnflow_amd import synthetic  # noqa: E402
from gnnflow_amd.cache import LRUCache  # noqa: E402
from gnnflow_amd.pipeline import ReplayPipeline  # noqa: E402
from gnnflow_amd.utils import bind_to_device_cpus  # noqa: E402

bind_to_device_cpus(0)
dev = torch.device("cuda", 0)
g = synthetic.reddit_like(seed=42)
MiB = 1 << 20
graph = gnnflow_amd.DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert")
for lo in range(0, g["num_edges"], 100000):
    hi = lo + 100000
    graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
sampler = gnnflow_amd.TemporalSampler(graph, [10, 10], "recent", seed=1234)
gen = torch.Generator().manual_seed(1)
ef = torch.rand((g["num_edges"], 172), generator=gen)
nf = torch.rand((g["num_nodes"], 172), generator=gen)
batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(e).to(dev))
           for r, t, e in synthetic.replay_batches(g, 600, seed=42)]
nb = len(batches)
now = time.perf_counter_ns
# time spent inside the native calls of the loop (wrappers on the CDLL's cached functions)
from gnnflow_amd import _capi  # noqa: E402
lib = _capi.load()
NATIVE = {}


def _wrap(name):
    fn = getattr(lib, name)

    def timed(*a):
        t0 = now()
        r = fn(*a)
        NATIVE[name] = NATIVE.get(name, 0) + now() - t0
        return r
    setattr(lib, name, timed)


for _n in ("gf_sampler_sample_end", "gf_sampler_sample_begin_async", "gf_cache_fetch_blocks_async",
           "gf_cache_fetch_wait"):
    _wrap(_n)
for ratio in (0.2, 0.0):
    cache = LRUCache(ratio, ratio, g["num_nodes"], g["num_edges"], dev, nf, ef, 172, 172)
    cache.init_cache()
    pipe = ReplayPipeline(sampler, cache, batches, dev, pipelined=True)
    pipe.run(0, 2 * nb)
    torch.cuda.synchronize()
    lanes, nl = pipe.lanes, len(pipe.lanes)
    main = torch.cuda.current_stream(dev)

    def begin(j):
        r, t, _ = batches[j % nb]
        s, side = lanes[j % nl]
        return s.sample_async(r, t, stream=side, worker_enqueue=True)
    T = dict(wait=0, begin=0, rec=0, fetch=0)
    NATIVE.clear()
    import ctypes as C
    _b, _j = C.c_double(0), C.c_uint64(0)
    sys.stderr.write("[before] "); sys.stderr.flush()
    lib.gf_worker_stats(C.byref(_b), C.byref(_j))
    last = 4 * nb
    pending = deque()
    nxt = 0
    t_all = now()
    while nxt < last and len(pending) < pipe.depth:
        pending.append(begin(nxt))
        nxt += 1
    for i in range(last):
        t0 = now()
        mfgs = pending.popleft().wait()
        t1 = now()
        if nxt < last:
            pending.append(begin(nxt))
            nxt += 1
        t2 = now()
        for mfg in mfgs:
            for b in mfg:
                b.record_stream(main)
        t3 = now()
        cache.fetch_feature(mfgs, batches[i % nb][2], async_enqueue=True)
        t4 = now()
        T["wait"] += t1 - t0
        T["begin"] += t2 - t1
        T["rec"] += t3 - t2
        T["fetch"] += t4 - t3
    cache.wait_enqueued()
    torch.cuda.synchronize()
    total = (now() - t_all) / last / 1e3
    print("cache ratio %.1f: %.1f us per step | sample.wait %.1f | sample_async %.1f | record_stream %.1f | "
          "fetch_feature %.1f | rest (loop, timers) %.1f" % (
              ratio, total, T["wait"] / last / 1e3, T["begin"] / last / 1e3, T["rec"] / last / 1e3,
              T["fetch"] / last / 1e3, total - sum(T.values()) / last / 1e3))
    sys.stderr.write("[after %d steps] " % last); sys.stderr.flush()
    lib.gf_worker_stats(C.byref(_b), C.byref(_j))
    print("   inside native calls: " + ", ".join("%s %.1f" % (k, v / last / 1e3) for k, v in NATIVE.items()))
    del pipe, cache
