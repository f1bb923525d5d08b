"""Host time per call of the pipelined loop over pinned tables (ReplayPipeline.run with a staging
cache): mean microseconds inside sample_async / PendingSample.wait / fetch_feature /
prefetch_feature, next to the step time.  python scripts/host_overhead_pinned.py [device|pinned]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import gnnflow_amd  # noqa: E402
from gnnflow_amd import synthetic, temporal_sampler  # noqa: E402
from gnnflow_amd.cache import LRUCache  # noqa: E402
from gnnflow_amd.pipeline import ReplayPipeline  # noqa: E402

placement = sys.argv[1] if len(sys.argv) > 1 else "pinned"
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
cache = LRUCache(0.2, 0.2, g["num_nodes"], g["num_edges"], dev, nf, ef, 172, 172,
                 feature_placement=placement)
cache.init_cache()
batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(e).to(dev))
           for r, t, e in synthetic.replay_batches(g, 600, seed=42)]
pipe = ReplayPipeline(sampler, cache, batches, dev, pipelined=True)
acc = {}


def timed(obj, name):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            rec = acc.setdefault(name, [0.0, 0])
            rec[0] += time.perf_counter() - t0
            rec[1] += 1
    setattr(obj, name, wrapper)


pipe.run(0, 1121)
pipe.run(0, 1121)
torch.cuda.synchronize()
t0 = time.perf_counter()
pipe.run(0, 1121)
torch.cuda.synchronize()
print("untimed loop: {:.1f} us per step".format(1e6 * (time.perf_counter() - t0) / 1121))
cache.init_cache()
torch.cuda.synchronize()
t0 = time.perf_counter()
pipe.run(0, 4 * 1121)
torch.cuda.synchronize()
print("untimed loop, 4 replays in one run: {:.1f} us per step".format(
    1e6 * (time.perf_counter() - t0) / (4 * 1121)))
timed(cache, "fetch_feature")
timed(cache, "prefetch_feature")
timed(temporal_sampler.PendingSample, "wait")
for s, _ in pipe.lanes:
    timed(s, "sample_async")
acc.clear()
t0 = time.perf_counter()
pipe.run(0, 1121)
torch.cuda.synchronize()
print("instrumented loop: {:.1f} us per step".format(1e6 * (time.perf_counter() - t0) / 1121))
for k, (t, n) in acc.items():
    print("  {:18s} {:7.2f} us per call, {} calls".format(k, 1e6 * t / max(n, 1), n))
if cache.staging:
    print(cache.staging_state())
