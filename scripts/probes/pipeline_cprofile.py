"""cProfile of the pipelined loop's Python side (ReplayPipeline.run), device or pinned tables:
where the host thread's time per step goes.  python scripts/probes/pipeline_cprofile.py [pinned]"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gnnflow_amd  # noqa: E402
from gnnflow_amd import synthetic  # noqa: E402
from gnnflow_amd.cache import LRUCache  # noqa: E402
from gnnflow_amd.pipeline import ReplayPipeline  # noqa: E402
from gnnflow_amd.utils import bind_to_device_cpus  # noqa: E402

placement = sys.argv[1] if len(sys.argv) > 1 else "pinned"
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
pipe.run(0, 2242)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
pipe.run(0, 2242)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
