"""Where the one-launch LRU list update (lru_list_fused_kernel) spends its time: every workgroup
stamps the 100 MHz wall clock at its role's stages (gf_debug_lru_trace); this replays the
REDDIT-shaped stream (LRU 0.2, batch 600, the headline's fetch) through the plain loop, traces the
edge cache's update of N late steps and prints, per role and stage, the median / 10th / 90th
percentile over workgroups and steps of the time since the launch's FIRST stamp.
    python scripts/lru_hop_trace.py [steps=40] > profiles/r06_lru_hop_trace.txt
Reference: gnnflow/cache/lru_cache.py:121-201 (the update this launch replaces)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gnnflow_amd  # noqa: E402
from gnnflow_amd import _capi, synthetic  # noqa: E402
from gnnflow_amd.cache import LRUCache  # noqa: E402
from gnnflow_amd.utils import bind_to_device_cpus  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
bind_to_device_cpus(0)
lib = _capi.load()
dev = torch.device("cuda", 0)
g = synthetic.reddit_like(seed=42)
MiB = 1 << 20
graph = gnnflow_amd.DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert")
for lo in range(0, g["num_edges"], 100000):
    hi = lo + 100000
    graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
sampler = gnnflow_amd.TemporalSampler(graph, [10, 10], "recent", seed=1234)
gen = torch.Generator(device=dev).manual_seed(1)
ef = torch.rand((g["num_edges"], 172), generator=gen, device=dev)
nf = torch.rand((g["num_nodes"], 172), generator=gen, device=dev)
cache = LRUCache(0.2, 0.2, g["num_nodes"], g["num_edges"], dev, nf, ef, 172, 172)
cache.init_cache()
batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(e).to(dev))
           for r, t, e in synthetic.replay_batches(g, 600, seed=42)]
first = len(batches) - steps - 1
for i in range(first):                       # bring the cache to the late part of the replay
    r, t, e = batches[i]
    cache.fetch_feature(sampler.sample(r, t), e)
torch.cuda.synchronize()
_capi.check(lib.gf_debug_lru_trace_enable(cache._edge.h, 1))
STAGES = {"count": ["entry", "marks / list / parity / counters in", "entries staged + drained, count published"],
          "row": ["entry", "rows read, representatives ranked, count published",
                  "look-back complete (row + count granules), prefixes in LDS",
                  "victims read, map / slot_id written", "installed rows copied",
                  "(thread 0's own granules in)", "(all threads' granules in: first barrier behind the polls)"],
          "write": ["entry", "row + count granules in", "list tile(s) rewritten"]}
acc = {role: [[] for _ in names] for role, names in STAGES.items()}
ends = []
buf = (C.c_uint64 * (4 + 8 * 6000))()
n = C.c_size_t(0)
for i in range(first, first + steps):
    r, t, e = batches[i]
    cache.fetch_feature(sampler.sample(r, t), e)
    torch.cuda.synchronize()
    _capi.check(lib.gf_debug_lru_trace(cache._edge.h, buf, len(buf), C.byref(n)))
    w = np.frombuffer(buf, dtype=np.uint64, count=n.value).astype(np.int64)
    cb, rb, wb = int(w[0]), int(w[1]), int(w[2])
    st = w[4:4 + 8 * (cb + rb + wb)].reshape(-1, 8)
    used = st[:, 0] > 0
    t0 = st[used, 0].min()
    last = 0
    for role, lo, hi in (("count", 0, cb), ("row", cb, cb + rb), ("write", cb + rb, cb + rb + wb)):
        for wg in range(lo, hi):
            if st[wg, 0] == 0:
                continue
            for k in range(len(STAGES[role])):
                if st[wg, k] >= st[wg, 0] and st[wg, k] > 0:
                    acc[role][k].append((st[wg, k] - t0) * 0.01)     # 100 MHz -> us
                    last = max(last, st[wg, k] - t0)
    ends.append(last * 0.01)
    lib.gf_debug_lru_trace_enable(cache._edge.h, 1)      # zero the stamps for the next update
print("lru_list_fused_kernel, edge cache of the headline replay (134 489 slots, ~9.5 k-row blocks),")
print("{} consecutive late updates; microseconds since the launch's first stamp:".format(steps))
print("{:6s} {:60s} {:>7s} {:>7s} {:>7s} {:>6s}".format("role", "stage", "p10", "median", "p90", "n"))
for role, names in STAGES.items():
    for k, name in enumerate(names):
        v = np.array(acc[role][k])
        if len(v):
            print("{:6s} {:60s} {:7.2f} {:7.2f} {:7.2f} {:6d}".format(
                role, name, np.percentile(v, 10), np.median(v), np.percentile(v, 90), len(v)))
print("last stamp of the launch: median {:.2f} us (p10 {:.2f}, p90 {:.2f})".format(
    np.median(ends), np.percentile(ends, 10), np.percentile(ends, 90)))
