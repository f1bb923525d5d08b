"""Host cost of the partitioned sample chain, one rank over RCCL (every message to the rank
itself): samples only (no fetch), `depth` in flight over `lanes` lanes — wall time per sample and
the enqueue thread's busy time per sample (gf_worker_stats).  If the two agree the chain is bound
by the ONE thread that issues it (13 stream operations per sample), not by the GPU."""
import ctypes as C
import json
import os
import sys
import time
from collections import deque

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

import gnnflow_amd
from gnnflow_amd import _capi, synthetic
from gnnflow_amd.dist import DevicePartitionedSampler, PartitionedGraph

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
lib = _capi.load()
g = synthetic.reddit_like(seed=42)
MiB = 1 << 20
batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev))
           for r, t, _ in synthetic.replay_batches(g, 600, seed=42)]
torch.cuda.synchronize()


def worker():
    b, j = C.c_double(0), C.c_uint64(0)
    lib.gf_worker_stats(C.byref(b), C.byref(j))
    return b.value, j.value


for exchange in (False, True):
    for lanes in ((1,) if not exchange else (1, 2, 3, 4)):
        graph = gnnflow_amd.DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert")
        pg = PartitionedGraph(graph, 0, 1)
        for lo in range(0, g["num_edges"], 100000):
            hi = lo + 100000
            pg.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
        s = DevicePartitionedSampler(gnnflow_amd.TemporalSampler(graph, [10, 10], "recent"),
                                     always_exchange=exchange, slot_roots=1800, lanes=lanes)
        side = torch.cuda.Stream()
        depth = 2 * s.lanes
        for timed in (False, True):
            b0, j0 = worker()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            q = deque()
            n = 0
            for rep in range(3 if timed else 1):
                for r, t in batches:
                    if len(q) >= depth:
                        q.popleft().wait()
                    q.append(s.sample_async(r, t, stream=side, worker_enqueue=True))
                    n += 1
            while q:
                q.popleft().wait()
            torch.cuda.synchronize()
            wall = time.perf_counter() - t0
            b1, j1 = worker()
        print(json.dumps({"exchange": exchange, "lanes": s.lanes, "depth": depth,
                          "us_per_sample_wall": 1e6 * wall / n,
                          "enqueue_busy_us_per_sample": (b1 - b0) / max(j1 - j0, 1),
                          "jobs": j1 - j0}), flush=True)
        del s, graph
dist.destroy_process_group()
