"""Host-time breakdown of the pipelined loop over the hash-partitioned graph, one rank over RCCL
(--always-exchange: every message to the rank itself): where the Python thread's time goes per
step (waiting for the sample, sample_async, record_stream, fetch_feature) and the enqueue
threads' busy time.  Diagnostic for DESIGN 6 (lanes)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist
import gnnflow_amd
from gnnflow_amd import _capi, synthetic
from gnnflow_amd.cache import LRUCache
from gnnflow_amd.dist import DevicePartitionedSampler, PartitionedGraph
ap = argparse.ArgumentParser()
ap.add_argument("--lanes", type=int, default=3)
ap.add_argument("--depth", type=int, default=0)
ap.add_argument("--chain", type=int, default=None)
ap.add_argument("--profile", action="store_true", help="cProfile of the last repetition (main thread)")
ap.add_argument("--no-exchange", action="store_true")
ap.add_argument("--replica", action="store_true", help="the plain sampler over the whole graph (the N = 1 headline loop) for comparison")
args = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev, rank=0, world_size=1)
g = synthetic.reddit_like(seed=42)
graph = gnnflow_amd.DynamicGraph(20 << 20, 1000 << 20, "cuda", 62, 1024, "insert")
pg = PartitionedGraph(graph, 0, 1)
for lo in range(0, g["num_edges"], 100000):
    pg.add_edges(g["src"][lo:lo+100000], g["dst"][lo:lo+100000], g["ts"][lo:lo+100000], g["eid"][lo:lo+100000])
sampler = DevicePartitionedSampler(gnnflow_amd.TemporalSampler(graph, [10, 10]), always_exchange=not args.no_exchange,
                                   slot_roots=1800, lanes=args.lanes, chain_samples=args.chain)
if args.replica:
    sampler = gnnflow_amd.TemporalSampler(graph, [10, 10])
    sampler.lanes, sampler.chain_samples = 1, 1
ef = torch.rand((g["num_edges"], 172), device=dev); nf = torch.rand((g["num_nodes"], 172), device=dev)
cache = LRUCache(0.2, 0.2, g["num_nodes"], g["num_edges"], dev, nf, ef, 172, 172); cache.init_cache()
batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev), torch.from_numpy(e).to(dev))
           for r, t, e in list(synthetic.replay_batches(g, 600))]
side = torch.cuda.Stream(device=dev); main = torch.cuda.current_stream(dev)
from collections import deque
T = dict(wait=0.0, begin=0.0, rec=0.0, fetch=0.0, wq=0.0)
orig_wait = cache.wait_enqueued
T["wq_c"] = T["wq_free"] = 0.0
DROP = [0.0] * 8
def timed_wait(upto=None):      # Cache.wait_enqueued with its two halves timed apart
    q = cache._tickets
    while q and (upto is None or q[0][0] <= upto):
        t0 = time.perf_counter()
        ticket, refs = q.popleft()
        _capi.check(cache._lib.gf_cache_fetch_wait(ticket))
        t1 = time.perf_counter()
        refs = list(refs)       # (jobs, descs, cdescs, mfgs, out_all, stats ring)
        while refs:
            ta = time.perf_counter()
            refs.pop()
            DROP[len(refs)] += time.perf_counter() - ta
        t2 = time.perf_counter()
        T["wq"] += t2 - t0; T["wq_c"] += t1 - t0; T["wq_free"] += t2 - t1
cache.wait_enqueued = timed_wait
DEPTH = args.depth or (sampler.lanes + 1)
import ctypes as C
lib = _capi.load()
def workers():
    bu, jb = C.c_double(0), C.c_uint64(0); lib.gf_worker_stats(C.byref(bu), C.byref(jb)); return bu.value, jb.value
lib.gf_debug_part_host_us((C.c_double * 8)(), 1)
import cProfile, pstats
for rep in range(3):
    prof = cProfile.Profile() if (args.profile and rep == 2) else None
    for k in T: T[k] = 0.0
    cache.init_cache()
    b0, j0 = workers()
    torch.cuda.synchronize(); t00 = time.perf_counter()
    if prof: prof.enable()
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
    if prof:
        prof.disable()
        pstats.Stats(prof).sort_stats("tottime").print_stats(32)
    b1, j1 = workers()
    n = len(batches)
    print("  dropping (jobs, descs, cdescs, mfgs, out_all, stats ring) us/step:", [round(1e6 * d / n, 2) for d in DROP[:6]])
    for k in range(8): DROP[k] = 0.0
    st = (C.c_double * 8)(); lib.gf_debug_part_host_us(st, 1)
    if st[7]:
        print("  issuing thread per sample us: begin %.1f plan %.1f a2a-req %.1f serve %.1f a2a-rep %.1f merge %.1f commit %.1f = %.1f" % (
            *[st[i] / st[7] for i in range(7)], sum(st[i] for i in range(7)) / st[7]))
    print("lanes %d chain %d depth %d per step us: total %.1f | sample.wait %.1f | sample_async %.1f | record_stream %.1f | fetch_feature %.1f (of which wait_enqueued %.1f = native wait %.1f + dropping the fetch's references %.1f) | enqueue threads busy %.1f us/step over %.1f jobs/step" % (
        sampler.lanes, sampler.chain_samples, DEPTH, 1e6*tot/n, 1e6*T["wait"]/n, 1e6*T["begin"]/n, 1e6*T["rec"]/n, 1e6*T["fetch"]/n, 1e6*T["wq"]/n, 1e6*T["wq_c"]/n, 1e6*T["wq_free"]/n,
        (b1 - b0) / n, (j1 - j0) / n), flush=True)
dist.destroy_process_group()
