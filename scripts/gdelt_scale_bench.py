"""BASELINE config 4's data at FULL size on ONE MI355X (the reference shards it over 8 GPUs because
a 40/80 GB card cannot hold it): GDELT-shaped synthetic stream — 16 682 nodes, 191 290 882 edges,
186-d edge features (142 GB) and 413-d node features resident in HBM — TGAT sampling (2 layers,
fanout [10, 10], uniform, minimum block 123; gnnflow/config.py:45-59,157-167), LRU cache ratio
0.2 (38 M edge slots = 28 GB: the replacement order is kept in the queue form),
chronological replay of the LAST `--batches` batches of 600 edges through the pipelined loop
bench.py times (gnnflow_amd.pipeline.ReplayPipeline).

Prints one JSON line: µs per step, sampled edges/s, cache hit ratios, LRU state, HBM in use.
Feature rows are a function of their id, so every fetched `f` / `h` row of the last step is
checked on the device.

  python scripts/gdelt_scale_bench.py                      # full size (≈190 GB of HBM)
  python scripts/gdelt_scale_bench.py --edges 20000000     # a tenth, for a quick look
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import gnnflow_amd                                  # noqa: E402
from gnnflow_amd import synthetic                   # noqa: E402
from gnnflow_amd.cache import LRUCache              # noqa: E402
from gnnflow_amd.pipeline import ReplayPipeline     # noqa: E402


def feature_table(rows, dim, dev, chunk=1 << 22):
    """table[i, c] = frac(i * 0.6180339887) + c / 1024: cheap to make, cheap to check."""
    t = torch.empty((rows, dim), dtype=torch.float32, device=dev)
    col = torch.arange(dim, device=dev, dtype=torch.float32) / 1024.0
    for lo in range(0, rows, chunk):
        hi = min(rows, lo + chunk)
        i = torch.arange(lo, hi, device=dev, dtype=torch.float64)
        t[lo:hi] = torch.frac(i * 0.6180339887).to(torch.float32)[:, None] + col[None, :]
    return t


def expected_rows(ids, dim):
    col = torch.arange(dim, device=ids.device, dtype=torch.float32) / 1024.0
    return torch.frac(ids.to(torch.float64) * 0.6180339887).to(torch.float32)[:, None] + col[None, :]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=16682)
    ap.add_argument("--edges", type=int, default=191_290_882)
    ap.add_argument("--dim-edge", type=int, default=186)
    ap.add_argument("--dim-node", type=int, default=413)
    ap.add_argument("--ratio", type=float, default=0.2)
    ap.add_argument("--batch", type=int, default=600)
    ap.add_argument("--batches", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--policy", default="uniform")
    ap.add_argument("--feature-placement", default="device", choices=["device", "pinned"],
                    help="pinned: the reference's placement — both tables in pinned HOST memory, the "
                         "LRU caches in HBM, the coming misses pulled into the staging ring")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from gnnflow_amd.utils import bind_to_device_cpus
    bind_to_device_cpus(0)   # host arrays and ingest threads on the GPU's NUMA node
    N, E = args.nodes, args.edges
    t0 = time.time()
    g = synthetic.powerlaw_device(N, E, dev, seed=42, alpha=1.0, t_max=1e6)
    gen_s = time.time() - t0
    graph = gnnflow_amd.DynamicGraph(1 << 30, 200 << 30, "cuda", 123, 1024, "insert")
    t0 = time.time()
    for lo in range(0, E, 10_000_000):
        hi = min(E, lo + 10_000_000)
        graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
    build_s = time.time() - t0
    t0 = time.time()
    if args.feature_placement == "pinned":
        # made on the GPU in pieces, kept in pinned host memory
        efeat = torch.empty((E, args.dim_edge), dtype=torch.float32, pin_memory=True)
        step = 1 << 24
        for lo in range(0, E, step):
            hi = min(E, lo + step)
            i = torch.arange(lo, hi, device=dev, dtype=torch.float64)
            col = torch.arange(args.dim_edge, device=dev, dtype=torch.float32) / 1024.0
            efeat[lo:hi].copy_(torch.frac(i * 0.6180339887).to(torch.float32)[:, None] + col[None, :])
        nfeat = feature_table(N, args.dim_node, dev).cpu().pin_memory()
    else:
        efeat = feature_table(E, args.dim_edge, dev)
        nfeat = feature_table(N, args.dim_node, dev)
    cache = LRUCache(args.ratio, args.ratio, N, E, dev, nfeat, efeat, args.dim_node, args.dim_edge,
                     feature_placement=args.feature_placement)
    cache.init_cache()
    torch.cuda.synchronize()
    feat_s = time.time() - t0
    sampler = gnnflow_amd.TemporalSampler(graph, [10, 10], args.policy, seed=1234)
    # the last `batches` batches of the chronological replay (the graph is complete: sampling
    # at a batch's timestamps sees exactly the edges older than them)
    nb = min(args.batches, E // args.batch)
    first_edge = E - nb * args.batch
    rng = np.random.RandomState(42)
    dbatches = []
    for b in range(nb):
        lo = first_edge + b * args.batch
        hi = lo + args.batch
        neg = rng.randint(0, N, args.batch).astype(np.int64)
        roots = np.concatenate([g["src"][lo:hi], g["dst"][lo:hi], neg])
        ts = np.tile(g["ts"][lo:hi], 3)
        dbatches.append((torch.from_numpy(roots).to(dev), torch.from_numpy(ts).to(dev),
                         torch.from_numpy(g["eid"][lo:hi]).to(dev)))
    pipe = ReplayPipeline(sampler, cache, dbatches, dev, pipelined=True)
    edges = [0]
    last = [None]

    def on_step(i, mfgs):
        edges[0] += sum(b.num_edges() for mfg in mfgs for b in mfg)
        last[0] = mfgs

    warm = min(args.warmup, nb // 4)
    pipe.run(0, warm, None)
    torch.cuda.synchronize()
    # HIP events on every 7th gather launch of the timed region (bench.py's mechanism); the
    # algorithmic bytes of the fetches: rows * row bytes, read + written
    import ctypes as C
    from gnnflow_amd import _capi
    lib = _capi.load()
    lib.gf_profile_reset()
    lib.gf_profile_set_stride(7)
    lib.gf_profile_enable(1 << _capi.PROFILE_SLOTS["gather"])
    cache.algorithmic_bytes = 0
    st0 = cache.staging_state() if cache.staging else None
    t0 = time.time()
    pipe.run(warm, nb - warm, on_step)
    torch.cuda.synchronize()
    dt = time.time() - t0
    st1 = cache.staging_state() if cache.staging else None
    lib.gf_profile_enable(0)
    lib.gf_profile_set_stride(1)
    g_ms, g_n, g_all = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
    lib.gf_profile_get(_capi.PROFILE_SLOTS["gather"], C.byref(g_ms), C.byref(g_n))
    lib.gf_profile_launches(_capi.PROFILE_SLOTS["gather"], C.byref(g_all))
    steps = nb - warm
    gather_us = g_ms.value * 1e3 / max(1, g_n.value)            # per launch, sampled
    gather_bytes = cache.algorithmic_bytes / max(1, g_all.value)  # per launch, all of them
    # every row fetched in the last step equals its table row
    ok = True
    for k, mfg in enumerate(last[0]):
        for b in mfg:
            if b.num_edges():
                ok &= bool(torch.equal(b.edata["f"], expected_rows(b.edata["ID"], args.dim_edge)))
            if k == 0:
                ok &= bool(torch.equal(b.srcdata["h"], expected_rows(b.srcdata["ID"], args.dim_node)))
    free, total = torch.cuda.mem_get_info(dev)
    print(json.dumps({
        "workload": "GDELT-shaped synthetic, TGAT sampling (2 layers, fanout [10,10], {}), batch {}, "
                    "LRU {} + {}-d edge / {}-d node feature gather, one GPU".format(
                        args.policy, args.batch, args.ratio, args.dim_edge, args.dim_node),
        "nodes": N, "edges": E, "steps": steps, "us_per_step": round(dt / steps * 1e6, 1),
        "sampled_edges_per_step": round(edges[0] / steps, 1),
        "sampled_edges_per_s": round(edges[0] / dt),
        "gather": {"launches_per_step": round(g_all.value / steps, 2), "timed_launches": int(g_n.value),
                   "us_per_launch": round(gather_us, 1),
                   "algorithmic_MB_per_launch": round(gather_bytes / 1e6, 1),
                   "GB_per_s": round(gather_bytes / max(gather_us, 1e-9) / 1e3, 1),
                   "frac_of_8TBps": round(gather_bytes / max(gather_us, 1e-9) / 1e3 / 8000.0, 3)},
        "cache_edge_ratio": round(float(cache.cache_edge_ratio), 4),
        "cache_node_ratio": round(float(cache.cache_node_ratio), 4),
        "edge_cache_slots": cache.edge_capacity,
        "edge_lru_state": cache._edge.lru_state(),
        "rows_checked_equal_table": ok,
        "feature_placement": args.feature_placement,
        "staging": None if st1 is None else {
            k: {"generations": st1[k]["generations"], "rows_per_generation": st1[k]["rows_per_generation"],
                "ring_GB": round(st1[k]["ring_bytes"] / 1e9, 2),
                "rows_pulled_per_step": round((st1[k]["rows_pulled"] - st0[k]["rows_pulled"]) / steps, 1),
                "rows_read_from_host_by_gather_per_step":
                    round((st1[k]["rows_read_from_host"] - st0[k]["rows_read_from_host"]) / steps, 1),
                "host_link_MB_per_step": round(
                    ((st1[k]["rows_pulled"] - st0[k]["rows_pulled"]) +
                     (st1[k]["rows_read_from_host"] - st0[k]["rows_read_from_host"])) / steps *
                    4 * (args.dim_node if k == "node" else args.dim_edge) / 1e6, 2),
                "generations_dropped": st1[k]["dropped"] - st0[k]["dropped"]} for k in st1},
        "gen_s": round(gen_s, 2), "graph_build_s": round(build_s, 2),
        "graph_build_Medges_per_s": round(E / build_s / 1e6, 1), "features_s": round(feat_s, 2),
        "edge_feature_table_GB": round(E * args.dim_edge * 4 / 1e9, 1),
        "hbm_in_use_GB": round((total - free) / 1e9, 1), "hbm_total_GB": round(total / 1e9, 1),
    }), flush=True)


if __name__ == "__main__":
    main()
