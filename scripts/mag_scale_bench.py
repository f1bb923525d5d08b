"""BASELINE config 5's GRAPH at full size on ONE MI355X: MAG-shaped synthetic stream — 121 751 665
nodes, 1 297 748 926 edges (the reference partitions it over 8 GPUs) — built with `add_edges`
in 10^7-edge chunks and sampled with the TGN setting (1 layer, fanout [10], most-recent,
batch 4000 = 12 000 roots; gnnflow/config.py:28-43,169-179: minimum block 11).  The 768-d node
features (374 GB) do not fit 288 GB of HBM and are not part of this run: it measures the edge
store and the sampler at that size.

Prints one JSON line: build time / rate, HBM in use, µs per sample() and sampled edges/s over
the LAST `--batches` batches of the chronological replay, and a spot check of one block
against a brute-force scan of the edge list.

  python scripts/mag_scale_bench.py                 # full size (~100 GB of HBM, ~60 GB of host RAM)
  python scripts/mag_scale_bench.py --edges 100000000 --nodes 10000000
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import gnnflow_amd                      # noqa: E402
from gnnflow_amd import synthetic       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=121_751_665)
    ap.add_argument("--edges", type=int, default=1_297_748_926)
    ap.add_argument("--batch", type=int, default=4000)
    ap.add_argument("--batches", type=int, default=2000)
    ap.add_argument("--fanout", type=int, default=10)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    from gnnflow_amd.utils import bind_to_device_cpus
    bind_to_device_cpus(0)
    N, E = args.nodes, args.edges
    t0 = time.time()
    g = synthetic.powerlaw_device(N, E, dev, seed=42, alpha=1.0, t_max=1e6)
    gen_s = time.time() - t0
    del g["device"]                       # the device copies of the edge list are not needed
    torch.cuda.empty_cache()
    graph = gnnflow_amd.DynamicGraph(1 << 30, 250 << 30, "cuda", 11, 1024, "insert")
    t0 = time.time()
    for lo in range(0, E, 10_000_000):
        hi = min(E, lo + 10_000_000)
        graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
    build_s = time.time() - t0
    free, total = torch.cuda.mem_get_info(dev)
    sampler = gnnflow_amd.TemporalSampler(graph, [args.fanout], "recent", seed=1234)
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
        dbatches.append((torch.from_numpy(roots).to(dev), torch.from_numpy(ts).to(dev)))
    for r, t in dbatches[:50]:
        sampler.sample(r, t)
    torch.cuda.synchronize()
    edges = 0
    t0 = time.time()
    pending = []
    for r, t in dbatches:            # two samples in flight, like the replay pipeline
        pending.append(sampler.sample_async(r, t))
        if len(pending) > 2:
            edges += sum(b.num_edges() for mfg in pending.pop(0).wait() for b in mfg)
    for p in pending:
        edges += sum(b.num_edges() for mfg in p.wait() for b in mfg)
    torch.cuda.synchronize()
    dt = time.time() - t0
    # spot check: the first root of the last batch against a scan of the whole edge list
    r, t = dbatches[-1]
    blk = sampler.sample(r, t)[0][0]
    root, root_t = int(r[0]), float(t[0])
    mine = np.nonzero((g["src"] == root) & (g["ts"] < root_t))[0][-args.fanout:][::-1]
    row = blk.edges()[1].cpu().numpy()
    got = blk.edata["ID"].cpu().numpy()[row == 0]
    ok = bool(np.array_equal(got, g["eid"][mine]))
    print(json.dumps({
        "workload": "MAG-shaped synthetic graph, TGN sampling (1 layer, fanout [{}], most recent), "
                    "batch {} ({} roots), one GPU, sampler only".format(
                        args.fanout, args.batch, 3 * args.batch),
        "nodes": N, "edges": E, "gen_s": round(gen_s, 2), "graph_build_s": round(build_s, 2),
        "graph_build_Medges_per_s": round(E / build_s / 1e6, 1),
        "hbm_in_use_GB_after_build": round((total - free) / 1e9, 1),
        "graph_memory_GB_logical": round(graph.get_graph_memory_usage() / 1e9, 1)
        if hasattr(graph, "get_graph_memory_usage") else None,
        "steps": nb, "us_per_sample": round(dt / nb * 1e6, 1),
        "sampled_edges_per_step": round(edges / nb, 1), "sampled_edges_per_s": round(edges / dt),
        "first_root_matches_edge_list_scan": ok,
    }), flush=True)


if __name__ == "__main__":
    main()
