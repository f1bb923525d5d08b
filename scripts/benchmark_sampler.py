#!/usr/bin/env python3
"""The reference's sampler benchmark harness (benchmarks/benchmark_sampler.py:50-98) on
gnnflow_amd: chunked graph build (100 000 edges per add_edges), chronological replay in
batches of --batch_size with roots = [src || dst || random nodes], timing only the
`_sample` call, --repeat passes, mean +- std of *target edges/s* (= len(df) / time, the
harness's own unit; bench.py reports sampled edges/s).

Data: data/<DATASET>/edges.csv (columns src,dst,time[,ext_roll]; gnnflow/utils.py:60-75) when
present, otherwise the seeded REDDIT-shaped synthetic stream."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import gnnflow_amd  # noqa: E402
from gnnflow_amd import synthetic  # noqa: E402

MODEL_CONFIGS = {   # gnnflow/config.py:28-116 (sampler-relevant keys)
    "tgn": dict(fanouts=[10], sample_strategy="recent"),
    "tgat": dict(fanouts=[10, 10], sample_strategy="uniform"),
    "dysat": dict(fanouts=[10, 10], sample_strategy="uniform", num_snapshots=3,
                  snapshot_time_window=10000, prop_time=True),
    "graphsage": dict(fanouts=[15, 10], sample_strategy="uniform", is_static=True),
    "gat": dict(fanouts=[10, 10], sample_strategy="uniform", is_static=True),
}


def load_edges(dataset):
    path = os.path.join("data", dataset, "edges.csv")
    if os.path.exists(path):
        import pandas as pd
        df = pd.read_csv(path)
        return dict(src=df["src"].values.astype(np.int64), dst=df["dst"].values.astype(np.int64),
                    ts=df["time"].values.astype(np.float32),
                    eid=np.arange(len(df), dtype=np.int64),
                    num_nodes=int(max(df["src"].max(), df["dst"].max())) + 1,
                    num_edges=len(df)), path
    return synthetic.reddit_like(seed=42), "synthetic REDDIT-shaped stream (no data/ directory)"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dataset", default="REDDIT")
    ap.add_argument("--model", default="tgn")
    ap.add_argument("--batch_size", type=int, default=600)
    ap.add_argument("--ingestion-batch-size", type=int, default=100000)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--repeat", type=int, default=10)
    ap.add_argument("--sort", action="store_true")
    args = ap.parse_args()
    np.random.seed(args.seed)
    torch.manual_seed(args.seed)
    g, source = load_edges(args.dataset)
    MiB = 1 << 20
    graph = gnnflow_amd.DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert")
    for lo in range(0, g["num_edges"], args.ingestion_batch_size):
        hi = lo + args.ingestion_batch_size
        graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
    sampler = gnnflow_amd.TemporalSampler(graph, **MODEL_CONFIGS[args.model.lower()])
    num_nodes = graph.num_vertices()
    throughput = []
    for _ in range(args.repeat):
        total = 0.0
        for lo in range(0, g["num_edges"], args.batch_size):
            hi = min(lo + args.batch_size, g["num_edges"])
            roots = np.concatenate([g["src"][lo:hi], g["dst"][lo:hi],
                                    np.random.randint(0, num_nodes, hi - lo)]).astype(np.int64)
            ts = np.tile(g["ts"][lo:hi], 3).astype(np.float32)
            t0 = time.time()
            _, sort_time = sampler._sample(roots, ts, sort=args.sort)
            total += time.time() - t0 - sort_time
        throughput.append(g["num_edges"] / total)
    print("data: {}".format(source))
    print("Throughput for {}'s sampling on {}: {:.2f} samples/s, std: {:.2f}, std/mean: {:.2f}".format(
        args.model, args.dataset, np.mean(throughput), np.std(throughput),
        np.std(throughput) / np.mean(throughput)))


if __name__ == "__main__":
    main()
