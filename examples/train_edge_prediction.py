"""Minimal offline edge-prediction loop in the shape of the reference's
scripts/offline_edge_prediction.py, on this package only: batches from
`gnnflow_amd.data`, neighbourhoods from `TemporalSampler`, features through `LRUCache`, a
2-layer GraphSAGE (`gnnflow_amd.nn.SAGEConv`, the layer the reference's GRAPHSAGE model uses) and
a dot-product edge scorer.  Synthetic REDDIT-shaped data; a usage example, not a benchmark.

    python examples/train_edge_prediction.py [--batches 50]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.utils.data import DataLoader, SequentialSampler

import gnnflow_amd
from gnnflow_amd import nn as gnn
from gnnflow_amd import synthetic
from gnnflow_amd.cache import LRUCache
from gnnflow_amd.data import (EdgePredictionDataset, RandomStartBatchSampler,
                              default_collate_ndarray)
from gnnflow_amd.utils import DstRandEdgeSampler, build_dynamic_graph


class SAGE(nn.Module):
    def __init__(self, dim_in, dim_hidden):
        super().__init__()
        self.l0 = gnn.SAGEConv(dim_in, dim_hidden, 'mean')
        self.l1 = gnn.SAGEConv(dim_hidden, dim_hidden, 'mean')
        self.score = nn.Sequential(nn.Linear(dim_hidden, dim_hidden), nn.ReLU(),
                                   nn.Linear(dim_hidden, 1))

    def forward(self, mfgs):
        # mfgs[0] is the outer (largest) layer, mfgs[-1] the roots' layer
        h = F.relu(self.l0(mfgs[0][0], mfgs[0][0].srcdata['h']))
        h = self.l1(mfgs[1][0], h)
        b = h.shape[0] // 3                     # roots = [src | dst | negative dst]
        src, pos, neg = h[:b], h[b:2 * b], h[2 * b:]
        return self.score(src * pos), self.score(src * neg)


def main(num_batches=50, batch_size=600, seed=0, verbose=True):
    import pandas as pd
    torch.manual_seed(seed)
    dev = torch.device("cuda", 0)
    g = synthetic.reddit_like(seed=42, num_edges=60000)
    df = pd.DataFrame({"src": g["src"], "dst": g["dst"], "time": g["ts"], "eid": g["eid"]})
    graph = build_dynamic_graph(20 << 20, 1000 << 20, "cuda", 62, 1024, "insert",
                                undirected=True, device=0, dataset_df=df)
    sampler = gnnflow_amd.TemporalSampler(graph, fanouts=[10, 10], sample_strategy="recent")
    d = 32
    node_feats = torch.randn(g["num_nodes"], d)
    cache = LRUCache(0.0, 0.2, g["num_nodes"], g["num_edges"], dev, node_feats, None, d, 0)
    cache.init_cache()

    ds = EdgePredictionDataset(df, DstRandEdgeSampler(df["dst"].to_numpy(), seed=seed))
    loader = DataLoader(ds, sampler=RandomStartBatchSampler(SequentialSampler(ds), batch_size, False),
                        collate_fn=default_collate_ndarray, num_workers=0)
    model = SAGE(d, 64).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    losses = []
    for i, (roots, ts, eid) in enumerate(loader):
        if i >= num_batches:
            break
        mfgs = sampler.sample(roots, ts)
        cache.fetch_feature(mfgs, eid)
        pos, neg = model(mfgs)
        loss = F.binary_cross_entropy_with_logits(pos, torch.ones_like(pos)) + \
            F.binary_cross_entropy_with_logits(neg, torch.zeros_like(neg))
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if verbose and i % 10 == 0:
            print("batch {:4d} loss {:.4f} node-cache hit ratio {:.3f}".format(
                i, losses[-1], float(cache.cache_node_ratio)))
    return losses


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=50)
    main(ap.parse_args().batches)
