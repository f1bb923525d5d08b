"""Seconds per training epoch of a TGN-shaped model on the MAG-shaped graph at 1/100 scale —
the stand-in for BASELINE config 5's "end-to-end offline_edge_prediction.py epoch time".

The reference's script and its model classes cannot travel to the GPU box (and need dgl), so
this is the builder's OWN loop in the shape of scripts/offline_edge_prediction.py:343-454 —
sample (prefetched two batches ahead, roots resident in HBM) -> fetch_feature -> memory.prepare_input -> memory updater
-> model -> update_mem_mail -> loss / backward / step — over this package only:
`TemporalSampler` (1 layer, fanout [10], most-recent, batch 4000 = 12 000 roots),
`LRUCache` ratio 0.2 over 768-d node features, `gnnflow_amd.memory.Memory` (dim 100), and a
TGN-shaped model written here on `gnnflow_amd.ops` (time encoding, GRU memory updater, one
2-head attention layer over the sampled block, edge scorer: the compute the reference's
config.py:28-43 TGN does per batch; NOT the reference's classes).  One JSON line: seconds per
epoch and the split sample / fetch / memory gather / memory update / model / write-back, as the
reference's loop accumulates them (the sample time is what the prefetch does not hide).

    python examples/tgn_epoch.py [--nodes 1220000 --edges 13000000 --epochs 2]
"""
import argparse
from collections import deque
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

import gnnflow_amd
from gnnflow_amd import ops, synthetic
from gnnflow_amd.cache import LRUCache
from gnnflow_amd.memory import Memory


class TimeEncoding(nn.Module):
    """cos(w * dt + b) with fixed geometric frequencies as initial weights (TGAT)."""

    def __init__(self, dim):
        super().__init__()
        self.lin = nn.Linear(1, dim)
        with torch.no_grad():
            self.lin.weight.copy_(torch.from_numpy(
                1.0 / 10 ** np.linspace(0, 9, dim, dtype=np.float32)).reshape(dim, 1))
            self.lin.bias.zero_()

    def forward(self, dt):
        return torch.cos(self.lin(dt.reshape(-1, 1)))


class MemoryUpdater(nn.Module):
    """GRU cell over [mailbox message || time encoding of (now - last update)]; the updated
    memory (plus a projection of the node features) becomes the block's node embedding."""

    def __init__(self, dim_node, dim_edge, dim_time, dim_memory):
        super().__init__()
        self.time = TimeEncoding(dim_time)
        self.cell = nn.GRUCell(2 * dim_memory + dim_edge + dim_time, dim_memory)
        self.proj = nn.Linear(dim_node, dim_memory) if dim_node else None

    def forward(self, b):
        x = torch.cat([b.srcdata['mem_input'], self.time(b.srcdata['ts'] - b.srcdata['mem_ts'])], 1)
        new_mem = self.cell(x, b.srcdata['mem'])
        R = b.num_dst_nodes()
        last = dict(last_updated_nid=b.srcdata['ID'][:R].detach(),
                    last_updated_memory=new_mem[:R].detach().clone(),
                    last_updated_ts=b.srcdata['ts'][:R].detach())
        b.srcdata['h'] = new_mem + self.proj(b.srcdata['h']) if self.proj is not None else new_mem
        return last


class Attention(nn.Module):
    """One multi-head attention layer over a sampled block: queries from the destination
    nodes (zero time encoding), keys / values from the sampled neighbours with the time
    encoding of the edge's age; softmax over each destination's in-edges, weighted sum."""

    def __init__(self, dim_node, dim_time, dim_out, heads):
        super().__init__()
        self.heads = heads
        self.time = TimeEncoding(dim_time)
        self.q = nn.Linear(dim_node + dim_time, dim_out)
        self.k = nn.Linear(dim_node + dim_time, dim_out)
        self.v = nn.Linear(dim_node + dim_time, dim_out)
        self.out = nn.Linear(dim_node + dim_out, dim_out)
        self.norm = nn.LayerNorm(dim_out)
        self.dim_out = dim_out

    def forward(self, b):
        R, E = b.num_dst_nodes(), b.num_edges()
        h = b.srcdata['h']
        if E == 0:
            return torch.zeros((R, self.dim_out), device=h.device)
        dst_h, src_h = h[:R], h[R:]
        row = b.edges()[1]
        q = self.q(torch.cat([dst_h, self.time(torch.zeros(R, device=h.device))], 1))[row]
        kv_in = torch.cat([src_h, self.time(b.edata['dt'])], 1)
        k, v = self.k(kv_in), self.v(kv_in)
        H = self.heads
        score = F.leaky_relu((q.view(E, H, -1) * k.view(E, H, -1)).sum(2), 0.2)
        att = ops.edge_softmax(b, score)                       # [E, H]
        msg = (v.view(E, H, -1) * att[:, :, None]).reshape(E, -1)
        # the sampler's blocks have source node R + k for edge k: pad the roots with zeros
        agg = ops.block_reduce(b, torch.cat([torch.zeros((R, msg.shape[1]), device=h.device), msg]))
        return self.norm(F.relu(self.out(torch.cat([agg, dst_h], 1))))


class EdgeScorer(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.src, self.dst, self.out = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, 1)

    def forward(self, h):
        s, p, n = h.tensor_split(3)
        s = self.src(s)
        return self.out(F.relu(s + self.dst(p))), self.out(F.relu(s + self.dst(n)))


class TGN(nn.Module):
    def __init__(self, dim_node, dim_time=100, dim_embed=100, dim_memory=100, heads=2):
        super().__init__()
        self.updater = MemoryUpdater(dim_node, 0, dim_time, dim_memory)
        self.att = Attention(dim_memory, dim_time, dim_embed, heads)
        self.score = EdgeScorer(dim_embed)

    def forward(self, mfgs):
        return self.score(self.att(mfgs[0][0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_220_000)
    ap.add_argument("--edges", type=int, default=13_000_000)
    ap.add_argument("--dim-node", type=int, default=768)
    ap.add_argument("--batch", type=int, default=4000)
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--train-frac", type=float, default=0.7)
    ap.add_argument("--max-batches", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    N, E, B = args.nodes, args.edges, args.batch
    t0 = time.time()
    g = synthetic.powerlaw_device(N, E, dev, seed=5, alpha=1.0, t_max=1e5)
    gd = g.pop("device")        # the same arrays, resident in HBM
    # gnnflow/config.py:169-179 (MAG): minimum block 11
    graph = gnnflow_amd.DynamicGraph(256 << 20, 16 << 30, "cuda", 11, 65536, "insert")
    for lo in range(0, E, 5_000_000):
        hi = min(E, lo + 5_000_000)
        graph.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
    build_s = time.time() - t0
    sampler = gnnflow_amd.TemporalSampler(graph, [10], "recent")
    node_feats = torch.rand((N, args.dim_node), device=dev)
    cache = LRUCache(0.0, 0.2, N, E, dev, node_feats, None, args.dim_node, 0)
    cache.init_cache()
    model = TGN(args.dim_node).to(dev)
    memory = Memory(N, 0, 100, dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    n_train = int(E * args.train_frac)
    nb = n_train // B
    if args.max_batches:
        nb = min(nb, args.max_batches)
    side = torch.cuda.Stream(device=dev)
    main_stream = torch.cuda.current_stream(dev)
    gen = torch.Generator(device=dev).manual_seed(0)
    depth = 2                       # samples begun ahead of the batch being trained on

    def epoch_roots():
        """Roots / timestamps of EVERY batch of the split, resident in HBM: [nb, 3 B] =
        [src || dst || negatives] (gnnflow/utils.py:385-391), drawn once per epoch on the
        device.  (Until round 5 each batch's roots were assembled in numpy and copied H2D on
        the training stream, i.e. BEHIND the previous batch's model kernels: 1.0 ms of every
        batch's "sample" time.)"""
        n = nb * B
        neg = torch.randint(0, N, (nb, B), device=dev, generator=gen, dtype=torch.int64)
        roots = torch.cat([gd["src"][:n].view(nb, B), gd["dst"][:n].view(nb, B), neg], dim=1)
        ts = gd["ts"][:n].view(nb, B).repeat(1, 3)
        return roots.contiguous(), ts.contiguous()

    out = []
    for epoch in range(args.epochs):
        memory.reset()
        cache.init_cache()
        T = dict(sample=0.0, fetch=0.0, mem_gather=0.0, mem_update=0.0, model=0.0, write_back=0.0)
        edges = 0
        torch.cuda.synchronize()
        e0 = time.perf_counter()
        roots, rts = epoch_roots()
        side.wait_stream(main_stream)               # the sampler reads them on the side stream
        pending = deque(sampler.sample_async(roots[j], rts[j], stream=side, worker_enqueue=True)
                        for j in range(min(depth, nb)))
        for i in range(nb):
            t0 = time.perf_counter()
            mfgs = pending.popleft().wait()           # what the prefetch did not hide
            if i + depth < nb:
                pending.append(sampler.sample_async(roots[i + depth], rts[i + depth], stream=side,
                                                    worker_enqueue=True))
            for mfg in mfgs:
                for b in mfg:
                    b.record_stream(main_stream)
            t1 = time.perf_counter()
            cache.fetch_feature(mfgs, None)
            t2 = time.perf_counter()
            b = mfgs[0][0]
            memory.prepare_input(b)
            t3 = time.perf_counter()
            last = model.updater(b)
            t4 = time.perf_counter()
            opt.zero_grad()
            pos, neg = model(mfgs)
            t5 = time.perf_counter()
            with torch.no_grad():
                memory.update_mem_mail(**last, edge_feats=None, neg_sample_ratio=1)
            t6 = time.perf_counter()
            loss = F.binary_cross_entropy_with_logits(pos, torch.ones_like(pos)) + \
                F.binary_cross_entropy_with_logits(neg, torch.zeros_like(neg))
            loss.backward()
            opt.step()
            t7 = time.perf_counter()
            T["sample"] += t1 - t0
            T["fetch"] += t2 - t1
            T["mem_gather"] += t3 - t2
            T["mem_update"] += t4 - t3
            T["model"] += (t5 - t4) + (t7 - t6)
            T["write_back"] += t6 - t5
            edges += b.num_edges()
        torch.cuda.synchronize()
        total = time.perf_counter() - e0
        out.append(dict(epoch=epoch, seconds=total, batches=nb, ms_per_batch=1e3 * total / nb,
                        sampled_edges=edges, loss=float(loss.detach()),
                        node_cache_hit_ratio=float(cache.cache_node_ratio),
                        host_seconds_by_stage={k: round(v, 3) for k, v in T.items()}))
    print(json.dumps({
        "what": "TGN-shaped training epoch, the builder's own loop (stand-in for config 5's "
                "offline_edge_prediction.py epoch; NOT the reference's script or model classes)",
        "graph": "MAG-shaped synthetic at 1/100 scale: {} nodes, {} edges, {}-d node features, "
                 "minimum block 11".format(N, E, args.dim_node),
        "config": "TGN: 1 layer, fanout [10], most-recent, batch {} ({} roots), memory dim 100, "
                  "time dim 100, embed dim 100, 2 heads, LRUCache node ratio 0.2, Adam; one GPU; "
                  "train split = first {:.0%} of the edges".format(B, 3 * B, args.train_frac),
        "stages": "host wall time per stage as scripts/offline_edge_prediction.py:403-454 "
                  "accumulates it (stages are asynchronous: GPU time shows up where the host "
                  "next waits — mostly in `model`, whose loss read-back synchronises)",
        "graph_build_s": round(build_s, 2), "epochs": out}))


if __name__ == "__main__":
    main()
