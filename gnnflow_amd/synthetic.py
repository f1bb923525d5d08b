"""Seeded synthetic temporal graphs shaped like the reference's datasets (no dataset
ships with the reference and there is no network).

reddit_like(): bipartite users -> items interaction stream with the published REDDIT
shape (10 984 node ids = 10 000 users + 984 items, 672 447 chronological edges,
float32 times in [0, 2.7e6], 172-d edge features); user activity and item popularity
are Zipf-distributed.  powerlaw(): SURVEY.md config 3 (N nodes / E directed edges,
Zipf sources, uniform destinations, sorted float32 times with heavy ties).
"""
import numpy as np

REDDIT = dict(num_users=10000, num_items=984, num_edges=672447, t_max=2.7e6,
              dim_edge=172, dim_node=172)


def _zipf_choice(rng, n, size, alpha):
    p = np.arange(1, n + 1, dtype=np.float64) ** (-alpha)
    p /= p.sum()
    return rng.permutation(n)[rng.choice(n, size=size, p=p)]


def reddit_like(num_edges=None, seed=42, user_alpha=0.9, item_alpha=1.0):
    cfg = REDDIT
    E = int(num_edges or cfg["num_edges"])
    rng = np.random.RandomState(seed)
    src = _zipf_choice(rng, cfg["num_users"], E, user_alpha).astype(np.int64)
    dst = (cfg["num_users"] +
           _zipf_choice(rng, cfg["num_items"], E, item_alpha)).astype(np.int64)
    ts = np.sort(rng.uniform(0, cfg["t_max"], size=E)).astype(np.float32)
    eid = np.arange(E, dtype=np.int64)
    return dict(src=src, dst=dst, ts=ts, eid=eid,
                num_nodes=cfg["num_users"] + cfg["num_items"], num_edges=E)


def powerlaw(num_nodes, num_edges, seed=42, alpha=1.0, t_max=1e6):
    rng = np.random.RandomState(seed)
    src = _zipf_choice(rng, num_nodes, num_edges, alpha).astype(np.int64)
    dst = rng.randint(0, num_nodes, size=num_edges).astype(np.int64)
    ts = np.sort(rng.uniform(0, t_max, size=num_edges).astype(np.float32))
    eid = np.arange(num_edges, dtype=np.int64)
    return dict(src=src, dst=dst, ts=ts, eid=eid, num_nodes=num_nodes, num_edges=num_edges)


def powerlaw_device(num_nodes, num_edges, device, seed=42, alpha=1.0, t_max=1e6):
    """powerlaw() generated on the GPU (inverse-CDF draws, seconds for 200 M edges instead of
    most of a minute in numpy): same distribution family, its own seeded stream.  Returns
    host numpy arrays (add_edges takes host arrays) plus the device tensors."""
    import torch
    g = torch.Generator(device=device).manual_seed(seed)
    w = torch.arange(1, num_nodes + 1, device=device, dtype=torch.float64) ** (-alpha)
    cdf = torch.cumsum(w / w.sum(), 0)
    perm = torch.randperm(num_nodes, generator=g, device=device)
    src = torch.empty(num_edges, dtype=torch.int64, device=device)
    step = 1 << 25
    for lo in range(0, num_edges, step):
        n = min(step, num_edges - lo)
        u = torch.rand(n, generator=g, device=device, dtype=torch.float64)
        r = torch.searchsorted(cdf, u).clamp_(max=num_nodes - 1)
        src[lo:lo + n] = perm[r]
    dst = torch.randint(0, num_nodes, (num_edges,), generator=g, device=device)
    ts = torch.sort(torch.rand(num_edges, generator=g, device=device) * t_max)[0].to(torch.float32)
    eid = torch.arange(num_edges, device=device, dtype=torch.int64)
    dev = dict(src=src, dst=dst, ts=ts, eid=eid)
    out = {k: v.cpu().numpy() for k, v in dev.items()}
    out.update(num_nodes=num_nodes, num_edges=num_edges, device=dev)
    return out


def replay_batches(graph, batch_size, seed=42):
    """benchmarks/benchmark_sampler.py:71-77: chronological replay; roots =
    [src || dst || uniform random node ids], timestamps = edge time x3."""
    rng = np.random.RandomState(seed)
    E = graph["num_edges"]
    for lo in range(0, E, batch_size):
        hi = min(lo + batch_size, E)
        n = hi - lo
        neg = rng.randint(0, graph["num_nodes"], n).astype(np.int64)
        roots = np.concatenate([graph["src"][lo:hi], graph["dst"][lo:hi], neg])
        ts = np.tile(graph["ts"][lo:hi], 3)
        yield roots, ts, graph["eid"][lo:hi]
