"""Seeded synthetic temporal graphs for parity tests (committed generator; no data
files).  Shapes follow SURVEY.md §8(d): power-law sources, chronological float32
timestamps with heavy ties, chunked ingestion."""
import numpy as np


def powerlaw_graph(num_nodes, num_edges, seed=42, alpha=1.0, t_max=1000.0, tie_levels=None):
    rng = np.random.RandomState(seed)
    ranks = np.arange(1, num_nodes + 1, dtype=np.float64)
    p = ranks ** (-alpha)
    p /= p.sum()
    perm = rng.permutation(num_nodes)
    src = perm[rng.choice(num_nodes, size=num_edges, p=p)].astype(np.int64)
    dst = rng.randint(0, num_nodes, size=num_edges).astype(np.int64)
    ts = np.sort(rng.uniform(0, t_max, size=num_edges)).astype(np.float32)
    if tie_levels:  # quantise to force many equal timestamps
        ts = (np.floor(ts / t_max * tie_levels) * (t_max / tie_levels)).astype(np.float32)
    eid = np.arange(num_edges, dtype=np.int64)
    return src, dst, ts, eid


def ingest_chunks(graph, src, dst, ts, eid, chunk, add_reverse=False):
    for i in range(0, len(src), chunk):
        graph.add_edges(src[i:i + chunk], dst[i:i + chunk], ts[i:i + chunk],
                        eid[i:i + chunk], add_reverse=add_reverse)


def random_roots(num_nodes, n, t_max, seed, extra_ids=()):
    rng = np.random.RandomState(seed)
    nodes = rng.randint(0, num_nodes, size=n).astype(np.int64)
    k = min(len(extra_ids), n)
    if k:
        nodes[:k] = np.asarray(extra_ids, dtype=np.int64)[:k]
    ts = rng.uniform(0, t_max * 1.1, size=n).astype(np.float32)
    return nodes, ts
