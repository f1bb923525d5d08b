"""Extended fuzz of the edge store's ingest against the oracle: random graphs, random chunk
sizes on both sides of the device-ordering threshold (forced low), both insertion policies,
adaptive block size on / off, reverse edges, an offload in the middle, a rejected batch — after
every few chunks the accessors, block-policy counters and per-vertex neighbour lists are
compared, and a sample at the end.

  python scripts/fuzz_ingest.py [--seeds 30]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GNNFLOW_INGEST_DEVICE_SORT_MIN", "20000")   # device ordering from 20 k edges

from tests import synth   # noqa: E402


def compare(g, o, rng, tag):
    assert g.num_edges() == o.num_edges(), tag
    assert g.num_vertices() == o.num_vertices(), tag
    assert g.num_source_vertices() == o.num_source_vertices(), tag
    assert g.max_vertex_id() == o.max_vertex_id(), tag
    assert np.array_equal(g.nodes(), o.nodes()), tag
    assert np.array_equal(g.src_nodes(), o.src_nodes()), tag
    ids = np.arange(0, o.max_vertex_id() + 1)
    assert np.array_equal(g.out_degree(ids), o.out_degree(ids)), tag
    assert abs(g.avg_linked_list_length() - o.avg_linked_list_length()) < 1e-4, tag
    assert g.get_graph_memory_usage() == o.get_graph_memory_usage(), tag
    for v in rng.choice(ids, size=min(60, len(ids)), replace=False):
        for a, b in zip(g.get_temporal_neighbors(int(v)), o.get_temporal_neighbors(int(v))):
            assert np.array_equal(a, b), (tag, int(v))


def run(seed):
    import gnnflow_amd
    from oracle import oracle as O
    rng = np.random.RandomState(seed)
    N = int(rng.choice([50, 2000, 60000]))
    E = int(rng.choice([3000, 60000, 400000]))
    policy = str(rng.choice(["insert", "replace"]))
    adaptive = bool(rng.randint(2))
    min_block = int(rng.choice([1, 4, 16, 62]))
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=seed, tie_levels=int(rng.choice([50, 5000])))
    g = gnnflow_amd.DynamicGraph(1 << 20, 4096 << 20, "cuda", min_block, 128, policy,
                                 adaptive_block_size=adaptive)
    o = O.OracleGraph(minimum_block_size=min_block, insertion_policy=policy,
                      adaptive_block_size=adaptive)
    rev = bool(rng.randint(2))
    lo, k, offloaded = 0, 0, False
    while lo < E:
        n = int(rng.choice([1, 37, 900, 15000, 30000, 150000]))
        hi = min(E, lo + n)
        for gr in (g, o):
            gr.add_edges(src[lo:hi], dst[lo:hi], ts[lo:hi], eid[lo:hi], add_reverse=rev)
        lo = hi
        k += 1
        if k % 7 == 0:
            compare(g, o, rng, "seed %d chunk %d" % (seed, k))
        if not offloaded and lo > E // 2 and rng.randint(3) == 0:
            t = float(ts[lo // 3])
            assert g.offload_old_blocks(t) == o.offload_old_blocks(t)
            offloaded = True
            compare(g, o, rng, "seed %d after offload" % seed)
        if k == 5:   # a batch older than what is stored: rejected, nothing changes
            try:
                g.add_edges(src[:3], dst[:3], ts[:3] - 1e6, eid[:3])
                raise SystemExit("old batch was accepted")
            except ValueError:
                pass
    compare(g, o, rng, "seed %d end" % seed)
    hs = gnnflow_amd.TemporalSampler(g, [5, 5], "recent")
    os_ = O.OracleSampler(o, [5, 5], "recent")
    nodes, t = synth.random_roots(N, 500, float(ts[-1]) + 1.0, seed=seed)
    for hl, ol in zip(hs.sample(nodes, t), os_.sample(nodes, t)):
        for hb, ob in zip(hl, ol):
            assert np.array_equal(hb.srcdata["ID"].cpu().numpy(), ob.srcdata["ID"])
            assert np.array_equal(hb.edata["ID"].cpu().numpy(), ob.edata["ID"])
    print("seed %3d ok  N %6d E %6d %s adaptive=%d min_block %d reverse=%d offload=%d chunks %d" % (
        seed, N, E, policy, adaptive, min_block, rev, offloaded, k), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=30)
    args = ap.parse_args()
    for seed in range(args.seeds):
        run(seed)
    print("all ok")


if __name__ == "__main__":
    main()
