"""Error behaviour at the drop-in boundary (SURVEY 8(b) "Errors").  The reference raises
ValueError / AssertionError in its Python layer for bad strings and shapes
(gnnflow/dynamic_graph.py:61-72,104-109; temporal_sampler.py:38-39; cache/cache.py:49-50) and
abort()s in native code; here native failures come back as status codes and become
ValueError / MemoryError, and — unlike an abort — must leave the object usable and unchanged."""
import numpy as np
import pytest


# ---- argument errors that never reach the GPU (run everywhere) ----------------------------
def test_bad_strings_raise_value_error_before_any_device_call():
    from gnnflow_amd import DynamicGraph
    with pytest.raises(ValueError):
        DynamicGraph(1, 2, "hbm3", 64, 1, "insert")
    with pytest.raises(ValueError):
        DynamicGraph(1, 2, "cuda", 64, 1, "append")


def test_cache_argument_errors():
    import torch
    from gnnflow_amd.cache import LRUCache
    nf = torch.zeros(10, 4)
    with pytest.raises(ValueError):                       # cache.py:49-50 "Cache must be on GPU"
        LRUCache(0.1, 0.1, 10, 20, "cpu", nf, None, 4, 0)
    with pytest.raises(AssertionError):
        LRUCache(1.5, 0.1, 10, 20, "cuda:0", nf, None, 4, 0)
    with pytest.raises(ValueError):                       # table does not match num_nodes
        LRUCache(0.1, 0.1, 11, 20, "cuda:0", nf, None, 4, 0)
    with pytest.raises(ValueError):
        LRUCache(0.1, 0.1, 10, 20, "cuda:0", nf, None, 4, 0, feature_placement="nvme")


# ---- errors that involve the device ----------------------------------------------------
@pytest.fixture()
def graph():
    from gnnflow_amd import DynamicGraph
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 4, 16, "insert")
    g.add_edges(np.array([0, 0, 1, 2]), np.array([1, 2, 2, 0]), np.array([1., 2., 3., 4.]))
    return g


@pytest.mark.gpu
def test_add_edges_shape_and_value_errors_leave_the_graph_intact(graph):
    from gnnflow_amd import TemporalSampler
    before = (graph.num_edges(), graph.num_vertices(), graph.out_degree(np.arange(3)).tolist())
    with pytest.raises(AssertionError):
        graph.add_edges(np.array([0, 1]), np.array([1]), np.array([5., 6.]))
    with pytest.raises(AssertionError):
        graph.add_edges(np.zeros((2, 2), np.int64), np.zeros((2, 2), np.int64), np.zeros((2, 2)))
    with pytest.raises(ValueError):                       # negative vertex id
        graph.add_edges(np.array([0, -1]), np.array([1, 2]), np.array([5., 6.]))
    with pytest.raises(ValueError):                       # empty batch (reference: CHECK_GT)
        graph.add_edges(np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.float32))
    # older than vertex 0's newest stored edge: rejected up front, nothing half-applied even
    # though the same batch also carries a valid edge for another vertex
    with pytest.raises(ValueError):
        graph.add_edges(np.array([3, 0]), np.array([1, 1]), np.array([9., 0.5]))
    assert (graph.num_edges(), graph.num_vertices(),
            graph.out_degree(np.arange(3)).tolist()) == before
    assert graph.max_vertex_id() == 2
    b = TemporalSampler(graph, [3]).sample(np.array([0, 3]), np.array([10., 10.]))[0][0]
    assert b.edata["ID"].cpu().tolist() == [1, 0]          # vertex 3 was never added
    graph.add_edges(np.array([0]), np.array([1]), np.array([2.0]))   # equal timestamp is fine
    assert graph.num_edges() == 5


@pytest.mark.gpu
@pytest.mark.parametrize("policy", ["insert", "replace"])
def test_rejected_batch_leaves_the_same_graph_intact(policy):
    """A batch that would exceed maximum_pool_size is rejected before anything is mutated:
    the SAME graph object keeps answering like the oracle that never saw the batch, and keeps
    ingesting (reference: rmm pool exhaustion -> abort, dynamic_graph.cu:164-166)."""
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from oracle import oracle as O
    from tests import synth
    N = 60
    src, dst, ts, eid = synth.powerlaw_graph(N, 3000, seed=5, tie_levels=200)
    g = DynamicGraph(1024, 100 * 1024, "cuda", 8, 1, policy)   # 100 KiB of logical blocks
    o = O.OracleGraph(minimum_block_size=8, insertion_policy=policy)
    for gr in (g, o):
        gr.add_edges(src[:1500], dst[:1500], ts[:1500], eid[:1500])
    mem = g.get_graph_memory_usage()

    def same_as_oracle():
        assert (g.num_edges(), g.num_vertices(), g.num_source_vertices()) == \
               (o.num_edges(), o.num_vertices(), o.num_source_vertices())
        assert g.max_vertex_id() == o.max_vertex_id()
        assert np.array_equal(g.out_degree(np.arange(N)), o.out_degree(np.arange(N)))
        for v in (0, 7, 31, N - 1):
            for a, b in zip(g.get_temporal_neighbors(v), o.get_temporal_neighbors(v)):
                assert np.array_equal(a, b)
        roots, rts = synth.random_roots(N, 200, 1000.0, seed=11)
        hm = TemporalSampler(g, [4, 4]).sample(roots, rts)
        om = O.OracleSampler(o, [4, 4]).sample(roots, rts)
        for hl, ol in zip(hm, om):
            assert np.array_equal(hl[0].edata["ID"].cpu().numpy(), ol[0].edata["ID"])
            assert np.array_equal(hl[0].srcdata["ID"].cpu().numpy(), ol[0].srcdata["ID"])

    same_as_oracle()
    # a batch far beyond the limit, touching old vertices and new ones (ids up to 4999)
    big = 20000
    rng = np.random.RandomState(1)
    with pytest.raises(MemoryError):
        g.add_edges(rng.randint(0, 5000, big), rng.randint(0, 5000, big),
                    np.sort(rng.uniform(2000, 3000, big)).astype(np.float32),
                    np.arange(10 ** 6, 10 ** 6 + big))
    assert g.get_graph_memory_usage() == mem
    same_as_oracle()
    # ... and the graph keeps ingesting what fits
    for gr in (g, o):
        gr.add_edges(src[1500:1800], dst[1500:1800], ts[1500:1800], eid[1500:1800])
    same_as_oracle()


@pytest.mark.gpu
def test_pool_limit_is_a_memory_error_not_an_abort():
    from gnnflow_amd import DynamicGraph
    g = DynamicGraph(1024, 4096, "cuda", 64, 1, "insert")       # 4 KiB of logical edge blocks
    n = 1000
    with pytest.raises(MemoryError):
        g.add_edges(np.arange(n), np.arange(n), np.arange(n, dtype=np.float32))
    # the object is still usable
    small = DynamicGraph(1024, 1 << 20, "cuda", 4, 1, "insert")
    small.add_edges(np.array([0]), np.array([1]), np.array([1.0]))
    assert small.num_edges() == 1


@pytest.mark.gpu
def test_sampler_argument_errors(graph):
    from gnnflow_amd import TemporalSampler
    with pytest.raises(ValueError):
        TemporalSampler(graph, [5], sample_strategy="oldest")
    with pytest.raises(ValueError):
        TemporalSampler(graph, [])
    with pytest.raises(ValueError):
        TemporalSampler(graph, [0, 5])
    s = TemporalSampler(graph, [2, 2], num_snapshots=1)
    with pytest.raises(AssertionError):
        s.sample(np.array([0, 1]), np.array([1.0]))
    with pytest.raises(ValueError):
        s.sample_layer(np.array([0]), np.array([1.0]), layer=2, snapshot=0)
    with pytest.raises(ValueError):
        s.sample_layer(np.array([0]), np.array([1.0]), layer=0, snapshot=1)
    # still works afterwards
    assert s.sample(np.array([0]), np.array([10.0]))[1][0].num_edges() == 2


@pytest.mark.gpu
def test_out_of_range_ids_fetch_zero_rows_and_never_fault():
    """An id outside the feature table (the reference would raise an IndexError from torch
    indexing) yields a zero row and no hit; the kernel never reads out of bounds."""
    import torch
    from gnnflow_amd.cache import LRUCache

    class Blk:
        def __init__(self, src, edge):
            self.srcdata, self.edata = {"ID": src}, {"ID": edge}
    nf = torch.arange(40, dtype=torch.float32).view(10, 4)
    c = LRUCache(0.0, 0.5, 10, 4, "cuda:0", nf, None, 4, 0)
    c.init_cache()
    ids = torch.tensor([3, 10, -1, 9, 1 << 40], device="cuda")
    b = Blk(ids, torch.zeros(0, dtype=torch.int64, device="cuda"))
    c.fetch_feature([[b]], None)
    got = b.srcdata["h"].cpu()
    assert torch.equal(got[0], nf[3]) and torch.equal(got[3], nf[9])
    assert torch.count_nonzero(got[[1, 2, 4]]) == 0
    assert float(c.cache_node_ratio) == pytest.approx(1 / 5)      # only id 3 was cached


@pytest.mark.gpu
def test_forked_child_does_not_free_the_parents_gpu_objects():
    """A process forked from one that holds handles (multiprocessing's fork start method: a
    Manager server, a DataLoader worker) inherits the Python objects; when its garbage
    collector finalises them, the destroy calls must not touch the parent's device memory
    (include/gnnflow_hip.h: handles belong to the process that loaded the library)."""
    import gc
    import os
    import numpy as np
    import gnnflow_amd
    from tests import synth
    src, dst, ts, eid = synth.powerlaw_graph(200, 4000, seed=3, tie_levels=100)
    g = gnnflow_amd.DynamicGraph(1 << 20, 1 << 28, "cuda", 8, 64, "insert")
    g.add_edges(src, dst, ts, eid)
    s = gnnflow_amd.TemporalSampler(g, [5, 5])
    nodes, t = synth.random_roots(200, 300, 1000.0, seed=1)
    before = [b.edata["ID"].cpu().numpy() for mfg in s.sample(nodes, t) for b in mfg]
    pid = os.fork()
    if pid == 0:            # the child: finalise its copies of the objects, then leave
        try:
            s.__del__()
            g.__del__()
            gc.collect()
        finally:
            os._exit(0)
    _, status = os.waitpid(pid, 0)
    assert status == 0
    after = [b.edata["ID"].cpu().numpy() for mfg in s.sample(nodes, t) for b in mfg]
    assert all(np.array_equal(a, b) for a, b in zip(before, after))
