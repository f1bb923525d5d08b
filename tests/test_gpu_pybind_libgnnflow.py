"""GPU parity, part 4: the pybind11 module named `libgnnflow` (the reference's native
module surface, gnnflow/csrc/api.cc) driven exactly the way the reference's Python wrappers
drive it (gnnflow/dynamic_graph.py:74-126, gnnflow/temporal_sampler.py:48-80,149-165), on
the reference's golden cases."""
import os
import sys

import numpy as np
import pytest

from tests.golden_runner import load_cases, run_case

pytestmark = pytest.mark.gpu
CASES = load_cases()


def _lib():
    from gnnflow_amd import _build
    path = _build.build_pybind()
    d = os.path.dirname(path)
    if d not in sys.path:
        sys.path.insert(0, d)
    import libgnnflow
    return libgnnflow


class RefStyleGraph:
    """What gnnflow/dynamic_graph.py does around libgnnflow._DynamicGraph."""

    def __init__(self, initial_pool_size=1 << 20, maximum_pool_size=1 << 26,
                 mem_resource_type="cuda", minimum_block_size=64, blocks_to_preallocate=128,
                 insertion_policy="insert", device=0, adaptive_block_size=True):
        L = _lib()
        mem = getattr(L.MemoryResourceType, mem_resource_type.upper())
        pol = getattr(L.InsertionPolicy, insertion_policy.upper())
        self._dgraph = L._DynamicGraph(initial_pool_size, maximum_pool_size, mem,
                                       minimum_block_size, blocks_to_preallocate, pol, device,
                                       adaptive_block_size)

    def add_edges(self, src, dst, ts, eids=None, add_reverse=False):
        if eids is None:
            n0 = self.num_edges()
            eids = np.arange(n0, n0 + len(src))
        if add_reverse:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
            ts = np.concatenate([ts, ts])
            eids = np.concatenate([eids, eids])
        self._dgraph.add_edges(src, dst, ts, eids)

    def __getattr__(self, name):
        return getattr(self._dgraph, name)


class NpBlock:
    def __init__(self, r):
        self.r = r
        self.srcdata = {"ID": r.all_nodes(), "ts": r.all_timestamps()}
        self.edata = {"dt": r.delta_timestamps(), "ID": r.eids()}

    def num_src_nodes(self): return self.r.num_src_nodes()
    def num_dst_nodes(self): return self.r.num_dst_nodes()
    def edges(self): return self.r.col(), self.r.row()


class RefStyleSampler:
    """What gnnflow/temporal_sampler.py does around libgnnflow._TemporalSampler."""

    def __init__(self, graph, fanouts, sample_strategy="recent", num_snapshots=1,
                 snapshot_time_window=0.0, prop_time=False, seed=1234):
        L = _lib()
        pol = L.SamplingPolicy.RECENT if sample_strategy == "recent" else L.SamplingPolicy.UNIFORM
        self._sampler = L._TemporalSampler(graph._dgraph, fanouts, pol, num_snapshots,
                                           snapshot_time_window, prop_time, seed)

    def sample(self, nodes, ts):
        res = self._sampler.sample(nodes, ts)
        mfgs = [[NpBlock(r) for r in layer] for layer in res]
        mfgs.reverse()
        return mfgs

    def sample_layer(self, nodes, ts, layer, snapshot):
        return NpBlock(self._sampler.sample_layer(nodes, ts, layer, snapshot))


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_pybind_module_matches_reference_golden(case):
    run_case(case, RefStyleGraph, RefStyleSampler)


def test_numpy_types_and_errors():
    L = _lib()
    g = RefStyleGraph()
    g.add_edges(np.array([0, 0, 1]), np.array([1, 2, 2]), np.array([1, 2, 3]))
    d, t, e = g.get_temporal_neighbors(0)
    assert d.dtype == np.int64 and t.dtype == np.float32 and e.dtype == np.int64
    assert g.out_degree([0, 1, 2]).tolist() == [2, 1, 0]
    with pytest.raises(ValueError):
        g.add_edges(np.array([0]), np.array([1]), np.array([0.5]))
    s = RefStyleSampler(g, [2])
    r = s._sampler.sample_layer(np.array([0, 1]), np.array([5, 5]), 0, 0)   # int ts, as upstream tests
    assert r.row().dtype == np.int64 and r.delta_timestamps().dtype == np.float32
    assert r.num_dst_nodes() == 2 and r.num_src_nodes() == 5
