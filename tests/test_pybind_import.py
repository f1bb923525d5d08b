"""CPU test: the pybind11 `libgnnflow` module builds, imports, exposes the reference's
native-module surface (gnnflow/csrc/api.cc), and — in the build container, where the
reference is mounted — the reference's own Python wrappers import against it unmodified."""
import os
import sys
import types

import pytest


@pytest.fixture(scope="module")
def lib():
    from gnnflow_amd import _build
    path = _build.build_pybind()
    d = os.path.dirname(path)
    if d not in sys.path:
        sys.path.insert(0, d)
    import libgnnflow
    return libgnnflow


def test_surface_matches_api_cc(lib):
    assert [m for m in ("INSERT", "REPLACE") if hasattr(lib.InsertionPolicy, m)] == ["INSERT", "REPLACE"]
    assert hasattr(lib.SamplingPolicy, "RECENT") and hasattr(lib.SamplingPolicy, "UNIFORM")
    for m in ("CUDA", "UNIFIED", "PINNED", "SHARED"):
        assert hasattr(lib.MemoryResourceType, m)
    for m in ("add_edges", "offload_old_blocks", "num_vertices", "num_source_vertices",
              "num_edges", "out_degree", "nodes", "src_nodes", "edges", "max_vertex_id",
              "get_temporal_neighbors", "avg_linked_list_length", "get_graph_memory_usage",
              "get_metadata_memory_usage"):
        assert hasattr(lib._DynamicGraph, m), m
    for m in ("row", "col", "all_nodes", "all_timestamps", "delta_timestamps", "eids",
              "num_src_nodes", "num_dst_nodes"):
        assert hasattr(lib.SamplingResult, m), m
    assert hasattr(lib._TemporalSampler, "sample") and hasattr(lib._TemporalSampler, "sample_layer")
    for m in ("set", "get", "memory_usage", "fill_zeros"):     # api.cc:122-127
        assert hasattr(lib.KVStore, m), m


def test_kvstore_is_the_reference_host_map(lib):
    """kvstore.cc: set keeps values[i] per key (a view, so fill_zeros reaches the caller's
    tensor), get returns rows in key order, memory_usage counts map entries only."""
    import torch
    kv = lib.KVStore()
    vals = torch.arange(20, dtype=torch.float32).view(5, 4)
    kv.set([7, 3, 4000000000, 0, 9], vals)
    got = torch.stack(kv.get([9, 7, 4000000000]))
    assert torch.equal(got, vals[[4, 0, 2]])
    kv.set([7], torch.full((1, 4), -1.0))                    # overwrite one key
    assert torch.equal(kv.get([7])[0], torch.full((4,), -1.0))
    assert kv.memory_usage() == 5 * 12                        # (sizeof(Key) + sizeof(at::Tensor))
    kv.fill_zeros()
    assert float(vals[1:].abs().sum()) == 0.0 and float(torch.stack(kv.get([7, 3])).abs().sum()) == 0.0
    assert kv.get([12345]) == [None]                          # operator[] default-constructs
    assert kv.memory_usage() == 6 * 12
    with pytest.raises(TypeError):
        kv.set([-1], vals[:1])                                # Key is unsigned int


@pytest.mark.skipif(not os.path.isdir("/root/reference/gnnflow"),
                    reason="reference tree only exists in the build container")
def test_reference_wrappers_import_against_our_module(lib):
    """gnnflow/dynamic_graph.py does `from libgnnflow import InsertionPolicy,
    MemoryResourceType, _DynamicGraph`; load that file as-is against our module."""
    import importlib.util
    sys.dont_write_bytecode = True
    pkg = types.ModuleType("gnnflow_ref_probe")
    pkg.__path__ = []
    sys.modules["gnnflow_ref_probe"] = pkg
    spec = importlib.util.spec_from_file_location(
        "gnnflow_ref_probe.dynamic_graph", "/root/reference/gnnflow/dynamic_graph.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod._DynamicGraph is lib._DynamicGraph
    # bad strings are rejected by the reference wrapper before any native call
    with pytest.raises(ValueError):
        mod.DynamicGraph(1, 2, "bogus", 4, 8, "insert")


_CHAIN = r"""
import sys
sys.dont_write_bytecode = True
sys.path[:0] = [{csrc!r}, {repo!r}, "/root/reference"]
import gnnflow_amd.dgl_compat as compat
compat.install(force=True)
import libgnnflow
import gnnflow.cache as caches                       # scripts/offline_edge_prediction.py:20
import gnnflow.cache.cache, gnnflow.distributed.kvstore as kvs
from gnnflow.temporal_sampler import TemporalSampler  # :28
from gnnflow.utils import build_dynamic_graph, mfgs_to_cuda   # :29-32
from gnnflow.config import get_default_config
assert kvs.KVStore is libgnnflow.KVStore
assert gnnflow.cache.cache.KVStoreClient is kvs.KVStoreClient
names = sorted(n for n in caches.__dict__ if not n.startswith("__") and callable(caches.__dict__[n]))
assert names == ["FIFOCache", "GNNLabStaticCache", "LFUCache", "LRUCache"], names   # :36-38
import gnnflow
assert gnnflow.DynamicGraph.__module__ == "gnnflow.dynamic_graph"
assert TemporalSampler.__module__ == "gnnflow.temporal_sampler"
import inspect
assert inspect.getsourcefile(caches.LRUCache).startswith("/root/reference/")
# the reference's own server class on top of our KVStore (USE_CPP_KVSTORE=1, kvstore.py:28-41)
import os, torch
os.environ["USE_CPP_KVSTORE"] = "1"
srv = kvs.KVStoreServer.__new__(kvs.KVStoreServer)
srv._use_cpp_kvstore = True
srv._node_feat_kvstore, srv._edge_feat_kvstore, srv._memory_kvstore = (libgnnflow.KVStore() for _ in range(3))
vals = torch.arange(12.).view(4, 3)
srv._node_feat_kvstore.set([1, 2, 3, 4], vals)
assert torch.equal(torch.stack(srv._node_feat_kvstore.get([4, 1])), vals[[3, 0]])   # kvstore.py:192
print("chain ok")
"""


@pytest.mark.skipif(not os.path.isdir("/root/reference/gnnflow"),
                    reason="reference tree only exists in the build container")
def test_unmodified_import_chain_of_the_training_script(lib):
    """scripts/offline_edge_prediction.py:20 `import gnnflow.cache as caches` ->
    gnnflow/cache/cache.py:7 -> gnnflow/distributed/kvstore.py:12 `from libgnnflow import
    KVStore`, plus gnnflow/temporal_sampler.py and gnnflow/utils.py: the reference's WHOLE
    package imports against our module (and `dgl_compat` for the dgl names), in a fresh
    interpreter so nothing of it leaks into this one.  (The script's other imports —
    GPUtil, and gnnflow/data.py's `torch._six` — fail on this image for reasons that have
    nothing to do with libgnnflow)"""
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _CHAIN.format(csrc=os.path.dirname(lib.__file__), repo=repo)
    r = subprocess.run([sys.executable, "-B", "-c", code], capture_output=True, text=True,
                       cwd="/tmp", timeout=300)
    assert r.returncode == 0 and "chain ok" in r.stdout, r.stdout + r.stderr
