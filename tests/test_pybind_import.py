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
