"""gnnflow.DynamicGraph on MI355X — same constructor, methods, argument meaning and
error behaviour as the reference's Python wrapper (gnnflow/dynamic_graph.py:8-204),
over the C ABI (include/gnnflow_hip.h) instead of the `libgnnflow` pybind module."""
import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _capi


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


class DynamicGraph:
    """
    A dynamic graph that can be updated at runtime; edges live in HBM as one
    time-sorted segment per source vertex (see gnnflow_amd/csrc/edge_store.hpp).
    """

    def __init__(
            self, initial_pool_size: int,
            maximum_pool_size: int,
            mem_resource_type: str,
            minimum_block_size: int,
            blocks_to_preallocate: int,
            insertion_policy: str,
            source_vertices: Optional[np.ndarray] = None,
            target_vertices: Optional[np.ndarray] = None,
            timestamps: Optional[np.ndarray] = None,
            eids: Optional[np.ndarray] = None,
            add_reverse: bool = False,
            device: int = 0,
            adaptive_block_size: bool = True):
        # gnnflow/dynamic_graph.py:49-72: case-insensitive strings, ValueError otherwise
        mem_resource_type = mem_resource_type.lower()
        if mem_resource_type not in _capi.MEM_RESOURCE:
            raise ValueError("Invalid memory resource type: {}".format(mem_resource_type))
        insertion_policy = insertion_policy.lower()
        if insertion_policy not in _capi.INSERTION_POLICY:
            raise ValueError("Invalid insertion policy: {}".format(insertion_policy))

        self._lib = _capi.load()
        self._h = C.c_void_p()
        self._device = int(device)
        _capi.check(self._lib.gf_graph_create(
            C.byref(self._h), int(initial_pool_size), int(maximum_pool_size),
            _capi.MEM_RESOURCE[mem_resource_type], int(minimum_block_size),
            int(blocks_to_preallocate), _capi.INSERTION_POLICY[insertion_policy],
            self._device, 1 if adaptive_block_size else 0))

        if source_vertices is not None and target_vertices is not None \
                and timestamps is not None:
            self.add_edges(source_vertices, target_vertices, timestamps, eids, add_reverse)

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.gf_graph_destroy(h)
            self._h = None

    @property
    def device(self) -> int:
        return self._device

    def add_edges(
            self, source_vertices: np.ndarray, target_vertices: np.ndarray,
            timestamps: np.ndarray, eids: Optional[np.ndarray] = None,
            add_reverse: bool = False):
        """
        Add edges (any order inside the batch; grouped by source and stable-sorted by
        timestamp on ingest).  gnnflow/dynamic_graph.py:85-126.

        Raises:
            ValueError: if the timestamps are older than the existing edges.
        """
        source_vertices = np.asarray(source_vertices)
        target_vertices = np.asarray(target_vertices)
        timestamps = np.asarray(timestamps)
        assert len(source_vertices.shape) == 1 and len(
            target_vertices.shape) == 1 and len(timestamps.shape) == 1, "Edges must be 1D tensors"
        assert source_vertices.shape[0] == target_vertices.shape[0] == \
            timestamps.shape[0], "The number of source vertices, target vertices, timestamps, " \
            "and edge ids must be the same."

        if eids is None:
            num_edges = self.num_edges()
            eids = np.arange(num_edges, num_edges + len(source_vertices))
        eids = np.asarray(eids)

        if add_reverse:
            source_vertices, target_vertices = (
                np.concatenate([source_vertices, target_vertices]),
                np.concatenate([target_vertices, source_vertices]))
            timestamps = np.concatenate([timestamps, timestamps])
            eids = np.concatenate([eids, eids])

        src, dst, eid = _i64(source_vertices), _i64(target_vertices), _i64(eids)
        ts = np.ascontiguousarray(timestamps, dtype=np.float32)
        _capi.check(self._lib.gf_graph_add_edges(
            self._h, src.ctypes.data, dst.ctypes.data, ts.ctypes.data, eid.ctypes.data,
            len(src)))

    def offload_old_blocks(self, timestamp: float, to_file: bool = False):
        n = C.c_size_t(0)
        _capi.check(self._lib.gf_graph_offload_old_blocks(
            self._h, float(timestamp), 1 if to_file else 0, C.byref(n)))
        return n.value

    def _size(self, fn) -> int:
        n = C.c_size_t(0)
        _capi.check(fn(self._h, C.byref(n)))
        return n.value

    def num_vertices(self) -> int:
        return self._size(self._lib.gf_graph_num_vertices)

    def num_source_vertices(self) -> int:
        return self._size(self._lib.gf_graph_num_source_vertices)

    def num_edges(self) -> int:
        return self._size(self._lib.gf_graph_num_edges)

    def ids_fit_u32(self) -> bool:
        """Every node id and edge id inserted so far is in [0, 2^32 - 2] (the partitioned
        sampler's shared chains may then use 12-byte reply slots)."""
        v = C.c_int(0)
        _capi.check(self._lib.gf_graph_ids_fit_u32(self._h, C.byref(v)))
        return bool(v.value)

    def max_vertex_id(self) -> int:
        v = C.c_int64(0)
        _capi.check(self._lib.gf_graph_max_vertex_id(self._h, C.byref(v)))
        return v.value

    def out_degree(self, vertexs: np.ndarray) -> np.ndarray:
        v = _i64(vertexs)
        out = np.zeros(len(v), dtype=np.uint64)
        _capi.check(self._lib.gf_graph_out_degree(self._h, v.ctypes.data, len(v),
                                                  out.ctypes.data))
        return out

    def _id_list(self, fn) -> np.ndarray:
        n = C.c_size_t(0)
        _capi.check(fn(self._h, None, 0, C.byref(n)))
        out = np.zeros(n.value, dtype=np.int64)
        _capi.check(fn(self._h, out.ctypes.data, len(out), C.byref(n)))
        return out

    def nodes(self) -> np.ndarray:
        return self._id_list(self._lib.gf_graph_nodes)

    def src_nodes(self) -> np.ndarray:
        return self._id_list(self._lib.gf_graph_src_nodes)

    def edges(self) -> np.ndarray:
        return self._id_list(self._lib.gf_graph_edges)

    def get_temporal_neighbors(self, vertex: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """(target_vertices, timestamps, edge_ids), newest first."""
        n = C.c_size_t(0)
        _capi.check(self._lib.gf_graph_get_temporal_neighbors(
            self._h, int(vertex), None, None, None, 0, C.byref(n)))
        d = np.zeros(n.value, np.int64)
        t = np.zeros(n.value, np.float32)
        e = np.zeros(n.value, np.int64)
        if n.value:
            _capi.check(self._lib.gf_graph_get_temporal_neighbors(
                self._h, int(vertex), d.ctypes.data, t.ctypes.data, e.ctypes.data,
                n.value, C.byref(n)))
        return d, t, e

    def _float(self, fn) -> float:
        v = C.c_float(0)
        _capi.check(fn(self._h, C.byref(v)))
        return v.value

    def avg_linked_list_length(self) -> float:
        return self._float(self._lib.gf_graph_avg_linked_list_length)

    def get_graph_memory_usage(self) -> int:
        return self._float(self._lib.gf_graph_memory_usage)

    def get_metadata_memory_usage(self) -> int:
        return self._float(self._lib.gf_graph_metadata_memory_usage)
