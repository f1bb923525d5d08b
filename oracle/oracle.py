"""
oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

ctypes front-end of oracle/gnnflow_oracle.c (CPU restatement of the reference's
edge store + sampler) plus the Python-level shaping the reference does around
its native module:

  * OracleGraph.add_edges     — gnnflow/dynamic_graph.py:85-126 (default eids,
                                add_reverse concat) over DynamicGraph::AddEdges
  * OracleSampler.sample      — TemporalSampler::Sample layer chaining
                                (gnnflow/csrc/temporal_sampler.cu:279-305) and the
                                [layer][snapshot] -> reversed MFG list of
                                gnnflow/temporal_sampler.py:149-165

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  gnnflow_amd never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libgnnflow_oracle.so")

_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "gnnflow_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    L.gfo_graph_create.restype = C.c_void_p
    L.gfo_graph_create.argtypes = [C.c_size_t, C.c_int, C.c_int]
    L.gfo_graph_destroy.argtypes = [C.c_void_p]
    L.gfo_graph_add_edges.restype = C.c_int
    L.gfo_graph_add_edges.argtypes = [C.c_void_p, _i64p, _i64p, _f32p, _i64p, C.c_size_t]
    L.gfo_graph_offload_old_blocks.restype = C.c_size_t
    L.gfo_graph_offload_old_blocks.argtypes = [C.c_void_p, C.c_float]
    for name in ("num_nodes", "num_src_nodes", "num_edges"):
        f = getattr(L, "gfo_graph_" + name)
        f.restype = C.c_size_t
        f.argtypes = [C.c_void_p]
    L.gfo_graph_max_node_id.restype = C.c_int64
    L.gfo_graph_max_node_id.argtypes = [C.c_void_p]
    L.gfo_graph_out_degree.restype = C.c_int
    L.gfo_graph_out_degree.argtypes = [C.c_void_p, _i64p, C.c_size_t, C.c_void_p]
    L.gfo_graph_nodes.restype = C.c_size_t
    L.gfo_graph_nodes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    L.gfo_graph_edges.restype = C.c_size_t
    L.gfo_graph_edges.argtypes = [C.c_void_p, C.c_void_p]
    L.gfo_graph_get_temporal_neighbors.restype = C.c_size_t
    L.gfo_graph_get_temporal_neighbors.argtypes = [
        C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
    for name in ("avg_linked_list_length", "mem_usage", "metadata_mem_usage"):
        f = getattr(L, "gfo_graph_" + name)
        f.restype = C.c_float
        f.argtypes = [C.c_void_p]
    L.gfo_graph_num_blocks.restype = C.c_size_t
    L.gfo_graph_num_blocks.argtypes = [C.c_void_p, C.c_int64]
    L.gfo_graph_block_info.restype = C.c_int
    L.gfo_graph_block_info.argtypes = [
        C.c_void_p, C.c_int64, C.c_size_t, C.POINTER(C.c_size_t),
        C.POINTER(C.c_size_t), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.gfo_sample_layer.restype = C.c_int
    L.gfo_sample_layer.argtypes = [
        C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int,
        C.c_uint64, C.c_uint64, _i64p, _f32p, C.c_size_t,
        _i64p, _f32p, _f32p, _i64p, _i64p, _i64p, C.POINTER(C.c_size_t)]
    L.gfo_sample_layer_mt.restype = C.c_int
    L.gfo_sample_layer_mt.argtypes = L.gfo_sample_layer.argtypes + [C.c_int]
    L.gfo_gather_rows.restype = C.c_int
    L.gfo_gather_rows.argtypes = [_f32p, C.c_size_t, C.c_size_t, _i64p, C.c_size_t, _f32p]
    L.gfo_gather_rows_mt.restype = C.c_int
    L.gfo_gather_rows_mt.argtypes = L.gfo_gather_rows.argtypes + [C.c_int]
    L.gfo_philox_first.restype = C.c_uint32
    L.gfo_philox_first.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
    _lib = L
    return L


class OracleError(ValueError):
    pass


class OracleGraph:
    """CPU restatement of gnnflow.DynamicGraph (block-list edge store)."""

    def __init__(self, minimum_block_size=64, insertion_policy="insert",
                 adaptive_block_size=True, **_ignored):
        pol = insertion_policy.lower()
        if pol not in ("insert", "replace"):
            raise ValueError("Invalid insertion policy: {}".format(insertion_policy))
        self._h = lib().gfo_graph_create(int(minimum_block_size),
                                         0 if pol == "insert" else 1,
                                         1 if adaptive_block_size else 0)

    def __del__(self):
        if getattr(self, "_h", None) and _lib is not None:   # not during interpreter exit
            _lib.gfo_graph_destroy(self._h)
            self._h = None

    # gnnflow/dynamic_graph.py:85-126
    def add_edges(self, source_vertices, target_vertices, timestamps, eids=None,
                  add_reverse=False):
        src = np.asarray(source_vertices)
        dst = np.asarray(target_vertices)
        ts = np.asarray(timestamps)
        assert src.ndim == 1 and dst.ndim == 1 and ts.ndim == 1
        assert src.shape[0] == dst.shape[0] == ts.shape[0]
        if eids is None:
            n0 = self.num_edges()
            eids = np.arange(n0, n0 + len(src))
        eids = np.asarray(eids)
        if add_reverse:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
            ts = np.concatenate([ts, ts])
            eids = np.concatenate([eids, eids])
        rc = lib().gfo_graph_add_edges(
            self._h, np.ascontiguousarray(src, np.int64),
            np.ascontiguousarray(dst, np.int64),
            np.ascontiguousarray(ts, np.float32),
            np.ascontiguousarray(eids, np.int64), len(src))
        if rc != 0:
            raise OracleError("add_edges failed with code {}".format(rc))

    def offload_old_blocks(self, timestamp, to_file=False):
        return int(lib().gfo_graph_offload_old_blocks(self._h, float(timestamp)))

    def num_vertices(self):
        return int(lib().gfo_graph_num_nodes(self._h))

    def num_source_vertices(self):
        return int(lib().gfo_graph_num_src_nodes(self._h))

    def num_edges(self):
        return int(lib().gfo_graph_num_edges(self._h))

    def max_vertex_id(self):
        return int(lib().gfo_graph_max_node_id(self._h))

    def out_degree(self, vertices):
        v = np.ascontiguousarray(vertices, np.int64)
        out = np.zeros(len(v), dtype=np.uint64)
        rc = lib().gfo_graph_out_degree(self._h, v, len(v), out.ctypes.data)
        if rc != 0:
            raise OracleError("out_degree: node out of range")
        return out

    def nodes(self):
        n = lib().gfo_graph_nodes(self._h, None, 0)
        out = np.zeros(n, np.int64)
        lib().gfo_graph_nodes(self._h, out.ctypes.data, 0)
        return out

    def src_nodes(self):
        n = lib().gfo_graph_nodes(self._h, None, 1)
        out = np.zeros(n, np.int64)
        lib().gfo_graph_nodes(self._h, out.ctypes.data, 1)
        return out

    def edges(self):
        n = lib().gfo_graph_edges(self._h, None)
        out = np.zeros(n, np.int64)
        lib().gfo_graph_edges(self._h, out.ctypes.data)
        return out

    def get_temporal_neighbors(self, vertex):
        n = lib().gfo_graph_get_temporal_neighbors(self._h, int(vertex), None, None, None)
        d = np.zeros(n, np.int64)
        t = np.zeros(n, np.float32)
        e = np.zeros(n, np.int64)
        lib().gfo_graph_get_temporal_neighbors(
            self._h, int(vertex), d.ctypes.data, t.ctypes.data, e.ctypes.data)
        return d, t, e

    def avg_linked_list_length(self):
        return float(lib().gfo_graph_avg_linked_list_length(self._h))

    def get_graph_memory_usage(self):
        return float(lib().gfo_graph_mem_usage(self._h))

    def get_metadata_memory_usage(self):
        return float(lib().gfo_graph_metadata_mem_usage(self._h))

    def blocks(self, node):
        """[(size, capacity, start_ts, end_ts)] oldest first."""
        out = []
        for i in range(int(lib().gfo_graph_num_blocks(self._h, int(node)))):
            s, c = C.c_size_t(), C.c_size_t()
            a, b = C.c_float(), C.c_float()
            lib().gfo_graph_block_info(self._h, int(node), i, C.byref(s), C.byref(c),
                                       C.byref(a), C.byref(b))
            out.append((s.value, c.value, a.value, b.value))
        return out


class OracleResult:
    """Stand-in for libgnnflow.SamplingResult (gnnflow/csrc/common.h:51-60)."""

    def __init__(self, row, col, all_nodes, all_timestamps, delta_timestamps, eids,
                 num_src_nodes, num_dst_nodes):
        self._row, self._col = row, col
        self._all_nodes, self._all_ts = all_nodes, all_timestamps
        self._dt, self._eids = delta_timestamps, eids
        self._ns, self._nd = num_src_nodes, num_dst_nodes

    def row(self): return self._row
    def col(self): return self._col
    def all_nodes(self): return self._all_nodes
    def all_timestamps(self): return self._all_ts
    def delta_timestamps(self): return self._dt
    def eids(self): return self._eids
    def num_src_nodes(self): return self._ns
    def num_dst_nodes(self): return self._nd


class OracleBlock:
    """Minimal DGLBlock look-alike: the fields gnnflow/temporal_sampler.py:149-165
    fills, as numpy arrays."""

    def __init__(self, r: OracleResult):
        self._r = r
        self.srcdata = {"ID": r.all_nodes(), "ts": r.all_timestamps()}
        self.edata = {"dt": r.delta_timestamps(), "ID": r.eids()}

    def num_src_nodes(self): return self._r.num_src_nodes()
    def num_dst_nodes(self): return self._r.num_dst_nodes()
    def num_edges(self): return len(self._r.eids())
    def edges(self): return self._r.col(), self._r.row()


class OracleSampler:
    """CPU restatement of gnnflow.TemporalSampler."""

    def __init__(self, graph: OracleGraph, fanouts, sample_strategy="recent",
                 num_snapshots=1, snapshot_time_window=0.0, prop_time=False,
                 seed=1234, *args, **kwargs):
        s = sample_strategy.lower()
        if s not in ("recent", "uniform"):
            raise ValueError("strategy must be 'recent' or 'uniform'")
        self._g = graph
        self._fanouts = [int(f) for f in fanouts]
        self._policy = 0 if s == "recent" else 1
        self._num_snapshots = int(num_snapshots)
        self._window = float(snapshot_time_window)
        self._prop_time = bool(prop_time)
        self._seed = int(seed)
        self._is_static = bool(kwargs.get("is_static", False))
        self._calls = 0  # RNG call counter (include/gnnflow_rng.h)
        # > 1: the OpenMP form of the same routine (bench.py's cpu_baseline); identical output
        self.threads = int(kwargs.get("threads", 1))

    def sample_layer_raw(self, nodes, ts, layer, snapshot) -> OracleResult:
        nodes = np.ascontiguousarray(nodes, np.int64)
        ts = np.ascontiguousarray(ts, np.float32)
        R = len(nodes)
        F = self._fanouts[layer]
        all_nodes = np.empty(R + R * F, np.int64)
        all_ts = np.empty(R + R * F, np.float32)
        dt = np.empty(R * F, np.float32)
        eids = np.empty(R * F, np.int64)
        row = np.empty(R * F, np.int64)
        col = np.empty(R * F, np.int64)
        S = C.c_size_t(0)
        call = self._calls
        self._calls += 1
        args = (self._g._h, self._policy, F, self._num_snapshots, snapshot,
                self._window, int(self._prop_time), self._seed, call, nodes, ts, R,
                all_nodes, all_ts, dt, eids, row, col, C.byref(S))
        if self.threads > 1:
            rc = lib().gfo_sample_layer_mt(*args, self.threads)
        else:
            rc = lib().gfo_sample_layer(*args)
        if rc != 0:
            raise OracleError("sample_layer failed with code {}".format(rc))
        S = S.value
        return OracleResult(row[:S].copy(), col[:S].copy(), all_nodes[:R + S].copy(),
                            all_ts[:R + S].copy(), dt[:S].copy(), eids[:S].copy(),
                            R + S, R)

    # TemporalSampler::Sample, temporal_sampler.cu:279-305
    def sample_raw(self, nodes, ts):
        results = []
        for layer in range(len(self._fanouts)):
            layer_results = []
            for snap in range(self._num_snapshots):
                if layer == 0:
                    n, t = nodes, ts
                else:
                    prev = results[-1][snap]
                    n, t = prev.all_nodes(), prev.all_timestamps()
                layer_results.append(self.sample_layer_raw(n, t, layer, snap))
            results.append(layer_results)
        return results

    # gnnflow/temporal_sampler.py:60-80,149-165
    def sample(self, target_vertices, timestamps):
        target_vertices = np.asarray(target_vertices)
        if self._is_static:
            timestamps = np.full(target_vertices.shape, np.finfo(np.float32).max)
        res = self.sample_raw(target_vertices, timestamps)
        mfgs = [[OracleBlock(r) for r in layer] for layer in res]
        mfgs.reverse()
        return mfgs

    def sample_layer(self, target_vertices, timestamps, layer, snapshot):
        return OracleBlock(self.sample_layer_raw(target_vertices, timestamps,
                                                 layer, snapshot))


def gather_rows(feats: np.ndarray, ids: np.ndarray, threads: int = 1) -> np.ndarray:
    """gnnflow/utils.py:465-474 prepare_input: feats[ids].float()."""
    feats = np.ascontiguousarray(feats, np.float32)
    ids = np.ascontiguousarray(ids, np.int64)
    out = np.empty((len(ids), feats.shape[1]), np.float32)
    if threads > 1:
        rc = lib().gfo_gather_rows_mt(feats, feats.shape[0], feats.shape[1], ids, len(ids), out,
                                      int(threads))
    else:
        rc = lib().gfo_gather_rows(feats, feats.shape[0], feats.shape[1], ids, len(ids), out)
    if rc != 0:
        raise OracleError("gather_rows: id out of range")
    return out


def philox_first(seed, tid, call):
    return int(lib().gfo_philox_first(seed, tid, call))
