"""ctypes binding of include/gnnflow_hip.h (the C-ABI drop-in boundary).

The library is gnnflow_amd/csrc/libgnnflow_hip.so, built in-tree by
gnnflow_amd/_build.py.  There is NO fallback: if the HIP library is missing or
fails to load, importing this module raises — the product path never runs on a
CPU restatement.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libgnnflow_hip.so")

GF_OK = 0
GF_ERR_INVALID_ARGUMENT = 1
GF_ERR_TIMESTAMP_ORDER = 2
GF_ERR_OUT_OF_MEMORY = 3
GF_ERR_HIP = 4
GF_ERR_IO = 5

INSERTION_POLICY = {"insert": 0, "replace": 1}
SAMPLING_POLICY = {"recent": 0, "uniform": 1}
MEM_RESOURCE = {"cuda": 0, "unified": 1, "pinned": 2, "shared": 3}

CACHE_POLICY = {"lru": 0, "lfu": 1, "fifo": 2}
PROFILE_SLOTS = {"search": 0, "emit": 1, "gather": 2, "scan": 3, "lru": 4}


class GfBlock(C.Structure):
    """struct gf_block (include/gnnflow_hip.h)."""
    _fields_ = [
        ("all_nodes", C.c_void_p),
        ("all_timestamps", C.c_void_p),
        ("delta_timestamps", C.c_void_p),
        ("eids", C.c_void_p),
        ("row", C.c_void_p),
        ("col", C.c_void_p),
        ("num_dst_nodes", C.c_uint64),
        ("num_src_nodes", C.c_uint64),
        ("num_edges", C.c_uint64),
    ]


class GfFetchDesc(C.Structure):
    """struct gf_fetch_desc (include/gnnflow_hip.h)."""
    _fields_ = [
        ("kind", C.c_int),
        ("update", C.c_int),
        ("d_ids", C.c_void_p),
        ("n", C.c_size_t),
        ("d_out", C.c_void_p),
        ("d_stats", C.c_void_p),
    ]


class GfPullDesc(C.Structure):
    """struct gf_pull_desc (include/gnnflow_hip.h)."""
    _fields_ = [
        ("cache", C.c_void_p),
        ("d_ids", C.c_void_p),
        ("n", C.c_size_t),
        ("d_key_base", C.c_void_p),
        ("d_key_index", C.c_void_p),
        ("num_ids", C.c_size_t),
        ("d_send_ids", C.c_void_p),
        ("d_req_pos", C.c_void_p),
    ]


class GfPullCtx(C.Structure):
    """struct gf_pull_ctx (include/gnnflow_hip.h)."""
    _fields_ = [
        ("pull", GfPullDesc),
        ("d_shard_rows", C.c_void_p),
        ("shard_rows", C.c_size_t),
        ("d_shard_index", C.c_void_p),
        ("dim", C.c_size_t),
        ("kind", C.c_int),
        ("update", C.c_int),
        ("d_out", C.c_void_p),
        ("d_stats", C.c_void_p),
    ]


class GfFetchPulledDesc(C.Structure):
    """struct gf_fetch_pulled_desc (include/gnnflow_hip.h)."""
    _fields_ = [
        ("kind", C.c_int),
        ("update", C.c_int),
        ("d_ids", C.c_void_p),
        ("n", C.c_size_t),
        ("d_out", C.c_void_p),
        ("d_stats", C.c_void_p),
        ("d_pulled_rows", C.c_void_p),
        ("d_req_pos", C.c_void_p),
    ]


class GfPartLayout(C.Structure):
    """struct gf_part_layout (include/gnnflow_hip.h): byte offsets inside the workspace of
    one (layer, snapshot) of a chained partitioned sample."""
    _fields_ = [(n, C.c_size_t) for n in ("root_bound", "requests", "replies", "counts", "pos",
                                          "scratch", "scratch_bytes", "total", "slot_stride",
                                          "inbox", "served")]


class GfGroupSample(C.Structure):
    """struct gf_group_sample (include/gnnflow_hip.h): one sample of a shared chain."""
    _fields_ = [("sampler", C.c_void_p), ("d_roots", C.c_void_p), ("d_root_ts", C.c_void_p),
                ("num_roots", C.c_size_t), ("d_out", C.c_void_p), ("out_bytes", C.c_size_t)]


GF_PART_GROUP_MAX = 4


# every symbol include/gnnflow_hip.h declares: name -> (restype, argtypes)
_p = C.c_void_p
_sz = C.c_size_t
PROTOTYPES = {
    "gf_last_error": (C.c_char_p, []),
    "gf_version": (C.c_char_p, []),
    "gf_graph_create": (C.c_int, [C.POINTER(_p), _sz, _sz, C.c_int, _sz, _sz, C.c_int,
                                  C.c_int, C.c_int]),
    "gf_graph_destroy": (C.c_int, [_p]),
    "gf_graph_add_edges": (C.c_int, [_p, _p, _p, _p, _p, _sz]),
    "gf_graph_offload_old_blocks": (C.c_int, [_p, C.c_float, C.c_int, C.POINTER(_sz)]),
    "gf_graph_num_vertices": (C.c_int, [_p, C.POINTER(_sz)]),
    "gf_graph_num_source_vertices": (C.c_int, [_p, C.POINTER(_sz)]),
    "gf_graph_num_edges": (C.c_int, [_p, C.POINTER(_sz)]),
    "gf_graph_ids_fit_u32": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "gf_graph_max_vertex_id": (C.c_int, [_p, C.POINTER(C.c_int64)]),
    "gf_graph_out_degree": (C.c_int, [_p, _p, _sz, _p]),
    "gf_graph_nodes": (C.c_int, [_p, _p, _sz, C.POINTER(_sz)]),
    "gf_graph_src_nodes": (C.c_int, [_p, _p, _sz, C.POINTER(_sz)]),
    "gf_graph_edges": (C.c_int, [_p, _p, _sz, C.POINTER(_sz)]),
    "gf_graph_get_temporal_neighbors": (C.c_int, [_p, C.c_int64, _p, _p, _p, _sz,
                                                  C.POINTER(_sz)]),
    "gf_graph_avg_linked_list_length": (C.c_int, [_p, C.POINTER(C.c_float)]),
    "gf_graph_memory_usage": (C.c_int, [_p, C.POINTER(C.c_float)]),
    "gf_graph_metadata_memory_usage": (C.c_int, [_p, C.POINTER(C.c_float)]),
    "gf_graph_device": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "gf_sampler_create": (C.c_int, [C.POINTER(_p), _p, C.POINTER(C.c_uint32), _sz, C.c_int,
                                    C.c_uint32, C.c_float, C.c_int, C.c_uint64]),
    "gf_sampler_destroy": (C.c_int, [_p]),
    "gf_sampler_output_bytes": (C.c_int, [_p, _sz, C.POINTER(_sz)]),
    "gf_sampler_sample": (C.c_int, [_p, _p, _p, _sz, _p, _sz, C.POINTER(GfBlock), _p]),
    "gf_sampler_sample_begin": (C.c_int, [_p, _p, _p, _sz, _p, _sz, _p]),
    "gf_sampler_sample_begin_async": (C.c_int, [_p, _p, _p, _sz, _p, _sz, _p]),
    "gf_sampler_sample_end": (C.c_int, [_p, C.POINTER(GfBlock)]),
    "gf_sampler_set_enqueue_lane": (C.c_int, [_p, C.c_int]),
    "gf_sampler_call_counter": (C.c_int, [_p, C.POINTER(C.c_uint64)]),
    "gf_sampler_set_call_counter": (C.c_int, [_p, C.c_uint64, C.c_int]),
    "gf_sampler_layer_output_bytes": (C.c_int, [_p, _sz, C.c_uint32, C.POINTER(_sz)]),
    "gf_sampler_sample_layer": (C.c_int, [_p, _p, _p, _sz, C.c_uint32, C.c_uint32, _p, _sz,
                                          C.POINTER(GfBlock), _p]),
    "gf_sampler_sample_host": (C.c_int, [_p, _p, _p, _sz, C.POINTER(GfBlock)]),
    "gf_sampler_sample_layer_host": (C.c_int, [_p, _p, _p, _sz, C.c_uint32, C.c_uint32,
                                               C.POINTER(GfBlock)]),
    "gf_host_blocks_free": (None, [C.POINTER(GfBlock), _sz]),
    "gf_cache_create": (C.c_int, [C.POINTER(_p), _sz, _sz, _sz, _p, C.c_int]),
    "gf_cache_destroy": (C.c_int, [_p]),
    "gf_cache_set_policy": (C.c_int, [_p, C.c_int]),
    "gf_cache_reset_order": (C.c_int, [_p, _p]),
    "gf_cache_init_ids": (C.c_int, [_p, _p, _sz, _p]),
    "gf_cache_init": (C.c_int, [_p, _p]),
    "gf_cache_resize": (C.c_int, [_p, _sz, _sz, _p, _p]),
    "gf_cache_fetch": (C.c_int, [_p, _p, _sz, _p, C.c_int, _p, _p]),
    "gf_cache_probe": (C.c_int, [_p, _p, _sz, _p, _p]),
    "gf_cache_fetch_pulled": (C.c_int, [_p, _p, _sz, _p, C.c_int, _p, _p, _p, _p]),
    "gf_pull_count": (C.c_int, [C.POINTER(GfPullDesc), _sz, C.c_int, _p, C.c_int, _p]),
    "gf_pull_scatter": (C.c_int, [C.POINTER(GfPullDesc), _sz, C.c_int, _p, _p, C.c_int, _p]),
    "gf_gather_rows_indexed": (C.c_int, [_p, _sz, _sz, _p, _sz, _p, _sz, _p, _p, C.c_int, _p]),
    "gf_cache_fetch_blocks_pulled": (C.c_int, [_p, _p, C.POINTER(GfFetchPulledDesc), _sz, _p]),
    "gf_pull_session_create": (C.c_int, [C.POINTER(_p), _p, C.c_int]),
    "gf_pull_session_destroy": (C.c_int, [_p]),
    "gf_pull_round": (C.c_int, [_p, _p, _p, C.POINTER(GfPullCtx), _sz, C.c_int, C.POINTER(C.c_int),
                                _p, _p, _p, _p]),
    "gf_cache_init_rows": (C.c_int, [_p, _p, _sz, _p, _p]),
    "gf_cache_fetch_blocks": (C.c_int, [_p, _p, C.POINTER(GfFetchDesc), _sz, _p]),
    "gf_cache_fetch_blocks_async": (C.c_int, [_p, _p, C.POINTER(GfFetchDesc), _sz, _p,
                                              C.POINTER(C.c_uint64)]),
    "gf_cache_fetch_wait": (C.c_int, [C.c_uint64]),
    "gf_cache_fetch_announce_async": (C.c_int, [_p, _p, C.POINTER(GfFetchDesc), _sz, _p,
                                                C.POINTER(GfFetchDesc), _sz, _p, _sz, _sz, _p,
                                                C.POINTER(C.c_uint64)]),
    "gf_cache_set_row_mirror": (C.c_int, [_p, C.c_int]),
    "gf_streams_share_queue": (C.c_int, [C.c_int, _p, _p, C.c_uint, C.POINTER(C.c_int)]),
    "gf_debug_lru_trace_enable": (C.c_int, [_p, C.c_int]),
    "gf_debug_lru_trace": (C.c_int, [_p, C.POINTER(C.c_uint64), _sz, C.POINTER(_sz)]),
    "gf_cache_set_staging": (C.c_int, [_p, _sz, _sz]),
    "gf_cache_invalidate_staging": (C.c_int, [_p]),
    "gf_cache_staging_state": (C.c_int, [_p, C.POINTER(C.c_uint64)]),
    "gf_cache_prefetch_blocks": (C.c_int, [_p, _p, C.POINTER(GfFetchDesc), _sz, _p,
                                           C.POINTER(C.c_int)]),
    "gf_cache_prefetch_blocks_async": (C.c_int, [_p, _p, C.POINTER(GfFetchDesc), _sz, _p,
                                                 C.POINTER(C.c_uint64)]),
    "gf_cache_set_staging_lag": (C.c_int, [_p, _sz]),
    "gf_memory_prepare_input": (C.c_int, [_p, _p, _p, _p, _sz, _sz, _sz, _p, _sz, _p, _p, _p, _p,
                                          C.c_int, _p]),
    "gf_memory_update": (C.c_int, [_p, _p, _p, _p, _sz, _sz, _sz, _p, _p, _p, _p, _sz, C.c_int,
                                   _p, _p, C.c_uint64, C.c_int, _p]),
    "gf_worker_stats": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "gf_gather_rows": (C.c_int, [_p, _sz, _sz, _p, _sz, _p, C.c_int, _p]),
    "gf_cache_slot_ids": (C.c_int, [_p, _p, _sz]),
    "gf_cache_mem_bytes": (C.c_int, [_p, C.POINTER(_sz)]),
    "gf_cache_lru_state": (C.c_int, [_p, C.POINTER(C.c_uint64)]),
    "gf_partition_scratch_bytes": (C.c_int, [_sz, C.c_int, C.POINTER(_sz)]),
    "gf_partition_plan": (C.c_int, [_p, _p, _sz, C.c_int, C.c_int, _p, _p, _p, _p, _sz, C.c_int,
                                    _p]),
    "gf_sampler_sample_layer_padded": (C.c_int, [_p, _p, _sz, C.c_uint32, C.c_uint32, _p, _p]),
    "gf_sampler_merge_padded": (C.c_int, [_p, _p, _p, _sz, C.c_uint32, _p, _p, _p, _sz,
                                          C.POINTER(GfBlock), _p]),
    "gf_sampler_part_layout": (C.c_int, [_p, _sz, C.c_uint32, C.c_int, C.POINTER(GfPartLayout)]),
    "gf_sampler_part_begin": (C.c_int, [_p, _p, _p, _sz, _p, _sz, C.c_int, C.c_int, _p]),
    "gf_sampler_part_plan_own": (C.c_int, [_p, C.c_uint32, C.c_uint32, _p, _sz, C.c_int]),
    "gf_sampler_part_merge": (C.c_int, [_p, C.c_uint32, C.c_uint32, _p, _sz]),
    "gf_sampler_part_layout_slotted": (C.c_int, [_p, _sz, C.c_uint32, C.c_int, C.c_double, _sz,
                                                 C.POINTER(GfPartLayout)]),
    "gf_sampler_part_group_slot": (C.c_int, [_p, _sz, C.c_uint32, C.c_int, C.c_double, _sz, C.c_int,
                                             C.c_double, C.POINTER(C.c_uint64)]),
    "gf_sampler_part_begin_slotted": (C.c_int, [_p, _p, _p, _sz, _p, _sz, C.c_int, C.c_int,
                                                C.c_double, _sz, _p]),
    "gf_sampler_part_serve": (C.c_int, [_p, C.c_uint32, C.c_uint32, _p, _sz]),
    "gf_sampler_part_overflowed": (C.c_int, [_p, C.POINTER(C.c_int)]),
    "gf_sampler_part_commit": (C.c_int, [_p]),
    "gf_sampler_part_abort": (C.c_int, [_p]),
    "gf_sampler_sample_partitioned": (C.c_int, [_p, _p, _p, _sz, _p, _sz, _p, _sz, _p]),
    "gf_sampler_sample_partitioned_async": (C.c_int, [_p, _p, _p, _sz, _p, _sz, _p, _sz, _p]),
    "gf_comm_unique_id": (C.c_int, [_p]),
    "gf_comm_create": (C.c_int, [C.POINTER(_p), _p, C.c_int, C.c_int, C.c_int]),
    "gf_comm_destroy": (C.c_int, [_p]),
    "gf_ipc_comm_create": (C.c_int, [C.POINTER(_p), C.c_int, C.c_int, C.c_int, _sz, C.c_char_p]),
    "gf_ipc_comm_handle": (C.c_int, [_p, _p]),
    "gf_ipc_comm_open": (C.c_int, [_p, _p]),
    "gf_loopback_comm_create": (C.c_int, [C.POINTER(_p), C.c_int, C.c_int]),
    "gf_comm_all_to_all": (C.c_int, [_p, _p, _p, _sz, _p]),
    "gf_comm_all_to_all_v": (C.c_int, [_p, _p, _p, _p, _p, _p, _p, _p]),
    "gf_sampler_sample_partitioned_comm": (C.c_int, [_p, _p, _p, _p, _sz, _p, _sz, _p, _sz,
                                                     C.c_double, _sz, C.c_int, _p]),
    "gf_sampler_sample_partitioned_comm_async": (C.c_int, [_p, _p, _p, _p, _sz, _p, _sz, _p, _sz,
                                                           C.c_double, _sz, C.c_int, _p]),
    "gf_sampler_part_group_ws_bytes": (C.c_int, [_p, C.POINTER(_sz), C.c_int, C.c_int, C.c_double,
                                                 _sz, C.c_int, C.POINTER(_sz)]),
    "gf_sampler_sample_partitioned_comm_group": (C.c_int, [_p, _p, C.c_int, _p, _sz, C.c_double,
                                                           _sz, C.c_int, C.c_int, _p]),
    "gf_sampler_sample_partitioned_comm_group_async": (C.c_int, [_p, _p, C.c_int, _p, _sz,
                                                                 C.c_double, _sz, C.c_int, C.c_int,
                                                                 _p]),
    "gf_block_segment_offsets": (C.c_int, [_p, _sz, _sz, _p, C.c_int, _p]),
    "gf_block_edge_softmax": (C.c_int, [_p, _sz, _sz, _sz, _p, _p, C.c_int, _p]),
    "gf_block_edge_softmax_backward": (C.c_int, [_p, _sz, _sz, _sz, _p, _p, _p, C.c_int, _p]),
    "gf_block_reduce": (C.c_int, [_p, _sz, _p, _p, _sz, _p, _sz, C.c_int, _p, C.c_int, _p]),
    "gf_block_reduce_backward": (C.c_int, [_p, _sz, _p, _p, _sz, _p, _sz, C.c_int, _p, _p, _sz,
                                           _p, C.c_int, _p]),
    "gf_block_reduce_max": (C.c_int, [_p, _sz, _p, _p, _sz, _p, _p, C.c_int, _p]),
    "gf_block_reduce_max_backward": (C.c_int, [_sz, _p, _sz, _p, _p, _p, _sz, C.c_int, _p]),
    "gf_debug_part_host_us": (C.c_int, [C.POINTER(C.c_double), C.c_int]),
    "gf_debug_merge_recounts": (C.c_int, [C.POINTER(C.c_uint64)]),
    "gf_debug_part_reused_roots": (C.c_int, [C.POINTER(C.c_uint64)]),
    "gf_debug_philox": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "gf_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32)]),
    "gf_comm_abort": (C.c_int, [C.c_void_p]),
    "gf_comm_time_all_to_all": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "gf_device_pci_bus_id": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
    "gf_debug_lru_recounts": (C.c_int, [C.POINTER(C.c_uint64)]),
    "gf_profile_enable": (C.c_int, [C.c_int]),
    "gf_profile_reset": (C.c_int, []),
    "gf_profile_get": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]),
    "gf_profile_set_stride": (C.c_int, [C.c_uint]),
    "gf_profile_launches": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
}

_raw_stream = None


def current_stream(device) -> C.c_void_p:
    """torch's current stream on `device` (a torch.device with an index) as the C ABI takes it.
    The raw-handle query costs ~0.3 us of host time; torch.cuda.current_stream() builds a
    Stream object (~6 us in the batch-600 loop's profile; the step itself did not change —
    the loop is not bound by this thread, profiles/README.md round 4)."""
    global _raw_stream
    if _raw_stream is None:
        import torch
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or \
            (lambda index: torch.cuda.current_stream(index).cuda_stream)
    return C.c_void_p(_raw_stream(device.index))


_lib = None


def load():
    """Loads libgnnflow_hip.so and binds every prototype.  Raises if missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "gnnflow_amd: HIP library not built ({}). Run `python -c 'import "
            "__graft_entry__ as g; g.build()'` or `python gnnflow_amd/_build.py`. "
            "There is no CPU fallback.".format(LIB_PATH))
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class GnnflowError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def check(rc):
    """Maps a GF_ERR_* status to the exception the reference's Python layer documents."""
    if rc == GF_OK:
        return
    msg = load().gf_last_error().decode("utf-8", "replace")
    if rc in (GF_ERR_INVALID_ARGUMENT, GF_ERR_TIMESTAMP_ORDER):
        raise ValueError(msg)
    if rc == GF_ERR_OUT_OF_MEMORY:
        raise MemoryError(msg)
    raise GnnflowError(rc, msg)
