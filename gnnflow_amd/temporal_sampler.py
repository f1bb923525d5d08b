"""gnnflow.TemporalSampler on MI355X — same constructor, methods and MFG layout as the
reference's Python wrapper (gnnflow/temporal_sampler.py:14-177); all layers and
snapshots are sampled in one device-resident call (gf_sampler_sample) and the returned
blocks hold torch tensors in HBM, so mfgs_to_cuda (gnnflow/utils.py:477-481) is a no-op."""
import ctypes as C
import struct
from collections import deque
from typing import List, Union

import numpy as np
import torch

from . import _capi
from .dynamic_graph import DynamicGraph
from .mfg import MFGBlock


class _NullCtx:
    def __enter__(self): return self
    def __exit__(self, *a): return False


class SamplingResult:
    """libgnnflow.SamplingResult look-alike (gnnflow/csrc/api.cc:87-109): accessor
    *methods* returning host numpy arrays; `*_tensor()` give the device tensors."""

    def __init__(self, block: MFGBlock):
        self._b = block

    def row(self): return self._b.edges()[1].cpu().numpy()
    def col(self): return self._b.edges()[0].cpu().numpy()
    def all_nodes(self): return self._b.srcdata['ID'].cpu().numpy()
    def all_timestamps(self): return self._b.srcdata['ts'].cpu().numpy()
    def delta_timestamps(self): return self._b.edata['dt'].cpu().numpy()
    def eids(self): return self._b.edata['ID'].cpu().numpy()
    def num_src_nodes(self): return self._b.num_src_nodes()
    def num_dst_nodes(self): return self._b.num_dst_nodes()
    def block(self) -> MFGBlock: return self._b


class PendingSample:
    """Handle of an in-flight TemporalSampler.sample_async().  Samples complete in the
    order they were begun: wait() first completes the older ones."""

    def __init__(self, sampler, buf, inputs, num_roots, marks):
        self._sampler, self._buf, self._inputs, self._R = sampler, buf, inputs, num_roots
        self._marks = marks
        self._result = None
        self._error = None
        self._after_end = None   # called right after the native sample_end of THIS sample

    def wait(self) -> List[List[MFGBlock]]:
        if self._error is not None:
            raise self._error
        q = self._sampler._inflight
        while self._result is None:
            head = q[0]
            if head is not self:
                head.wait()      # older sample first (native FIFO)
                continue
            q.popleft()          # whatever happens, the native side has popped it too
            try:
                self._result = self._sampler._finish(self._buf, self._R, self._marks)
                if self._after_end is not None:
                    self._after_end()
            except Exception as e:   # the failed sample must not wedge the sampler
                self._error = e
                raise
            finally:
                self._inputs = None
        return self._result


class TemporalSampler:
    """
    TemporalSampler samples k-hop multi-snapshots neighbors of given vertices.
    """

    def __init__(
            self, graph: DynamicGraph, fanouts: List[int],
            sample_strategy: str = "recent", num_snapshots: int = 1,
            snapshot_time_window: float = 0.0, prop_time: bool = False,
            seed: int = 1234, *args, **kwargs):
        sample_strategy = sample_strategy.lower()
        if sample_strategy not in ["recent", "uniform"]:
            raise ValueError("strategy must be 'recent' or 'uniform'")

        self._lib = _capi.load()
        self._strategy = sample_strategy
        self._graph = graph  # keep the graph alive (temporal_sampler.h:62 holds a ref)
        self._device = torch.device("cuda", graph.device)
        self._fanouts = [int(f) for f in fanouts]
        self._num_layers = len(self._fanouts)
        self._num_snapshots = int(num_snapshots)
        self._h = C.c_void_p()
        arr = (C.c_uint32 * self._num_layers)(*self._fanouts)
        _capi.check(self._lib.gf_sampler_create(
            C.byref(self._h), graph._h, arr, self._num_layers,
            _capi.SAMPLING_POLICY[sample_strategy], self._num_snapshots,
            float(snapshot_time_window), 1 if prop_time else 0, int(seed)))
        self._is_static = bool(kwargs.get('is_static', False))
        self._ctor = dict(fanouts=list(self._fanouts), sample_strategy=sample_strategy,
                          num_snapshots=self._num_snapshots,
                          snapshot_time_window=float(snapshot_time_window),
                          prop_time=bool(prop_time), seed=int(seed), is_static=self._is_static)
        self._bytes_cache = {}
        self._inflight = deque()       # PendingSamples, oldest first
        self._max_inflight = 4         # gf::Sampler::kMaxInFlight
        self._slab = None
        self._gf_blocks = (_capi.GfBlock * (self._num_layers * self._num_snapshots))()
        # struct gf_block = six addresses + three counts (include/gnnflow_hip.h)
        self._blocks_unpack = struct.Struct(
            "=%dQ" % (9 * self._num_layers * self._num_snapshots)).unpack_from

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            self._lib.gf_sampler_destroy(h)
            self._h = None

    def clone(self) -> "TemporalSampler":
        """A second sampler over the same graph with the same arguments: its own native
        workspace, publish ring and draw counter, so the two can have samples in flight on
        different streams at once (the lanes of gnnflow_amd.dist.DevicePartitionedSampler)."""
        return TemporalSampler(self._graph, **self._ctor)

    # ---- helpers ------------------------------------------------------------------
    def _to_device(self, target_vertices, timestamps, stream=None):
        """int64 roots / float32 timestamps on the device, contiguous.  Any conversion is
        issued on `stream` — the stream the sampling kernels will read them on."""
        nodes, ts = target_vertices, timestamps
        ok_n = isinstance(nodes, torch.Tensor) and nodes.device == self._device and \
            nodes.dtype == torch.int64 and nodes.is_contiguous()
        ok_t = self._is_static or (
            isinstance(ts, torch.Tensor) and ts.device == self._device and
            ts.dtype == torch.float32 and ts.is_contiguous())
        if not (ok_n and ok_t) or self._is_static:
            ctx = torch.cuda.stream(stream) if stream is not None else _NullCtx()
            with ctx:
                if not ok_n:
                    if isinstance(nodes, torch.Tensor):
                        nodes = nodes.to(self._device, torch.int64).contiguous()
                    else:
                        nodes = torch.from_numpy(
                            np.ascontiguousarray(nodes, dtype=np.int64)).to(self._device)
                if self._is_static:
                    # gnnflow/temporal_sampler.py:72-76
                    ts = torch.full((nodes.shape[0],), float(np.finfo(np.float32).max),
                                    dtype=torch.float32, device=self._device)
                elif not ok_t:
                    if isinstance(ts, torch.Tensor):
                        ts = ts.to(self._device, torch.float32).contiguous()
                    else:
                        ts = torch.from_numpy(
                            np.ascontiguousarray(ts, dtype=np.float32)).to(self._device)
        assert nodes.dim() == 1 and ts.shape == nodes.shape, \
            "target_vertices and timestamps must be 1D and of equal length"
        return nodes, ts

    def _block(self, buf, gb, marks=None, base=None) -> MFGBlock:
        """MFG over one gf_block; the tensor views into `buf` (the output buffer, or the slab
        it was carved from, whose address is `base`) are built on first access."""
        if gb.all_nodes is None:
            raise RuntimeError("sampler returned a null block")
        device = self._device
        if base is None:
            base = buf.data_ptr()

        def view(ptr, count, dtype, itemsize):
            if count == 0:
                return torch.empty(0, dtype=dtype, device=device)
            off = ptr - base
            return buf[off:off + count * itemsize].view(dtype)

        b = MFGBlock(gb.num_src_nodes, gb.num_dst_nodes, keepalive=buf,
                     num_edges=gb.num_edges, device=device,
                     raw=(view, gb.all_nodes, gb.all_timestamps, gb.delta_timestamps,
                          gb.eids, gb.col, gb.row))
        b._stream_marks = marks
        return b

    def _empty_block(self) -> MFGBlock:
        e64 = torch.empty(0, dtype=torch.int64, device=self._device)
        e32 = torch.empty(0, dtype=torch.float32, device=self._device)
        b = MFGBlock(0, 0, e64, e64)
        b.srcdata['ID'], b.srcdata['ts'] = e64, e32
        b.edata['dt'], b.edata['ID'] = e32, e64
        return b

    def _stream(self):
        return _capi.current_stream(self._device)

    # ---- reference API ------------------------------------------------------------
    def sample(self, target_vertices: np.ndarray, timestamps: np.ndarray) -> List[List[MFGBlock]]:
        """
        Sample k-hop neighbors of given vertices.

        Returns:
            list of message flow graphs (# of graphs = # of snapshots) for each layer;
            mfgs[0] is the last-sampled (largest) layer, mfgs[-1] the roots' layer
            (gnnflow/temporal_sampler.py:149-165).
        """
        return self.sample_async(target_vertices, timestamps).wait()

    def set_enqueue_lane(self, lane: int):
        """Which sampling enqueue thread of the library issues this sampler's asynchronous samples
        (include/gnnflow_hip.h gf_sampler_set_enqueue_lane)."""
        _capi.check(self._lib.gf_sampler_set_enqueue_lane(self._h, int(lane)))

    def call_counter(self) -> int:
        """Number of sample_layer invocations so far = the `call` word of the next uniform draw
        (include/gnnflow_hip.h gf_sampler_call_counter).  No sample may be in flight."""
        n = C.c_uint64(0)
        _capi.check(self._lib.gf_sampler_call_counter(self._h, C.byref(n)))
        return int(n.value)

    def set_call_counter(self, value: int, through_enqueue_thread: bool = False):
        """The next begun sample draws from call number `value` on."""
        _capi.check(self._lib.gf_sampler_set_call_counter(
            self._h, int(value), 1 if through_enqueue_thread else 0))

    @property
    def calls_per_sample(self) -> int:
        return self._num_layers * self._num_snapshots

    def sample_async(self, target_vertices, timestamps, stream=None,
                     worker_enqueue=False, call_base=None) -> "PendingSample":
        """Enqueues sample() on `stream` (default: the current stream) and returns at
        once; `.wait()` blocks until the kernels finished and returns the MFGs.  Lets a
        single Python thread overlap the sampling of batch i+1 with the feature fetch /
        training of batch i (the reference uses a prefetch thread for this,
        scripts/offline_edge_prediction.py:343-346).  One sample in flight per sampler.
        worker_enqueue=True lets the library's enqueue thread issue the launches.
        call_base: the call number this sample's uniform draws start from (default: where the
        sampler's previous sample left off) — lets clones of a sampler take turns and reproduce
        the one sampler's stream (gnnflow_amd.pipeline.ReplayPipeline's lanes)."""
        if len(self._inflight) >= self._max_inflight:
            self._inflight[0].wait()   # the native ring holds kMaxInFlight begun samples
        if call_base is not None:
            rc = self._lib.gf_sampler_set_call_counter(self._h, call_base,
                                                       1 if worker_enqueue else 0)
            if rc:
                _capi.check(rc)
        if stream is None:
            stream = torch.cuda.current_stream(self._device)
        nodes, ts = self._to_device(target_vertices, timestamps, stream)
        R = int(nodes.shape[0])
        slab = None
        if R:
            nbytes = self._bytes_cache.get(R)
            if nbytes is None:
                n = C.c_size_t(0)
                _capi.check(self._lib.gf_sampler_output_bytes(self._h, R, C.byref(n)))
                nbytes = self._bytes_cache[R] = n.value
            slab, off = self._output_buffer(nbytes, stream)
            begin = self._lib.gf_sampler_sample_begin_async if worker_enqueue \
                else self._lib.gf_sampler_sample_begin
            rc = begin(self._h, nodes.data_ptr(), ts.data_ptr(), R, slab[5] + off, nbytes, slab[6])
        else:
            begin = self._lib.gf_sampler_sample_begin_async if worker_enqueue \
                else self._lib.gf_sampler_sample_begin
            rc = begin(self._h, None, None, 0, None, 0, C.c_void_p(stream.cuda_stream))
        if rc:
            _capi.check(rc)
        pending = PendingSample(self, slab, (nodes, ts), R, None)
        self._inflight.append(pending)
        return pending

    def _output_buffer(self, nbytes, stream):
        """Output memory for one sample(), owned by `stream` in the caching allocator: a slab
        record [tensor, stream, step, used, stream marks, address, stream handle] and the byte
        offset of this call's share.  Small outputs share a slab that is allocated once per 16
        calls (switching torch's current stream for a torch.empty costs more host time than a
        batch-600 sampling kernel runs); no per-call tensor is created — the blocks' views are
        cut out of the slab when they are first read."""
        per_slab = min(16, (256 << 20) // max(nbytes, 1))
        step = (nbytes + 255) & ~255
        if per_slab < 2:
            per_slab, step = 1, nbytes
        slab = self._slab
        # (identity first: torch.cuda.Stream.__ne__ is a Python-level call)
        if slab is None or (slab[1] is not stream and slab[1] != stream) or slab[2] != step \
                or slab[3] >= per_slab:
            with torch.cuda.stream(stream):
                mem = torch.empty(step * per_slab, dtype=torch.uint8, device=self._device)
            # slab[4]: streams already told (record_stream) that they use this allocation
            slab = self._slab = [mem, stream, step, 0, set(), mem.data_ptr(),
                                 C.c_void_p(stream.cuda_stream)]
        i = slab[3]
        slab[3] = i + 1
        return slab, i * step

    def _finish(self, buf, R, marks=None) -> List[List[MFGBlock]]:
        blocks = self._gf_blocks
        rc = self._lib.gf_sampler_sample_end(self._h, blocks)
        if rc:
            _capi.check(rc)
        if R == 0:
            return [[self._empty_block() for _ in range(self._num_snapshots)]
                    for _ in range(self._num_layers)]
        ns = self._num_snapshots
        base = None
        if isinstance(buf, list):          # a slab record of _output_buffer
            buf, marks, base = buf[0], buf[4], buf[5]
        # every field of every gf_block in ONE unpack (nine ctypes field reads per block cost more
        # host time than the block's sampling kernel runs), one view function for the sample
        vals = self._blocks_unpack(blocks)
        device = self._device
        if base is None:
            base = buf.data_ptr()

        def view(ptr, count, dtype, itemsize):
            if count == 0:
                return torch.empty(0, dtype=dtype, device=device)
            off = ptr - base
            return buf[off:off + count * itemsize].view(dtype)

        mfgs = []
        k = 0
        for layer in range(self._num_layers):
            row_ = []
            for s in range(ns):
                an, ats, dts, eids, row, col, ndst, nsrc, ne = vals[k:k + 9]
                k += 9
                if not an:
                    raise RuntimeError("sampler returned a null block")
                b = MFGBlock(nsrc, ndst, keepalive=buf, num_edges=ne, device=device,
                             raw=(view, an, ats, dts, eids, col, row))
                b._stream_marks = marks
                row_.append(b)
            mfgs.append(row_)
        if self._strategy == "recent":
            # Layer l+1's roots start with layer l's roots (all_nodes = roots ++ neighbours,
            # temporal_sampler.cu:294-299) carrying the same timestamps, so with the same
            # fanout most-recent sampling picks the same edges for them again and emits them
            # first (root-major order): block l's edge arrays are a PREFIX of block l+1's.
            fan = self._fanouts
            for layer in range(self._num_layers - 1):
                if fan[layer] == fan[layer + 1]:
                    for s in range(ns):
                        mfgs[layer][s]._edge_prefix_of = mfgs[layer + 1][s]
        mfgs.reverse()
        # (a private copy of the block array: `blocks` is reused by this sampler's next sample)
        mfgs[0][0]._sample_blocks = (type(blocks).from_buffer_copy(blocks), self._num_layers, ns)
        return mfgs

    def sample_layer(self, target_vertices: np.ndarray, timestamps: np.ndarray,
                     layer: int, snapshot: int, to_dgl_block: bool = True) \
            -> Union[MFGBlock, SamplingResult]:
        """
        Sample neighbors of given vertices in a specific layer and snapshot
        (gnnflow/temporal_sampler.py:124-147).
        """
        nodes, ts = self._to_device(target_vertices, timestamps)
        R = int(nodes.shape[0])
        gb = _capi.GfBlock()
        if R == 0:
            _capi.check(self._lib.gf_sampler_sample_layer(
                self._h, None, None, 0, int(layer), int(snapshot), None, 0, C.byref(gb),
                self._stream()))
            b = self._empty_block()
        else:
            n = C.c_size_t(0)
            _capi.check(self._lib.gf_sampler_layer_output_bytes(self._h, R, int(layer),
                                                                C.byref(n)))
            with torch.cuda.device(self._device):
                buf = torch.empty(n.value, dtype=torch.uint8, device=self._device)
                _capi.check(self._lib.gf_sampler_sample_layer(
                    self._h, nodes.data_ptr(), ts.data_ptr(), R, int(layer), int(snapshot),
                    buf.data_ptr(), n.value, C.byref(gb), self._stream()))
            b = self._block(buf, gb)
        return b if to_dgl_block else SamplingResult(b)

    def _sample(self, target_vertices: np.ndarray, timestamps: np.ndarray, sort: bool = False):
        """benchmarks/benchmark_sampler.py:83 entry point: [layer][snapshot]
        SamplingResults (not reversed) and the time spent sorting roots."""
        import time
        results, sort_time = [], 0.0
        for layer in range(self._num_layers):
            layer_results = []
            for snapshot in range(self._num_snapshots):
                if layer == 0:
                    n, t = target_vertices, timestamps
                else:
                    prev = results[layer - 1][snapshot].block()
                    n, t = prev.srcdata['ID'], prev.srcdata['ts']
                if sort:
                    s0 = time.time()
                    if isinstance(n, torch.Tensor):
                        idx = torch.argsort(n)
                    else:
                        idx = np.argsort(n)
                    n, t = n[idx], t[idx]
                    sort_time += time.time() - s0
                layer_results.append(self.sample_layer(n, t, layer, snapshot,
                                                       to_dgl_block=False))
            results.append(layer_results)
        return results, sort_time
