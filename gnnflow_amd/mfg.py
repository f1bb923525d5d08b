"""Message-flow-graph block: the subset of dgl.heterograph.DGLBlock that the
reference's sampler fills and its callers read (gnnflow/temporal_sampler.py:149-165,
gnnflow/utils.py:465-481, gnnflow/cache/cache.py:272-400), holding device-resident
torch tensors that alias the sampler's output buffer (no copies, no DGL)."""
from typing import Dict, Tuple

import torch


class LazyTensorDict(dict):
    """dict whose well-known entries are created on first access.  The sampler registers
    thunks for 'ID' / 'ts' / 'dt' (views into its output buffer); building every view
    eagerly costs more host time per batch than the sampling kernels themselves."""

    def __init__(self):
        super().__init__()
        self._thunks = {}

    def set_lazy(self, key, thunk):
        self._thunks[key] = thunk

    def __missing__(self, key):
        thunk = self._thunks.pop(key, None)
        if thunk is None:
            raise KeyError(key)
        value = thunk()
        self[key] = value
        return value

    def materialize(self):
        for key in list(self._thunks):
            self[key]
        return self

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._thunks

    def get(self, key, default=None):
        return self[key] if key in self else default

    def keys(self):
        return self.materialize() and dict.keys(self)

    def items(self):
        self.materialize()
        return dict.items(self)

    def values(self):
        self.materialize()
        return dict.values(self)

    def __iter__(self):
        self.materialize()
        return dict.__iter__(self)

    def __len__(self):
        return dict.__len__(self) + len(self._thunks)


class MFGBlock:
    """Bipartite block: `num_dst_nodes` roots, `num_src_nodes` = roots ++ sampled
    neighbours, edges (src index -> dst index) = (col, row) of the SamplingResult.

    Blocks built by the sampler carry only the device addresses of their arrays
    (`raw`); the torch views — `srcdata['ID' / 'ts']`, `edata['dt' / 'ID']`, `edges()` — and
    the three data dicts themselves are created on first access, and the feature cache
    reads the id arrays through `raw_ids()` without creating any."""

    __slots__ = ("_num_src", "_num_dst", "_col", "_row", "_num_edges", "_device",
                 "_srcdata", "_dstdata", "_edata", "_keepalive", "_raw", "_pending", "_sample_blocks",
                 "_segments", "_edge_prefix_of", "_stream_marks")

    def __init__(self, num_src_nodes: int, num_dst_nodes: int, col=None, row=None,
                 keepalive=None, num_edges=None, device=None, raw=None):
        """col / row: tensors, or zero-argument callables that build them on first use.
        raw: (view_fn, nodes_ptr, ts_ptr, dt_ptr, eid_ptr, col_ptr, row_ptr) with
        view_fn(ptr, count, dtype, itemsize) -> tensor aliasing the sampler output."""
        self._num_src = int(num_src_nodes)
        self._num_dst = int(num_dst_nodes)
        self._col, self._row = col, row
        self._num_edges = int(num_edges) if num_edges is not None else int(row.shape[0])
        self._device = device if device is not None else row.device
        self._srcdata = self._dstdata = self._edata = None
        self._keepalive = keepalive
        self._raw = raw
        # first block of a sample(): (bytes of the sample's gf_block array, layers, snapshots) —
        # lets Cache announce the whole sample to its staging ring without touching the blocks
        self._sample_blocks = None
        self._pending = None     # [(which dict, key, thunk)] registered before the dicts exist
        self._segments = None    # (offsets[num_dst + 1], col grouped by destination, perm)
        # set by TemporalSampler: the block of the NEXT sampled layer whose edge arrays start
        # with this block's edge arrays (same roots, same windows, same fanout); see
        # Cache.fetch_feature
        self._edge_prefix_of = None
        # shared by all blocks carved out of one sampler output slab: streams that were
        # already told (record_stream) that they use that allocation
        self._stream_marks = None

    # ---- data dicts, created on demand ----------------------------------------------
    def _make(self, which):
        d = LazyTensorDict()
        raw = self._raw
        if raw is not None:
            view = raw[0]
            if which == "src":
                ns = self._num_src
                d.set_lazy('ID', lambda: view(raw[1], ns, torch.int64, 8))
                d.set_lazy('ts', lambda: view(raw[2], ns, torch.float32, 4))
            elif which == "e":
                ne = self._num_edges
                d.set_lazy('dt', lambda: view(raw[3], ne, torch.float32, 4))
                d.set_lazy('ID', lambda: view(raw[4], ne, torch.int64, 8))
        if self._pending:
            keep = []
            for w, key, thunk in self._pending:
                if w == which:
                    d.set_lazy(key, thunk)
                else:
                    keep.append((w, key, thunk))
            self._pending = keep or None
        return d

    @property
    def srcdata(self) -> Dict[str, torch.Tensor]:
        if self._srcdata is None:
            self._srcdata = self._make("src")
        return self._srcdata

    @srcdata.setter
    def srcdata(self, value):
        self._srcdata = value

    @property
    def dstdata(self) -> Dict[str, torch.Tensor]:
        if self._dstdata is None:
            self._dstdata = self._make("dst")
        return self._dstdata

    @dstdata.setter
    def dstdata(self, value):
        self._dstdata = value

    @property
    def edata(self) -> Dict[str, torch.Tensor]:
        if self._edata is None:
            self._edata = self._make("e")
        return self._edata

    @edata.setter
    def edata(self, value):
        self._edata = value

    def set_lazy(self, which: str, key: str, thunk):
        """Registers `thunk` as the producer of srcdata / edata [key] (which = 'src' | 'e')
        without creating the dict."""
        d = self._srcdata if which == "src" else self._edata
        if d is not None and hasattr(d, "set_lazy"):
            dict.pop(d, key, None)
            d.set_lazy(key, thunk)
        elif d is not None:
            d[key] = thunk()
        else:
            if self._pending is None:
                self._pending = []
            self._pending.append((which, key, thunk))

    def raw_ids(self, which: str):
        """(device address, count) of srcdata['ID'] ('src') or edata['ID'] ('e') if they are
        still the sampler's own arrays, else None (the caller then reads the tensor)."""
        raw = self._raw
        if raw is None:
            return None
        d = self._srcdata if which == "src" else self._edata
        if d is not None and dict.__contains__(d, 'ID'):
            return None        # materialised — possibly replaced by the caller
        return (raw[1], self._num_src) if which == "src" else (raw[4], self._num_edges)

    def segments(self):
        """(offsets, col, perm): edges grouped by destination node — offsets[d]..offsets[d+1]
        are the edges into d, `col` their source indices in that order (None for the sampler's
        own layout, source of edge k = node num_dst + k), `perm` the edge permutation that
        groups them (None when the edges already are, as the sampler's)."""
        if self._segments is None:
            import ctypes as C
            from . import _capi
            col, row = self.edges()
            perm = None
            sampler_layout = self._raw is not None   # col[k] = num_dst + k by construction
            if self._raw is None and self._num_edges > 1 and \
                    not bool((row[1:] >= row[:-1]).all()):
                perm = torch.argsort(row, stable=True)
                col, row = col[perm], row[perm]
            col, row = col.contiguous(), row.contiguous()
            offsets = torch.empty(self._num_dst + 1, dtype=torch.int64, device=self._device)
            with torch.cuda.device(self._device):
                _capi.check(_capi.load().gf_block_segment_offsets(
                    row.data_ptr() if self._num_edges else None, self._num_edges, self._num_dst,
                    offsets.data_ptr(), self._device.index,
                    _capi.current_stream(self._device)))
            self._segments = (offsets, None if sampler_layout else col, perm)
        return self._segments

    def update_all(self, message_func, reduce_func):
        """dgl's block.update_all for the message / reduce pairs of gnnflow_amd.function:
        dstdata[reduce.out] = reduce over in-edges of message (layers.py:159)."""
        from . import ops
        if reduce_func.msg != message_func.out:
            raise KeyError("reducer reads '{}' but the message is '{}'".format(
                reduce_func.msg, message_func.out))
        weight = self.edata[message_func.edge] if message_func.kind == "u_mul_e" else None
        if reduce_func.kind == "max":
            if weight is not None:
                raise NotImplementedError("update_all(u_mul_e, max)")
            self.dstdata[reduce_func.out] = ops.block_max(self, self.srcdata[message_func.src])
            return
        self.dstdata[reduce_func.out] = ops.block_reduce(
            self, self.srcdata[message_func.src], weight, mean=reduce_func.kind == "mean")

    def in_degrees(self) -> torch.Tensor:
        offsets = self.segments()[0]
        return offsets[1:] - offsets[:-1]

    def num_src_nodes(self) -> int:
        return self._num_src

    def num_dst_nodes(self) -> int:
        return self._num_dst

    def num_edges(self) -> int:
        return self._num_edges

    def edges(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(source node index, destination node index) per edge, as dgl's edges()."""
        if self._col is None and self._raw is not None:
            view, ne = self._raw[0], self._num_edges
            self._col = view(self._raw[5], ne, torch.int64, 8)
            self._row = view(self._raw[6], ne, torch.int64, 8)
        if callable(self._col):
            self._col = self._col()
        if callable(self._row):
            self._row = self._row()
        return self._col, self._row

    @property
    def device(self) -> torch.device:
        return self._device

    def record_stream(self, stream):
        """Marks the sampler output buffer behind this block as in use on `stream`
        (needed when the block was sampled on a side stream, e.g. by a prefetch thread,
        and is consumed on another one)."""
        if self._keepalive is None:
            return
        marks = self._stream_marks
        if marks is not None:
            # record_stream works on the whole allocation (the slab): once per slab and stream
            key = stream.cuda_stream
            if key in marks:
                return
            marks.add(key)
        self._keepalive.record_stream(stream)

    def to(self, device, **kwargs):
        device = torch.device(device)
        if device == self.device:
            return self
        col, row = self.edges()
        b = MFGBlock(self._num_src, self._num_dst, col.to(device, **kwargs),
                     row.to(device, **kwargs))
        b.srcdata = {k: v.to(device, **kwargs) for k, v in self.srcdata.items()}
        b.dstdata = {k: v.to(device, **kwargs) for k, v in self.dstdata.items()}
        b.edata = {k: v.to(device, **kwargs) for k, v in self.edata.items()}
        return b

    def __repr__(self):
        return "MFGBlock(num_src_nodes={}, num_dst_nodes={}, num_edges={}, device={})".format(
            self._num_src, self._num_dst, self.num_edges(), self.device)
