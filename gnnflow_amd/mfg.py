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
    neighbours, edges (src index -> dst index) = (col, row) of the SamplingResult."""

    def __init__(self, num_src_nodes: int, num_dst_nodes: int, col, row, keepalive=None,
                 num_edges=None, device=None):
        """col / row: tensors, or zero-argument callables that build them on first use."""
        self._num_src = int(num_src_nodes)
        self._num_dst = int(num_dst_nodes)
        self._col, self._row = col, row
        self._num_edges = int(num_edges) if num_edges is not None else int(row.shape[0])
        self._device = device if device is not None else row.device
        self.srcdata: Dict[str, torch.Tensor] = LazyTensorDict()
        self.dstdata: Dict[str, torch.Tensor] = LazyTensorDict()
        self.edata: Dict[str, torch.Tensor] = LazyTensorDict()
        self._keepalive = keepalive

    def num_src_nodes(self) -> int:
        return self._num_src

    def num_dst_nodes(self) -> int:
        return self._num_dst

    def num_edges(self) -> int:
        return self._num_edges

    def edges(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(source node index, destination node index) per edge, as dgl's edges()."""
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
        if self._keepalive is not None:
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
