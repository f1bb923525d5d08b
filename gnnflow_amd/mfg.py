"""Message-flow-graph block: the subset of dgl.heterograph.DGLBlock that the
reference's sampler fills and its callers read (gnnflow/temporal_sampler.py:149-165,
gnnflow/utils.py:465-481, gnnflow/cache/cache.py:272-400), holding device-resident
torch tensors that alias the sampler's output buffer (no copies, no DGL)."""
from typing import Dict, Tuple

import torch


class MFGBlock:
    """Bipartite block: `num_dst_nodes` roots, `num_src_nodes` = roots ++ sampled
    neighbours, edges (src index -> dst index) = (col, row) of the SamplingResult."""

    def __init__(self, num_src_nodes: int, num_dst_nodes: int, col: torch.Tensor,
                 row: torch.Tensor, keepalive=None):
        self._num_src = int(num_src_nodes)
        self._num_dst = int(num_dst_nodes)
        self._col, self._row = col, row
        self.srcdata: Dict[str, torch.Tensor] = {}
        self.dstdata: Dict[str, torch.Tensor] = {}
        self.edata: Dict[str, torch.Tensor] = {}
        self._keepalive = keepalive

    def num_src_nodes(self) -> int:
        return self._num_src

    def num_dst_nodes(self) -> int:
        return self._num_dst

    def num_edges(self) -> int:
        return int(self._row.shape[0])

    def edges(self) -> Tuple[torch.Tensor, torch.Tensor]:
        """(source node index, destination node index) per edge, as dgl's edges()."""
        return self._col, self._row

    @property
    def device(self) -> torch.device:
        return self._row.device

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
        b = MFGBlock(self._num_src, self._num_dst, self._col.to(device, **kwargs),
                     self._row.to(device, **kwargs))
        b.srcdata = {k: v.to(device, **kwargs) for k, v in self.srcdata.items()}
        b.dstdata = {k: v.to(device, **kwargs) for k, v in self.dstdata.items()}
        b.edata = {k: v.to(device, **kwargs) for k, v in self.edata.items()}
        return b

    def __repr__(self):
        return "MFGBlock(num_src_nodes={}, num_dst_nodes={}, num_edges={}, device={})".format(
            self._num_src, self._num_dst, self.num_edges(), self.device)
