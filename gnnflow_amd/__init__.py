"""gnnflow_amd — MI355X-native temporal neighbour sampler + feature gather for GNNFlow.

Drop-in for the hot path of jasperzhong/GNNFlow: `DynamicGraph`, `TemporalSampler`,
`cache.LRUCache` keep the reference's Python API; the work runs in hand-written HIP
kernels (gfx950) behind the C ABI of include/gnnflow_hip.h.
"""
from .dynamic_graph import DynamicGraph
from .mfg import MFGBlock
from .temporal_sampler import SamplingResult, TemporalSampler

__all__ = ["DynamicGraph", "TemporalSampler", "SamplingResult", "MFGBlock"]
