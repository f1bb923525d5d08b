"""The reference's other cache policies on the same fused gather:
LFUCache (gnnflow/cache/lfu_cache.py), FIFOCache (fifo_cache.py) and GNNLabStaticCache
(gnnlab_static_cache.py).  The replacement itself is native (gf_cache_set_policy)."""
import numpy as np
import torch

from .. import _capi
from .cache import Cache


class LFUCache(Cache):
    """
    Least-frequently-used (LFU) cache: a hit adds 1 to the slot's use count (in blocks that
    also miss), the k smallest counts are evicted, a new entry starts at 1
    (lfu_cache.py:134-210).
    """
    _policy = "lfu"

    def __init__(self, *args, **kwargs):
        super(LFUCache, self).__init__(*args, **kwargs)
        self.name = 'lfu'

    def reset(self):
        """NB: only the edge cache is reset, and its use counts restart at 0 — not at the 1
        init_cache() leaves (lfu_cache.py:86-118)."""
        self.wait_enqueued()
        if self._edge is not None:
            with torch.cuda.device(self.device):
                self._edge.init(self._stream())
                _capi.check(self._lib.gf_cache_reset_order(self._edge.h, self._stream()))


class FIFOCache(Cache):
    """
    First-in-first-out (FIFO) cache: slots are refilled in rotation, hits change nothing
    (fifo_cache.py:77-161).
    """
    _policy = "fifo"

    def __init__(self, *args, **kwargs):
        super(FIFOCache, self).__init__(*args, **kwargs)
        self.name = 'fifo'

    def reset(self):
        """The reference only rewinds the edge pointer (fifo_cache.py:70-75): the cached
        ids stay, the rotation restarts at slot 0."""
        self.wait_enqueued()
        if self._edge is not None:
            with torch.cuda.device(self.device):
                _capi.check(self._lib.gf_cache_reset_order(self._edge.h, self._stream()))


class GNNLabStaticCache(Cache):
    """
    GNNLab static cache: pre-sample a few epochs, cache the most frequently sampled ids once,
    never replace (gnnlab_static_cache.py:87-182).
    """

    def __init__(self, *args, **kwargs):
        super(GNNLabStaticCache, self).__init__(*args, **kwargs)
        self.name = 'gnnlab'

    def init_cache(self, *args, **kwargs):
        """kwargs: sampler, train_df (columns src, dst, time), pre_sampling_rounds=2,
        batch_size=600 — as the reference (gnnlab_static_cache.py:96-118).  Ties among
        equally often sampled ids go to the lowest id (torch.topk leaves them unspecified)."""
        sampler = kwargs['sampler']
        train_df = kwargs['train_df']
        rounds = kwargs.get('pre_sampling_rounds', 2)
        batch_size = kwargs.get('batch_size', 600)
        dev = self.device
        node_cnt = torch.zeros(self.num_nodes, dtype=torch.int32, device=dev)
        edge_cnt = torch.zeros(self.num_edges, dtype=torch.int32, device=dev)
        src = np.asarray(train_df['src'], dtype=np.int64)
        dst = np.asarray(train_df['dst'], dtype=np.int64)
        ts = np.asarray(train_df['time'], dtype=np.float32)
        for _ in range(rounds):
            for lo in range(0, len(src), batch_size):    # get_batch_no_neg (utils.py:398-410)
                hi = lo + batch_size
                roots = np.concatenate([src[lo:hi], dst[lo:hi]])
                t = np.concatenate([ts[lo:hi], ts[lo:hi]])
                mfgs = sampler.sample(roots, t)
                if self._node is not None:
                    for b in mfgs[0]:
                        # the reference's `count[ids] += 1` adds 1 per DISTINCT id per block
                        node_cnt[torch.unique(b.srcdata['ID'])] += 1
                if self._edge is not None:
                    for mfg in mfgs:
                        for b in mfg:
                            if b.num_src_nodes() > b.num_dst_nodes():
                                ids = b.edata['ID']
                                edge_cnt[torch.unique(ids)] += 1
        with torch.cuda.device(dev):
            if self._node is not None:
                self._node.init_ids(self._top(node_cnt, self.node_capacity), self._stream())
            if self._edge is not None:
                self._edge.init_ids(self._top(edge_cnt, self.edge_capacity), self._stream())

    @staticmethod
    def _top(counts: torch.Tensor, k: int) -> torch.Tensor:
        order = torch.argsort(counts.to(torch.int64), descending=True, stable=True)
        return order[:k].contiguous()

    def reset(self):
        """Nothing to reset: the cache is static."""

    def fetch_feature(self, mfgs, eid=None, update_cache=True, target_edge_features=True,
                      **kwargs):
        # gnnlab_static_cache.py:170-182: never updates
        return super(GNNLabStaticCache, self).fetch_feature(
            mfgs, eid=eid, update_cache=False, target_edge_features=target_edge_features,
            **kwargs)
