"""gnnflow.cache.LRUCache on MI355X (gnnflow/cache/lru_cache.py:9-201): the LRU
replacement itself lives in gnnflow_amd/csrc/feature_cache.hip; this class only adds the
name and the edge-only reset of the reference."""
import torch

from .cache import Cache


class LRUCache(Cache):
    """
    Least-recently-used (LRU) cache
    """

    def __init__(self, *args, **kwargs):
        super(LRUCache, self).__init__(*args, **kwargs)
        self.name = 'lru'

    def reset(self):
        """Reset the cache — NB: only the edge cache is reset (lru_cache.py:74-105)."""
        self.wait_enqueued()
        if self._edge is not None:
            with torch.cuda.device(self.device):
                self._edge.init(self._stream())
