from .cache import Cache
from .lru_cache import LRUCache
from .policies import FIFOCache, GNNLabStaticCache, LFUCache

__all__ = ["Cache", "LRUCache", "LFUCache", "FIFOCache", "GNNLabStaticCache"]
