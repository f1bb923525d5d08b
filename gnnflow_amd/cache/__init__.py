from .cache import Cache
from .lru_cache import LRUCache

__all__ = ["Cache", "LRUCache"]
