"""
cache_oracle.py — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

numpy restatement of the reference's GPU feature cache with LRU replacement:
  Cache.__init__ / init_cache / fetch_feature   gnnflow/cache/cache.py:15-134,157-195,255-413
  LRUCache.update_*_cache / reset               gnnflow/cache/lru_cache.py:74-105,121-201

Pinned against the reference's own Python (imported in the build container with stubbed
dgl / libgnnflow; tests/golden/make_cache_fixtures.py -> tests/golden/cache_reference.npz):
fetched features bit-exact, hit ratios exact wherever the reference's `torch.topk` has no
tie to break.  Two places where the reference is under-specified are made deterministic
here and in the HIP path the same way:
  * eviction ties (equal `count`; torch.topk's tie order is unspecified and differs between
    its CPU and CUDA kernels).  LRU: STABLE — slots with equal `count` keep the relative
    order they had before (slot order after init_cache()), i.e. the cache is an LRU list:
    an update moves the hit slots, in list order, behind the others, evicts the first k of
    that list and appends them, refilled, in eviction order (the i-th victim receives the
    i-th new id) — what a stable sort by `count` yields.  LFU: lowest slot index first,
    evicted slots refilled in slot order;
  * overflow_rule: when a block has more distinct missed ids than the cache has slots,
    the reference keeps the `capacity` smallest ids (an artefact of torch.unique
    sorting); "first_seen" keeps the first `capacity` distinct missed ids in block order,
    which is what the HIP path does.  "smallest_ids" reproduces the reference.
"""
import numpy as np


class LRUKind:
    """One cache kind (node or edge): lru_cache.py state + one block of fetch_feature."""

    def __init__(self, num_ids, capacity, feats, overflow_rule="first_seen", policy="lru"):
        self.policy = policy          # "lru" | "lfu" | "fifo"
        self.updates = 0              # update_*_cache calls since init (LRU: -count of an untouched slot)
        self.pointer = int(capacity) - 1   # fifo_cache.py:66-69
        self.num_ids, self.capacity = int(num_ids), int(capacity)
        self.feats = np.ascontiguousarray(feats, np.float32)
        self.dim = self.feats.shape[1]
        self.overflow_rule = overflow_rule
        self.buffer = np.zeros((self.capacity, self.dim), np.float32)
        self.flag = np.zeros(self.num_ids, bool)
        self.map = np.full(self.num_ids, -1, np.int64)
        self.index_to_id = np.full(self.capacity, -1, np.int64)
        self.count = np.zeros(self.capacity, np.int32)
        # LRU list: slots, least recently refreshed first (primary order == `count`)
        self.order = np.arange(self.capacity, dtype=np.int64)

    # cache.py:175-195 / lru_cache.py:91-105
    def init(self):
        c = self.capacity
        self.flag[:] = False
        self.map[:] = -1
        ids = np.arange(c)
        self.buffer[ids] = self.feats[:c]
        self.flag[ids] = True
        self.index_to_id = ids.astype(np.int64)
        self.map[ids] = ids
        self.count[:] = 1 if self.policy == "lfu" else 0     # lfu_cache.py:80-84
        self.updates = 0
        self.pointer = self.capacity - 1
        self.order = np.arange(c, dtype=np.int64)

    def reset_order(self):   # fifo_cache.py:70-75 (pointer) / lfu_cache.py:118 (count.zero_())
        self.pointer = self.capacity - 1
        self.count[:] = 0

    # one block of cache.py:269-323 (nodes) / :326-400 (edges)
    def fetch(self, ids, update=True):
        ids = np.asarray(ids, np.int64)
        n = len(ids)
        mask = self.flag[ids]
        hits = int(mask.sum())
        out = np.zeros((n, self.dim), np.float32)
        cached_index = self.map[ids[mask]]
        out[mask] = self.buffer[cached_index]
        miss_ids = ids[~mask]
        out[~mask] = self.feats[miss_ids]
        if update and len(miss_ids) > 0 and self.capacity > 0:
            if self.overflow_rule == "smallest_ids":
                uniq = np.unique(miss_ids)                      # torch.unique sorts
            else:
                _, first = np.unique(miss_ids, return_index=True)
                uniq = miss_ids[np.sort(first)]                 # block order
            self._update(cached_index, uniq)
        return out, hits, n

    # lru_cache.py:121-160 / lfu_cache.py:134-172 / fifo_cache.py:77-119
    def _update(self, cached_index, uncached_ids):
        k = min(len(uncached_ids), self.capacity)
        ids_to_cache = uncached_ids[:k]
        self.updates += 1
        if self.policy == "fifo":
            C, p = self.capacity, self.pointer
            if p + k < C:
                removing = np.arange(p + 1, p + k + 1)
                self.pointer = p + k
            else:
                removing = np.concatenate([np.arange(k - (C - 1 - p)), np.arange(p + 1, C)])
                self.pointer = k - (C - 1 - p) - 1
            self._install(removing, ids_to_cache, None)
            return
        if self.policy == "lfu":
            self.count[np.unique(cached_index)] += 1     # `count[idx] += 1`: once per slot
            removing = np.sort(np.argsort(self.count, kind="stable")[:k])
            self._install(removing, ids_to_cache, 1)
            return
        self.count -= 1
        self.count[cached_index] = 0
        # topk(k, largest=False): the k least recently refreshed slots, ties resolved
        # stably: the hit slots move behind the others keeping their order, the first k of
        # the list are evicted and re-appended, refilled, in that order.
        hit = np.zeros(self.capacity, bool)
        hit[cached_index] = True
        in_order = hit[self.order]
        lst = np.concatenate([self.order[~in_order], self.order[in_order]])
        removing = lst[:k]
        assert np.array_equal(np.sort(self.count[removing]),
                              np.sort(np.partition(self.count, k - 1)[:k]))   # a valid topk
        self._install(removing, ids_to_cache, 0)
        self.order = np.concatenate([lst[k:], removing])

    def _install(self, removing, ids_to_cache, new_count):
        removing_ids = self.index_to_id[removing]
        self.buffer[removing] = self.feats[ids_to_cache]
        if new_count is not None:
            self.count[removing] = new_count
        live = removing_ids >= 0
        self.flag[removing_ids[live]] = False
        self.flag[ids_to_cache] = True
        self.map[removing_ids[live]] = -1
        self.map[ids_to_cache] = removing
        self.index_to_id[removing] = ids_to_cache

    def cached_ids(self):
        return np.sort(self.index_to_id[self.index_to_id >= 0])

    # Cache.resize / LRUCache.resize (cache.py:197-221, lru_cache.py:107-119).  The reference
    # grows its tensors with torch's resize_, which leaves the new elements UNINITIALISED;
    # the deterministic completion used here and in the HIP path: new ids are uncached, new
    # slots are empty and the first to be refilled, in slot order (LRU), unused (LFU, count
    # 0), or simply further along the ring (FIFO, pointer unchanged).
    def resize(self, num_ids, capacity, feats):
        num_ids, capacity = int(num_ids), max(int(capacity), self.capacity)
        add_ids, add_slots = num_ids - self.num_ids, capacity - self.capacity
        self.feats = np.ascontiguousarray(feats, np.float32)
        self.flag = np.concatenate([self.flag, np.zeros(add_ids, bool)])
        self.map = np.concatenate([self.map, np.full(add_ids, -1, np.int64)])
        self.buffer = np.concatenate([self.buffer, np.zeros((add_slots, self.dim), np.float32)])
        self.index_to_id = np.concatenate([self.index_to_id, np.full(add_slots, -1, np.int64)])
        fill = -self.updates - 1 if self.policy == "lru" else 0
        self.count = np.concatenate([self.count, np.full(add_slots, fill, np.int32)])
        self.order = np.concatenate([np.arange(self.capacity, capacity, dtype=np.int64),
                                     self.order])
        self.num_ids, self.capacity = num_ids, capacity


class OracleLRUCache:
    """Cache + LRUCache protocol over MFG-like blocks with numpy srcdata/edata."""

    def __init__(self, edge_cache_ratio, node_cache_ratio, num_nodes, num_edges,
                 node_feats=None, edge_feats=None, dim_node_feat=0, dim_edge_feat=0,
                 overflow_rule="first_seen", policy="lru"):
        self.node = self.edge = None
        self.node_cache_ratio, self.edge_cache_ratio = node_cache_ratio, edge_cache_ratio
        self.num_nodes, self.num_edges = int(num_nodes), int(num_edges)
        self.node_capacity = int(node_cache_ratio * num_nodes)   # cache.py:82
        self.edge_capacity = int(edge_cache_ratio * num_edges)   # cache.py:83
        if dim_node_feat:
            self.node = LRUKind(num_nodes, self.node_capacity, node_feats, overflow_rule, policy)
        if dim_edge_feat:
            self.edge = LRUKind(num_edges, self.edge_capacity, edge_feats, overflow_rule, policy)
        self.policy = policy
        self.cache_node_ratio = 0
        self.cache_edge_ratio = 0
        self.target_edge_features = None

    def init_cache(self):
        if self.node:
            self.node.init()
        if self.edge:
            self.edge.init()

    def resize(self, new_num_nodes, new_num_edges, node_feats=None, edge_feats=None):
        if self.node and new_num_nodes > self.num_nodes:
            self.num_nodes = new_num_nodes
            self.node_capacity = max(int(self.node_cache_ratio * new_num_nodes), self.node_capacity)
            self.node.resize(new_num_nodes, self.node_capacity, node_feats)
        if self.edge and new_num_edges > self.num_edges:
            self.num_edges = new_num_edges
            self.edge_capacity = max(int(self.edge_cache_ratio * new_num_edges), self.edge_capacity)
            self.edge.resize(new_num_edges, self.edge_capacity, edge_feats)

    def reset(self):   # lru/lfu: only the edge cache is re-initialised; fifo: pointer rewind
        if self.edge:
            if self.policy != "fifo":
                self.edge.init()
            if self.policy != "lru":
                self.edge.reset_order()

    def fetch_feature(self, mfgs, eid=None, update_cache=True, target_edge_features=True):
        if self.node:
            ratios = []
            for b in mfgs[0]:
                out, hits, n = self.node.fetch(b.srcdata["ID"], update_cache)
                b.srcdata["h"] = out
                ratios.append(np.float32(hits) / np.float32(n))
            self.cache_node_ratio = float(np.mean(ratios)) if ratios else 0
        if self.edge:
            ratios = []
            for mfg in mfgs:
                for b in mfg:
                    ids = b.edata["ID"]
                    if len(ids) == 0:
                        continue
                    out, hits, n = self.edge.fetch(ids, update_cache)
                    b.edata["f"] = out
                    ratios.append(np.float32(hits) / np.float32(n))
            self.cache_edge_ratio = float(np.mean(ratios)) if ratios else 0
            if target_edge_features and eid is not None:
                self.target_edge_features = self.edge.feats[np.asarray(eid, np.int64)]
        return mfgs
