"""gnnflow.cache.Cache on MI355X — the reference's feature-cache protocol
(gnnflow/cache/cache.py:10-413) over the fused HIP gather (gf_cache_fetch).

Same constructor arguments, attributes read by callers (`cache_node_ratio`,
`cache_edge_ratio`, `target_edge_features`, `name`) and methods (`init_cache`, `reset`,
`fetch_feature`, `get_mem_size`, `resize`).  Differences, all internal:
  * the per-block torch op chains + host round trip for misses are one kernel per block;
  * the feature tables must be device-readable: they are placed in HBM when they fit
    (`feature_placement="device"`, the default on a 288 GB MI355X) or kept in pinned
    host memory that the gather kernel reads directly over PCIe (`"pinned"`);
  * hit ratios are accumulated on the device and only synchronised when read.
"""
import ctypes as C
from collections import deque
import os
import struct
from typing import List, Optional, Union

import numpy as np
import torch

from .. import _capi

_DESC = struct.Struct("iiQQQQ")     # struct gf_fetch_desc: kind, update, d_ids, n, d_out, d_stats
_current_device = getattr(torch._C, "_cuda_getDevice", None) or torch.cuda.current_device

# asynchronous fetch_feature() submissions that may be with the enqueue thread at a time
_MAX_QUEUED = max(1, int(os.environ.get("GNNFLOW_FETCH_QUEUED", "4")))
assert _DESC.size == C.sizeof(_capi.GfFetchDesc)


class _Kind:
    """One cache kind (node or edge): C handle + the device-readable feature table."""

    def __init__(self, lib, num_ids, capacity, feats: torch.Tensor, dim, device, placement,
                 policy="lru", row_mirror=True):
        self.lib = lib
        self.num_ids, self.capacity, self.dim = int(num_ids), int(capacity), int(dim)
        self.device = device
        feats = feats.detach()
        if feats.dtype != torch.float32:
            feats = feats.to(torch.float32)       # cache.py:71-74 (bool -> float32)
        assert feats.dim() == 2 and feats.shape[1] == self.dim, \
            "feature table must be [num_ids, dim]"
        self.local_rows = int(feats.shape[0])
        if placement == "pinned":
            feats = feats.cpu().contiguous()
            self.table = feats if feats.is_pinned() else feats.pin_memory()
        else:
            self.table = feats.to(device).contiguous()
        self.h = C.c_void_p()
        _capi.check(lib.gf_cache_create(C.byref(self.h), self.num_ids, self.capacity,
                                        self.dim, self.table.data_ptr(), device.index))
        _capi.check(lib.gf_cache_set_policy(self.h, _capi.CACHE_POLICY[policy]))
        if not row_mirror and placement != "pinned":
            # the table itself is in HBM: no second copy of the cached rows (include/gnnflow_hip.h
            # gf_cache_set_row_mirror) — GNNFLOW_CACHE_ROW_MIRROR=1 keeps it (A/B, tests)
            _capi.check(lib.gf_cache_set_row_mirror(self.h, 0))

    def close(self):
        if self.h is not None and self.h.value:
            self.lib.gf_cache_destroy(self.h)
            self.h = None

    def init(self, stream):
        _capi.check(self.lib.gf_cache_init(self.h, stream))

    def init_ids(self, ids: torch.Tensor, stream):
        _capi.check(self.lib.gf_cache_init_ids(self.h, ids.data_ptr(), int(ids.shape[0]), stream))

    def mem_bytes(self) -> int:
        n = C.c_size_t(0)
        _capi.check(self.lib.gf_cache_mem_bytes(self.h, C.byref(n)))
        return n.value

    def lru_state(self) -> dict:
        """gf_cache_lru_state: form of the LRU order and its counters (diagnostics / tests)."""
        out = (C.c_uint64 * 7)()
        _capi.check(self.lib.gf_cache_lru_state(self.h, out))
        keys = ("queue_form", "queue_entries", "head", "tail", "compactions",
                "list_form_updates", "lone_walks")
        return dict(zip(keys, (int(v) for v in out)))

    def slot_ids(self) -> np.ndarray:
        out = np.zeros(self.capacity, np.int64)
        if self.capacity:
            _capi.check(self.lib.gf_cache_slot_ids(self.h, out.ctypes.data, self.capacity))
        return out


class Cache:
    """
    Feature cache on GPU
    """
    _policy = "lru"   # replacement policy of the native cache (subclasses override)

    def __init__(self, edge_cache_ratio: int, node_cache_ratio: int,
                 num_nodes: int, num_edges: int,
                 device: Union[str, torch.device],
                 node_feats: Optional[torch.Tensor] = None,
                 edge_feats: Optional[torch.Tensor] = None,
                 dim_node_feat: Optional[int] = 0,
                 dim_edge_feat: Optional[int] = 0,
                 pinned_nfeat_buffs: Optional[torch.Tensor] = None,
                 pinned_efeat_buffs: Optional[torch.Tensor] = None,
                 kvstore_client=None,
                 distributed: Optional[bool] = False,
                 neg_sample_ratio: Optional[int] = 1,
                 feature_placement: Optional[str] = None,
                 staging=None):
        if device == 'cpu' or device == torch.device('cpu'):
            raise ValueError('Cache must be on GPU')
        shards = None
        if distributed:
            # the feature tables are sharded over the GPUs by owner (gnnflow_amd.dist.
            # ShardedFeatures as `kvstore_client`); misses are pulled with all-to-all-v
            shards = kvstore_client
            if shards is None or not (hasattr(shards, "node") and hasattr(shards, "edge")):
                raise ValueError('distributed=True needs kvstore_client=gnnflow_amd.dist.'
                                 'ShardedFeatures(node=..., edge=...)')
            if shards.node is None and shards.edge is None:
                raise ValueError('At least one of node and edge shards must be provided')
            node_feats = edge_feats = None
        elif node_feats is None and edge_feats is None:
            raise ValueError('At least one of node_feats and edge_feats must be provided')
        if node_feats is not None and node_feats.shape[0] != num_nodes:
            raise ValueError(
                'The number of nodes in node_feats {} does not match num_nodes {}'.format(
                    node_feats.shape[0], num_nodes))
        if edge_feats is not None and edge_feats.shape[0] != num_edges:
            raise ValueError(
                'The number of edges in edge_feats {} does not match num_edges {}'.format(
                    edge_feats.shape[0], num_edges))
        assert 0 <= edge_cache_ratio <= 1, 'edge_cache_ratio must be in [0, 1]'
        assert 0 <= node_cache_ratio <= 1, 'node_cache_ratio must be in [0, 1]'

        device = torch.device(device)
        if device.type != 'cuda':
            raise ValueError('Cache must be on GPU')
        if device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        placement = (feature_placement or
                     os.environ.get('GNNFLOW_FEATURE_PLACEMENT', 'device')).lower()
        if placement not in ('device', 'pinned'):
            raise ValueError("feature_placement must be 'device' or 'pinned'")

        self.edge_cache_ratio = edge_cache_ratio
        self.node_cache_ratio = node_cache_ratio
        self.node_capacity = int(node_cache_ratio * num_nodes)   # cache.py:82
        self.edge_capacity = int(edge_cache_ratio * num_edges)   # cache.py:83
        self.num_nodes, self.num_edges = num_nodes, num_edges
        self.node_feats, self.edge_feats = node_feats, edge_feats
        self.dim_node_feat = dim_node_feat if node_feats is not None else 0
        self.dim_edge_feat = dim_edge_feat if edge_feats is not None else 0
        if shards is not None:
            self.dim_node_feat = shards.node.dim if shards.node is not None else 0
            self.dim_edge_feat = shards.edge.dim if shards.edge is not None else 0
        self.device = device
        self.pinned_nfeat_buffs = pinned_nfeat_buffs   # accepted, unused: misses never
        self.pinned_efeat_buffs = pinned_efeat_buffs   # stage through the host here
        self.kvstore_client = kvstore_client
        self.distributed = bool(distributed)
        self._shards = shards
        self.neg_sample_ratio = neg_sample_ratio
        self._target_edge_features = None
        self._tickets = deque()    # (ticket, keepalive) of asynchronous fetches not yet enqueued
        self.feature_placement = placement
        # Staging ring of a host-resident table (feature_placement="pinned"; include/gnnflow_hip.h
        # gf_cache_set_staging): prefetch_feature() pulls the rows a coming fetch_feature() will
        # miss into HBM on a side stream.  `staging`: None = GNNFLOW_STAGING or "auto" (32
        # generations, rows per generation sized from the blocks prefetched), 0 / False = off,
        # (generations, rows_per_generation) = fixed.
        if staging is None:
            staging = os.environ.get('GNNFLOW_STAGING', 'auto')
        if isinstance(staging, str):
            staging = staging.strip().lower()
            if staging in ('0', 'off', 'false', ''):
                staging = None
            elif staging != 'auto':
                g, _, r = staging.partition(',')
                staging = (int(g), int(r) if r else 0)
        elif not staging:
            staging = None
        elif staging is True:
            staging = 'auto'
        if placement != 'pinned' or distributed:
            staging = None
        self._staging = staging          # None | 'auto' | (generations, rows; 0 rows: auto)
        self._staging_rows = 0           # rows per generation the native rings were set up with
        self._prefetch_stream = None
        self._prefetch_handle = None
        # serve an edge block that is a prefix of the previously fetched one from that
        # block's rows (LRU only; fetch_feature); GNNFLOW_PREFIX_ALIAS=0 turns it off
        self.prefix_alias = os.environ.get('GNNFLOW_PREFIX_ALIAS', '1') != '0'

        self._lib = _capi.load()
        self._node = self._edge = None
        with torch.cuda.device(device):
            if shards is not None:
                # no local table: a missed row always comes out of the pulled rows
                def stub(dim):
                    return torch.zeros((1, dim), dtype=torch.float32)
                if self.dim_node_feat != 0:
                    self._node = _Kind(self._lib, num_nodes, self.node_capacity,
                                       stub(self.dim_node_feat), self.dim_node_feat, device,
                                       "device", self._policy)
                if self.dim_edge_feat != 0:
                    self._edge = _Kind(self._lib, num_edges, self.edge_capacity,
                                       stub(self.dim_edge_feat), self.dim_edge_feat, device,
                                       "device", self._policy)
            else:
                mirror = placement == "pinned" or \
                    os.environ.get("GNNFLOW_CACHE_ROW_MIRROR", "0") == "1"
                if self.dim_node_feat != 0:
                    self._node = _Kind(self._lib, num_nodes, self.node_capacity, node_feats,
                                       self.dim_node_feat, device, placement, self._policy, mirror)
                if self.dim_edge_feat != 0:
                    self._edge = _Kind(self._lib, num_edges, self.edge_capacity, edge_feats,
                                       self.dim_edge_feat, device, placement, self._policy, mirror)
        self._stats_span = None
        self._target_edge_thunk = None
        self.num_gather_launches = 0   # gather launches issued so far (one per round)
        # rows x (8 B id + 2 x 4 x dim B) over every row the gather launches moved so far
        # (SURVEY.md 8(d) algorithmic bytes; read by bench.py's roofline)
        self.algorithmic_bytes = 0
        self.rows_moved = 0            # rows those launches gathered (aliased blocks move none)

    def __del__(self):
        sess = self.__dict__.get("_pull_session")
        if sess is not None and sess[0].value:       # before the caches it fetches into
            self._lib.gf_pull_session_destroy(sess[0])
            self._pull_session = None
        for k in (getattr(self, "_node", None), getattr(self, "_edge", None)):
            if k is not None:
                k.close()

    # ---- hit ratios: mean over blocks of hits/len (cache.py:277,323,337,400) ---------
    @staticmethod
    def _ratio(stats, all_hit_blocks=0):
        # stats[b] = 16 x int32: hits in the even words (8 shards), n in word 1;
        # all_hit_blocks: blocks served as a prefix of another block's rows (ratio 1 each)
        if stats is None or stats.shape[0] == 0:
            return 1.0 if all_hit_blocks else 0
        s = stats.to(torch.float32)
        s = s[s[:, 1] > 0]               # a block without ids fetched nothing (reference: skipped)
        if s.shape[0] == 0:
            return 1.0 if all_hit_blocks else 0
        r = s[:, 0::2].sum(dim=1) / s[:, 1]
        if all_hit_blocks:
            return (r.sum() + all_hit_blocks) / (r.shape[0] + all_hit_blocks)
        return r.mean()

    @property
    def cache_node_ratio(self):
        if self._node is None or self._stats_span is None:
            return 0
        pos, n_node, n_cached, ring, _ = self._stats_span
        return self._ratio(ring[pos:pos + n_node])

    @property
    def cache_edge_ratio(self):
        if self._edge is None or self._stats_span is None:
            return 0
        pos, n_node, n_cached, ring, n_alias = self._stats_span
        return self._ratio(ring[pos + n_node:pos + n_cached], n_alias)

    def _stream(self):
        return _capi.current_stream(self.device)

    def slot_ids(self, kind: str) -> np.ndarray:
        """ids cached per slot (-1: empty) of the 'node' or 'edge' cache, on the host
        (diagnostics / tests; synchronises)."""
        self.wait_enqueued()
        k = self._node if kind == "node" else self._edge
        return k.slot_ids()

    def get_mem_size(self) -> int:
        """Memory size of the cache in bytes (cache.py:136-155)."""
        return sum(k.mem_bytes() for k in (self._node, self._edge) if k is not None)

    def init_cache(self, *args, **kwargs):
        """Fill the cache with the first `capacity` rows (cache.py:157-195)."""
        self.wait_enqueued()      # an asynchronous fetch may still be with the enqueue thread
        if self.distributed:
            return self._init_cache_distributed()
        with torch.cuda.device(self.device):
            for k in (self._node, self._edge):
                if k is not None:
                    k.init(self._stream())

    def resize(self, new_num_nodes: int, new_num_edges: int):
        """Grow the id spaces (cache.py:197-221).  The caller must have replaced
        `node_feats` / `edge_feats` with tables covering the new ids."""
        self.wait_enqueued()
        if self.distributed:
            raise NotImplementedError("resize of a cache over sharded feature tables")
        with torch.cuda.device(self.device):
            if self._node is not None and new_num_nodes > self.num_nodes:
                self._regrow("_node", self.node_feats, new_num_nodes,
                             int(self.node_cache_ratio * new_num_nodes), self.dim_node_feat)
                self.num_nodes = new_num_nodes
                self.node_capacity = self._node.capacity
            if self._edge is not None and new_num_edges > self.num_edges:
                self._regrow("_edge", self.edge_feats, new_num_edges,
                             int(self.edge_cache_ratio * new_num_edges), self.dim_edge_feat)
                self.num_edges = new_num_edges
                self.edge_capacity = self._edge.capacity

    def _regrow(self, attr, feats, num_ids, capacity, dim):
        kind = getattr(self, attr)
        if feats is None or feats.shape[0] < num_ids:
            raise ValueError("resize: feature table does not cover the new ids")
        f = feats.detach()
        if f.dtype != torch.float32:
            f = f.to(torch.float32)
        if self.feature_placement == "pinned":
            f = f.cpu().contiguous()
            table = f if f.is_pinned() else f.pin_memory()
        else:
            table = f.to(self.device).contiguous()
        _capi.check(self._lib.gf_cache_resize(kind.h, num_ids, max(capacity, kind.capacity),
                                              table.data_ptr(), self._stream()))
        kind.table = table
        kind.num_ids, kind.capacity = num_ids, max(capacity, kind.capacity)

    def reset(self):
        raise NotImplementedError

    def _ids(self, t) -> torch.Tensor:
        if not isinstance(t, torch.Tensor):
            t = torch.from_numpy(np.ascontiguousarray(t, dtype=np.int64))
        if t.device != self.device or t.dtype != torch.int64 or not t.is_contiguous():
            t = t.to(self.device, torch.int64).contiguous()
        return t

    def _stats_rows(self, n):
        """Position of n zeroed 16-word stats records in a ring that is replaced by a fresh
        zeroed one when it is used up (a torch.zeros per call is one more kernel launch on
        the critical path)."""
        n = max(n, 1)
        ring = getattr(self, "_stats_ring", None)
        if ring is None or self._stats_pos + n > ring.shape[0]:
            self._stats_ring = ring = torch.zeros((max(1024, 4 * n), 16), dtype=torch.int32,
                                                  device=self.device)
            self._stats_pos = 0
        pos = self._stats_pos
        self._stats_pos += n
        return pos

    def wait_enqueued(self, upto=None):
        """Blocks until the asynchronous fetch_feature() calls so far (`upto`: up to that
        ticket) have been enqueued on their stream (no-op otherwise).  Stream order then
        guarantees the results.  Submissions are enqueued in ticket order."""
        q = self.__dict__.get("_tickets")
        while q and (upto is None or q[0][0] <= upto):
            ticket, _refs = q.popleft()
            _capi.check(self._lib.gf_cache_fetch_wait(ticket))

    @property
    def target_edge_features(self):
        if self._target_edge_features is None and self._target_edge_thunk is not None:
            self._target_edge_features = self._target_edge_thunk()
            self._target_edge_thunk = None
        return self._target_edge_features

    @target_edge_features.setter
    def target_edge_features(self, value):
        self._target_edge_thunk = None
        self._target_edge_features = value

    def fetch_feature(self, mfgs: List[List], eid: Optional[np.ndarray] = None,
                      update_cache: bool = True, target_edge_features: bool = True,
                      async_enqueue: bool = False, announce=None):
        """Fetching the node/edge features of input_node_ids (cache.py:255-413):
        node features for the blocks of mfgs[0] -> srcdata['h'], edge features for every
        block -> edata['f'], target edge features for TGN memory.  One native call
        (gf_cache_fetch_blocks) issues every gather + LRU update; node and edge caches
        proceed concurrently on the device.

        async_enqueue=True returns as soon as the work has been handed to the library's
        enqueue thread; `b.srcdata['h']` / `b.edata['f']` / `target_edge_features` then wait
        for the enqueue on first access (blocks built by gnnflow_amd.TemporalSampler), or
        call wait_enqueued().

        announce=(later_mfgs, later_eid): with async_enqueue, a later batch's
        prefetch_feature() rides in the same submission (one hand-over to the enqueue thread
        per pipelined step instead of two)."""
        # at most _MAX_QUEUED submissions with the enqueue thread: when that thread also issues
        # the partitioned sampler's chains (tens of microseconds each, up to four batches' worth
        # in one job), the previous fetches may still be queued behind one, and waiting for them
        # here would stall the caller
        q = self.__dict__.get("_tickets")
        if q and (len(q) >= _MAX_QUEUED or self.distributed):
            self.wait_enqueued(None if self.distributed else q[0][0])
        if self.distributed:
            return self._fetch_distributed(mfgs, eid, update_cache, target_edge_features)
        upd = 1 if update_cache else 0
        dev = self.device
        jobs, n_node, n_cached, aliases = self._jobs(mfgs, eid, upd, target_edge_features)
        if not jobs:
            if announce is not None:
                self.prefetch_feature(announce[0], announce[1], update_cache, target_edge_features,
                                      async_enqueue=async_enqueue)
            return mfgs
        ann = None
        if announce is not None and self._staging is not None:
            if async_enqueue:
                ann = self._announce_descs(announce, upd, target_edge_features)
            else:
                self.prefetch_feature(announce[0], announce[1], update_cache, target_edge_features)
        if _current_device() != dev.index:
            with torch.cuda.device(dev):
                return self._submit(mfgs, jobs, n_node, n_cached, upd, async_enqueue, aliases, ann)
        return self._submit(mfgs, jobs, n_node, n_cached, upd, async_enqueue, aliases, ann)

    def _announce_descs(self, announce, upd, target_edge_features):
        """A later batch's prefetch for a combined submission: (descriptor reference or None,
        count, gf_block array or None, layers, snapshots, stream handle, keepalive) or None."""
        mfgs, eid = announce[0], announce[1]
        st = self._prefetch_handle
        if st is None:
            from ..pipeline import side_stream
            self._prefetch_stream = side_stream(self.device, int(os.environ.get('GNNFLOW_PREFETCH_STREAM_K', '3')))
            st = self._prefetch_handle = C.c_void_p(self._prefetch_stream.cuda_stream)
        # (no record_stream: the blocks stay alive until their own fetch, which is issued after
        # the pull that reads their ids has been)
        b0 = mfgs[0][0]
        sb = getattr(b0, "_sample_blocks", None)
        want_target = self._edge is not None and target_edge_features and eid is not None
        if sb is not None:
            # the sampler's own block array (+ the target ids): nothing to assemble on this side
            rows = b0._num_src + b0._num_edges
            if not want_target:
                self._ensure_staging(rows)
                return None, 0, sb[0], sb[1], sb[2], st, mfgs
            t = self._ids(eid)
            n = int(t.shape[0])
            self._ensure_staging(rows + n)
            descs, cdescs, cdescs_ref = self._desc_buf(1)
            _DESC.pack_into(descs, 0, 2, upd, t.data_ptr(), n, 0, 0)
            return cdescs_ref, 1, sb[0], sb[1], sb[2], st, (mfgs, t, descs, cdescs)
        jobs = self._jobs(mfgs, eid, upd, target_edge_features)[0]
        if not jobs:
            return None
        self._ensure_staging(sum(job[2] for job in jobs))
        descs, cdescs, cdescs_ref = self._desc_buf(len(jobs))
        pack = _DESC.pack_into
        for i, job in enumerate(jobs):
            pack(descs, i * _DESC.size, job[0], upd, job[1] or 0, job[2], 0, 0)
        return cdescs_ref, len(jobs), None, 0, 0, st, (jobs, descs, cdescs, mfgs)

    def _jobs(self, mfgs, eid, upd, target_edge_features):
        """The block gathers of one fetch_feature() call: (jobs, #node jobs, #jobs through a
        cache, aliases); job = (kind, ids address, n, keepalive, dim, block, which, key)."""
        jobs = []
        if self._node is not None:
            dim = self.dim_node_feat
            for b in mfgs[0]:
                jobs.append((0,) + self._id_array(b, "src") + (dim, b, "src", "h"))
        n_node = len(jobs)
        aliases = []      # (block, index of the job whose output rows start with this block's, n)
        if self._edge is not None:
            dim = self.dim_edge_feat
            # Edge blocks are fetched in the reference's order (cache.py:329-331).  A block
            # whose edge ids are a prefix of the block fetched just before it (see
            # TemporalSampler._finish) needs no second lookup under LRU: after the outer
            # block's update every id of that block is cached as long as the block has no
            # more rows than the cache has slots (hit slots carry the newest stamp, so the
            # k = #distinct misses <= capacity - #hit slots victims are all other slots and
            # every miss is installed) -> the inner block is all hits, which changes no
            # replacement state (cache.py:318-321,395-398) and fetches rows equal to the
            # first rows of the outer block's output.  Its rows alias that output and its
            # hit ratio is 1.  DESIGN.md 3.4c.
            can_alias = (upd and self._policy == "lru" and self.prefix_alias
                         and all(len(mfg) == 1 for mfg in mfgs))
            prev = None   # (block, owner job index) of the previous edge block
            for mfg in mfgs:
                for b in mfg:
                    job = (1,) + self._id_array(b, "e") + (dim, b, "e", "f")
                    n = job[2]
                    if n <= 0:
                        prev = None
                        continue
                    # (no edge cache at all — capacity 0: nothing to keep consistent, the rows
                    # are the outer block's first rows whatever its size)
                    if can_alias and prev is not None and job[3] is b \
                            and getattr(b, "_edge_prefix_of", None) is prev[0] \
                            and n <= jobs[prev[1]][2] \
                            and (jobs[prev[1]][2] <= self.edge_capacity or not self.edge_capacity):
                        aliases.append((b, prev[1], n))
                        prev = (b, prev[1])
                        continue
                    jobs.append(job)
                    prev = (b, len(jobs) - 1) if job[3] is b else None
        n_cached = len(jobs)
        if self._edge is not None and target_edge_features and eid is not None:
            t = self._ids(eid)
            jobs.append((2, t.data_ptr(), int(t.shape[0]), t, self.dim_edge_feat, None, None, None))
        return jobs, n_node, n_cached, aliases

    # ---- host-resident tables: pulling the coming misses ahead of the fetch ------------------
    @property
    def staging(self) -> bool:
        """True when prefetch_feature() does something (pinned tables with a staging ring)."""
        return self._staging is not None

    def _ensure_staging(self, rows: int):
        """Sets the native rings up (or enlarges them: a synchronising call) so that a
        generation holds an eighth of the rows of the round being prefetched — a round of uniform
        sampling over a large graph misses that many (GDELT-shaped, LRU 0.2: hit ratio 0.66, every
        ninth row a distinct miss); 32 generations: the ring takes four times the rows of one
        round's blocks (the reference's pinned staging buffers, utils.py get_pinned_buffers, take
        once that)."""
        st = self._staging
        gens, fixed = (32, 0) if st == 'auto' else (int(st[0]), int(st[1]))
        want = fixed or (1 << max(int(max(rows, 1) // 8 - 1).bit_length(), 10))
        if want <= self._staging_rows:
            return
        self.wait_enqueued()
        for k in (self._node, self._edge):
            if k is not None:
                _capi.check(self._lib.gf_cache_set_staging(k.h, gens, want))
        self._staging_rows = want

    def prefetch_feature(self, mfgs: List[List], eid: Optional[np.ndarray] = None,
                         update_cache: bool = True, target_edge_features: bool = True,
                         stream: Optional[torch.cuda.Stream] = None,
                         async_enqueue: bool = False) -> bool:
        """Announces a coming `fetch_feature(mfgs, eid, ...)` (same arguments): with the tables
        in pinned host memory, the rows of the ids that are neither cached nor staged yet are
        pulled into the staging ring in HBM on `stream` (default: a side stream of this cache),
        beside whatever the fetch stream is doing; the fetch then waits for the pull on its own
        stream and reads those rows from HBM.  The reference does the host -> pinned -> device
        trip inside fetch_feature (cache.py:288-313,381-388).  A hint: results, hit ratios and
        the cache's contents do not depend on it.  Returns False if there is nothing to do."""
        if self._staging is None:
            return False
        upd = 1 if update_cache else 0
        jobs, _n_node, _n_cached, _aliases = self._jobs(mfgs, eid, upd, target_edge_features)
        if not jobs:
            return False
        dev = self.device
        self._ensure_staging(sum(job[2] for job in jobs))
        if stream is None:
            stream = self._prefetch_stream
            if stream is None:
                from ..pipeline import side_stream
                stream = self._prefetch_stream = side_stream(dev, int(os.environ.get('GNNFLOW_PREFETCH_STREAM_K', '3')))
        for mfg in mfgs:
            for b in mfg:
                if hasattr(b, "record_stream"):
                    b.record_stream(stream)
        nj = len(jobs)
        descs, cdescs, cdescs_ref = self._desc_buf(nj)
        pack = _DESC.pack_into
        for i, (kind, ids_ptr, n, _keep, _dim, _b, _which, _key) in enumerate(jobs):
            pack(descs, i * _DESC.size, kind, upd, ids_ptr or 0, n, 0, 0)
        node_h = self._node.h if self._node is not None else None
        edge_h = self._edge.h if self._edge is not None else None
        st = C.c_void_p(stream.cuda_stream)
        with torch.cuda.device(dev):
            if async_enqueue:
                q = self._tickets
                if len(q) >= _MAX_QUEUED:
                    self.wait_enqueued(q[0][0])
                t = C.c_uint64(0)
                _capi.check(self._lib.gf_cache_prefetch_blocks_async(
                    node_h, edge_h, cdescs_ref, nj, st, C.byref(t)))
                q.append((t.value, (jobs, descs, cdescs, mfgs)))
                return True
            issued = C.c_int(0)
            _capi.check(self._lib.gf_cache_prefetch_blocks(
                node_h, edge_h, cdescs_ref, nj, st, C.byref(issued)))
            return bool(issued.value)

    def set_staging_lag(self, lag: int):
        """A loop that announces a batch `lag` + 1 steps before fetching it says so: a fetch then
        depends on (waits for, reads rows of) all but the `lag` newest announcements
        (include/gnnflow_hip.h gf_cache_set_staging_lag)."""
        self.wait_enqueued()
        self._staging_lag = int(lag)
        for k in (self._node, self._edge):
            if k is not None:
                _capi.check(self._lib.gf_cache_set_staging_lag(k.h, int(lag)))

    def staging_state(self) -> dict:
        """Per kind: generations, rows per generation, generations issued / dropped, rows pulled
        over the host link so far, HBM bytes of the ring + index, rows the gathers read from the
        host table after all (synchronises)."""
        self.wait_enqueued()
        out = {}
        keys = ("generations", "rows_per_generation", "issued", "dropped", "rows_pulled",
                "ring_bytes", "rows_read_from_host", "issue_wait_us", "stream_waits")
        for name, k in (("node", self._node), ("edge", self._edge)):
            if k is None:
                continue
            v = (C.c_uint64 * 9)()
            _capi.check(self._lib.gf_cache_staging_state(k.h, v))
            out[name] = dict(zip(keys, (int(x) for x in v)))
        return out

    def invalidate_staging(self):
        """The feature tables' contents changed in place: staged rows are forgotten."""
        self.wait_enqueued()
        for k in (self._node, self._edge):
            if k is not None:
                _capi.check(self._lib.gf_cache_invalidate_staging(k.h))

    # ---- sharded feature tables (distributed=True) ----------------------------------------
    def _init_cache_distributed(self):
        """cache.py:161-173: the edge cache starts with the first `capacity` edges of THIS
        rank's shard (no exchange), the node cache starts empty."""
        with torch.cuda.device(self.device):
            st = self._stream()
            if self._node is not None:
                _capi.check(self._lib.gf_cache_init_rows(self._node.h, None, 0, None, st))
            if self._edge is not None:
                sh = self._shards.edge
                n = min(self.edge_capacity, int(sh.local_ids.shape[0]))
                ids = sh.local_ids[:n].contiguous()
                rows = sh.rows[:n].contiguous()
                _capi.check(self._lib.gf_cache_init_rows(
                    self._edge.h, ids.data_ptr() if n else None, n,
                    rows.data_ptr() if n else None, st))
                self._init_refs = (ids, rows)

    def _fetch_pulled(self, kind, shards, ids, keys, upd, stats_row):
        """One block over a sharded table: probe the cache, pull the distinct missed rows from
        their owners (a collective: called for EVERY block on every rank, with no ids when
        the block is empty), then the usual fused gather + replacement with the pulled rows
        standing in for the local table.  Returns the block's rows."""
        lib, dev, st = self._lib, self.device, self._stream()
        n = int(ids.shape[0])
        slot = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
        if n:
            _capi.check(lib.gf_cache_probe(kind.h, ids.data_ptr(), n, slot.data_ptr(), st))
        miss_pos = torch.nonzero(slot[:n] == -1).flatten()
        miss_ids = ids[miss_pos]
        uniq, inverse = torch.unique(miss_ids, return_inverse=True)
        key_u = torch.empty_like(uniq)
        key_u[inverse] = keys[miss_pos]          # every occurrence of an id has the same key
        rows_u = shards.pull(uniq, key_u)        # collective
        out = torch.empty((n, kind.dim), dtype=torch.float32, device=dev)
        if n:
            if rows_u.shape[0] == 0:             # all hits: the kernel still wants a pointer
                rows_u = torch.zeros((1, kind.dim), dtype=torch.float32, device=dev)
            miss_index = torch.zeros(n, dtype=torch.int32, device=dev)
            miss_index[miss_pos] = inverse.to(torch.int32)
            _capi.check(lib.gf_cache_fetch_pulled(
                kind.h, ids.data_ptr(), n, out.data_ptr(), upd, stats_row, rows_u.data_ptr(),
                miss_index.data_ptr(), st))
            # rows_u / miss_index are read by kernels already queued on this stream; the
            # allocator reuses their memory only behind them on the same stream
        return out

    def _fetch_distributed(self, mfgs, eid, update_cache, target_edge_features):
        """fetch_feature over sharded tables (cache.py:288-313,351-388,403-411): same rows, same
        per-block cache semantics.  Planned natively (include/gnnflow_hip.h gf_pull_*): per
        fetch ROUND — the i-th node block, the i-th edge block and, in the first round, the
        target rows — claims + per-owner counts, ONE count exchange + read-back (the round's
        only host synchronisation), ids out, rows served by their owners and back, then the
        usual gather + replacement with the pulled rows standing in for the local table.  No
        torch `nonzero / unique / argsort / bincount`.  Every rank runs the same rounds: a
        block that is a prefix of the one fetched before it (LRU, most-recent, equal fanouts)
        is served from that block's rows unless ANY rank cannot do so — the ranks tell each
        other in the count exchange."""
        if not update_cache:
            return self._fetch_distributed_torch(mfgs, eid, update_cache, target_edge_features)
        dev, sh = self.device, self._shards
        with torch.cuda.device(dev):
            node_blocks = list(mfgs[0]) if self._node is not None else []
            edge_blocks = [b for mfg in mfgs for b in mfg] if self._edge is not None else []
            n_node, n_edge = len(node_blocks), len(edge_blocks)
            pos = self._stats_rows(n_node + n_edge)
            ring = self._stats_ring
            target = None
            if self._edge is not None and target_edge_features and eid is not None:
                t = self._ids(eid)
                # cache.py:403-409: the batch's source nodes are the first roots
                num = mfgs[-1][0].num_dst_nodes() // (self.neg_sample_ratio + 2)
                nid = self._ids(mfgs[-1][0].srcdata['ID'])[:num]
                if nid.shape[0] != t.shape[0]:
                    raise ValueError("target edge ids and the batch's source roots differ "
                                     "in length")
                target = (t, nid)

            def node_ctx(i):
                b = node_blocks[i]
                return dict(kind=0, cache=self._node, shard=sh.node, ids=self._ids(b.srcdata['ID']),
                            keys=None, key_index=None, block=b,
                            stats=ring.data_ptr() + 64 * (pos + i))

            def edge_ctx(j):
                b = edge_blocks[j]
                ids = self._ids(b.edata['ID'])
                n = int(ids.shape[0])
                # owner of an edge's features: its source node = the block's root of that edge
                return dict(kind=1, cache=self._edge, shard=sh.edge, ids=ids,
                            keys=self._ids(b.srcdata['ID']) if n else None,
                            key_index=b.edges()[1] if n else None, block=b,
                            stats=ring.data_ptr() + 64 * (pos + n_node + j))

            # Edge blocks after the first may be served from the first one's rows
            # if they form a prefix chain and the LRU argument holds — on EVERY rank: this
            # rank's verdict travels in the first round's count exchange.
            structural = (self._policy == "lru" and self.prefix_alias and n_edge > 1
                          and all(len(mfg) == 1 for mfg in mfgs))
            chain = structural and all(
                edge_blocks[j].num_edges() == 0      # (a rank without roots: nothing to serve)
                or (getattr(edge_blocks[j], "_edge_prefix_of", None) is edge_blocks[j - 1]
                    and edge_blocks[j].num_edges() <= edge_blocks[j - 1].num_edges()
                    <= self.edge_capacity)
                for j in range(1, n_edge))
            n_alias, ei, rnd = 0, 0, 0
            while rnd < n_node or ei < n_edge or (rnd == 0 and target is not None):
                ctxs = []
                if rnd < n_node:
                    ctxs.append(node_ctx(rnd))
                if ei < n_edge:
                    ctxs.append(edge_ctx(ei))
                if rnd == 0 and target is not None:
                    ctxs.append(dict(kind=2, cache=None, shard=sh.edge, ids=target[0],
                                     keys=target[1], key_index=None, block=None, stats=0))
                first_edge = ei == 0 and ei < n_edge
                outs, all_fit = self._pull_round(
                    ctxs, 1, flag=0 if (chain or not (first_edge and structural)) else 1)
                for c, out in zip(ctxs, outs):
                    if c["kind"] == 0:
                        c["block"].srcdata['h'] = out
                    elif c["kind"] == 1:
                        if c["n"]:
                            c["block"].edata['f'] = out
                        ei += 1
                        if first_edge and structural and all_fit:
                            for b in edge_blocks[1:]:       # the whole chain, on every rank
                                n = b.num_edges()
                                if n:
                                    b.edata['f'] = out[:n]
                                n_alias += 1
                            ei = n_edge
                    else:
                        self._target_edge_thunk = None
                        self._target_edge_features = out
                rnd += 1
            self._stats_span = (pos, n_node, n_node + n_edge, ring, n_alias)
        return mfgs

    def _arena(self, name, numel, dtype):
        """Grow-only device scratch of the pull rounds, by name (a torch.empty per buffer per
        round costs more host time than the round's kernels run)."""
        pool = self.__dict__.setdefault("_arena_pool", {})
        t = pool.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = pool[name] = torch.empty(max(int(numel * 1.25), 64), dtype=dtype, device=self.device)
        return t

    def _pull_round_native(self, ctxs, upd, flag, comm):
        """The round as ONE native call (gf_pull_round): the library's communicator carries the
        exchanges (`comm` None: one rank, nothing travels)."""
        lib, dev, sh = self._lib, self.device, self._shards
        nctx = len(ctxs)
        sess = self.__dict__.get("_pull_session")
        if sess is None or sess[1] is not comm:
            h = C.c_void_p()
            _capi.check(lib.gf_pull_session_create(C.byref(h), comm.h if comm is not None else None,
                                                   dev.index))
            sess = self._pull_session = (h, comm)
        arr = (_capi.GfPullCtx * nctx)()
        outs = []
        for k, c in enumerate(ctxs):
            n = int(c["ids"].shape[0])
            c["n"] = n
            shard = c["shard"]
            out = torch.empty((n, shard.dim), dtype=torch.float32, device=dev)
            outs.append(out)
            a = arr[k]
            a.pull.d_ids = c["ids"].data_ptr() if n else None
            a.pull.n = n
            a.pull.d_key_base = c["keys"].data_ptr() if (c["keys"] is not None and n) else None
            a.pull.d_key_index = c["key_index"].data_ptr() if (c["key_index"] is not None and n) else None
            a.pull.num_ids = shard.num_ids
            a.d_shard_rows = shard.rows.data_ptr()
            a.shard_rows = int(shard.rows.shape[0])
            a.d_shard_index = shard.index.data_ptr()
            a.dim = shard.dim
            a.kind, a.update = c["kind"], upd
            a.d_out = out.data_ptr() if n else None
            a.d_stats = c["stats"] if c["kind"] != 2 else None
        any_flag = C.c_int(0)
        rows = (C.c_uint64 * nctx)()
        nbytes = (C.c_uint64 * nctx)()
        _capi.check(lib.gf_pull_round(
            sess[0], self._node.h if self._node is not None else None,
            self._edge.h if self._edge is not None else None, arr, nctx, int(flag),
            C.byref(any_flag), rows, nbytes, self._pull_flag().data_ptr(), self._stream()))
        sh.host_syncs += 1
        for k, c in enumerate(ctxs):
            c["shard"].rows_pulled += rows[k]
            c["shard"].bytes_sent += nbytes[k]
            sh.rows_pulled += rows[k]
            sh.bytes_sent += nbytes[k]
        return outs, not any_flag.value

    def _pull_round(self, ctxs, upd, flag=0):
        """One fetch round over sharded tables; returns (rows per context, True iff no rank
        raised `flag`).  A collective: every rank calls it with the same kinds of contexts."""
        lib, dev, sh = self._lib, self.device, self._shards
        P, nctx = sh.P, len(ctxs)
        if P == 1 and not sh.always_exchange:
            return self._pull_round_native(ctxs, upd, flag, None)
        comm = sh.comm()
        if comm is not None:
            return self._pull_round_native(ctxs, upd, flag, comm)
        st = self._stream()
        i32, i64, f32 = torch.int32, torch.int64, torch.float32
        descs = (_capi.GfPullDesc * nctx)()
        for k, c in enumerate(ctxs):
            n = int(c["ids"].shape[0])
            c["n"] = n
            c["send_ids"] = self._arena("send%d" % k, max(n, 1), i64)
            c["req_pos"] = self._arena("pos%d" % k, max(n, 1), i32)
            d = descs[k]
            d.cache = c["cache"].h if c["cache"] is not None else None
            d.d_ids = c["ids"].data_ptr() if n else None
            d.n = n
            d.d_key_base = c["keys"].data_ptr() if (c["keys"] is not None and n) else None
            d.d_key_index = c["key_index"].data_ptr() if (c["key_index"] is not None and n) else None
            d.num_ids = c["shard"].num_ids
            d.d_send_ids = c["send_ids"].data_ptr()
            d.d_req_pos = c["req_pos"].data_ptr()
        # 1. claims + per-owner counts; row nctx of the table carries this rank's flag
        words = (nctx + 1) * P
        both = self._arena("counts", 2 * words + nctx * P, i32)
        counts = both[:words].view(nctx + 1, P)
        _capi.check(lib.gf_pull_count(descs, nctx, P, counts.data_ptr(), dev.index, st))
        counts[nctx].fill_(int(flag))
        recv_counts = sh.exchange_counts(counts)
        both[words:2 * words].view(nctx + 1, P).copy_(recv_counts)
        # 2. the round's one host synchronisation: own and received counts, through pinned memory
        pin = self.__dict__.get("_pin_counts")
        if pin is None or pin.numel() < 2 * words:
            pin = self._pin_counts = torch.empty(max(2 * words, 256), dtype=i32).pin_memory()
        pin[:2 * words].copy_(both[:2 * words], non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()
        sh.host_syncs += 1
        host = pin[:2 * words].view(2, nctx + 1, P)
        sc = host[0, :nctx].tolist()
        rc = host[1, :nctx].tolist()
        all_fit = not bool(host[1, nctx].any()) and not flag
        # 3. ids into the compact owner-major send buffers (offsets: prefix of the counts)
        cursor = both[2 * words:2 * words + nctx * P]
        _capi.check(lib.gf_pull_scatter(descs, nctx, P, counts.data_ptr(), cursor.data_ptr(),
                                        dev.index, st))
        n_send = [sum(x) for x in sc]
        n_recv = [sum(x) for x in rc]
        send = [c["send_ids"][:n_send[k]] for k, c in enumerate(ctxs)]
        got = [self._arena("got%d" % k, max(n_recv[k], 1), i64)[:n_recv[k]] for k in range(nctx)]
        sh.exchange_segments(send, sc, got, rc)
        # 4. this rank serves what it was asked for ...
        served = []
        for k, c in enumerate(ctxs):
            shard = c["shard"]
            rows = self._arena("served%d" % k, max(n_recv[k], 1) * shard.dim, f32)
            rows = rows[:n_recv[k] * shard.dim].view(n_recv[k], shard.dim)
            if n_recv[k]:
                _capi.check(lib.gf_gather_rows_indexed(
                    shard.rows.data_ptr(), int(shard.rows.shape[0]), shard.dim,
                    shard.index.data_ptr(), shard.num_ids, got[k].data_ptr(), n_recv[k],
                    rows.data_ptr(), self._pull_flag().data_ptr(), dev.index, st))
            served.append(rows)
        # 5. ... and the rows come back in the order of the ids
        pulled = []
        for k, c in enumerate(ctxs):
            dim = c["shard"].dim
            t = self._arena("pulled%d" % k, max(n_send[k], 1) * dim, f32)
            pulled.append(t[:max(n_send[k], 1) * dim].view(max(n_send[k], 1), dim))
        sh.exchange_segments(served, rc, [p[:n_send[k]] for k, p in enumerate(pulled)], sc)
        me = sh.rank
        for k, c in enumerate(ctxs):      # traffic figures, per shard and in all
            rows = n_send[k] - sc[k][me]
            nbytes = 8 * rows + 4 * c["shard"].dim * (n_recv[k] - rc[k][me])
            c["shard"].rows_pulled += rows
            c["shard"].bytes_sent += nbytes
            sh.rows_pulled += rows
            sh.bytes_sent += nbytes
        # 6. gather + replacement, the pulled rows standing in for the local table
        fdescs = (_capi.GfFetchPulledDesc * nctx)()
        outs, nf = [], 0
        for k, c in enumerate(ctxs):
            n, dim = c["n"], c["shard"].dim
            if c["kind"] == 2:       # cache-free: every row travelled, req_pos is its place
                outs.append(pulled[k][c["req_pos"][:n].long()] if n else
                            torch.empty((0, dim), dtype=f32, device=dev))
                continue
            out = torch.empty((n, dim), dtype=f32, device=dev)
            outs.append(out)
            if n == 0:
                continue
            f = fdescs[nf]
            nf += 1
            f.kind, f.update = c["kind"], upd
            f.d_ids, f.n, f.d_out = c["ids"].data_ptr(), n, out.data_ptr()
            f.d_stats = c["stats"]
            f.d_pulled_rows = pulled[k].data_ptr()
            f.d_req_pos = c["req_pos"].data_ptr()
        if nf:
            _capi.check(lib.gf_cache_fetch_blocks_pulled(
                self._node.h if self._node is not None else None,
                self._edge.h if self._edge is not None else None, fdescs, nf, st))
        # every buffer above is used on this stream only: the next round's kernels, which reuse
        # the arena, are ordered behind this round's
        return outs, all_fit

    def _pull_flag(self):
        f = getattr(self, "_pull_flag_t", None)
        if f is None:
            f = self._pull_flag_t = torch.zeros(1, dtype=torch.int32, device=self.device)
        return f

    def check_pulls(self):
        """Raises KeyError if any rank asked this one for a row it does not own since the last
        check (a device flag: reading it synchronises)."""
        f = getattr(self, "_pull_flag_t", None)
        if f is not None and int(f.item()):
            f.zero_()
            raise KeyError("sharded feature pull: asked for an id this rank does not own")

    def _fetch_distributed_torch(self, mfgs, eid, update_cache, target_edge_features):
        """The pull expressed with torch ops (probe -> unique -> FeatureShards.pull), one pull
        per block and kind: used for fetches that do not update the cache."""
        dev = self.device
        upd = 1 if update_cache else 0
        with torch.cuda.device(dev):
            n_node = len(mfgs[0]) if self._node is not None else 0
            n_edge = sum(len(mfg) for mfg in mfgs) if self._edge is not None else 0
            pos = self._stats_rows(n_node + n_edge)
            ring = self._stats_ring
            k = 0
            if self._node is not None:
                for b in mfgs[0]:
                    ids = self._ids(b.srcdata['ID'])
                    out = self._fetch_pulled(self._node, self._shards.node, ids, ids, upd,
                                             ring.data_ptr() + 64 * (pos + k))
                    b.srcdata['h'] = out
                    k += 1
            if self._edge is not None:
                for mfg in mfgs:
                    for b in mfg:
                        ids = self._ids(b.edata['ID'])
                        if ids.shape[0]:
                            # owner of an edge's features: its source node = the block's root
                            keys = self._ids(b.srcdata['ID'])[b.edges()[1]]
                        else:
                            keys = ids
                        out = self._fetch_pulled(self._edge, self._shards.edge, ids, keys, upd,
                                                 ring.data_ptr() + 64 * (pos + k))
                        if ids.shape[0]:
                            b.edata['f'] = out
                        k += 1
                if target_edge_features and eid is not None:
                    t = self._ids(eid)
                    # cache.py:403-409: the batch's source nodes are the first roots
                    num = mfgs[-1][0].num_dst_nodes() // (self.neg_sample_ratio + 2)
                    nid = self._ids(mfgs[-1][0].srcdata['ID'])[:num]
                    if nid.shape[0] != t.shape[0]:
                        raise ValueError("target edge ids and the batch's source roots differ "
                                         "in length")
                    self._target_edge_thunk = None
                    self._target_edge_features = self._shards.edge.pull(t, nid)
            self._stats_span = (pos, n_node, n_node + n_edge, ring, 0)
        return mfgs

    def _id_array(self, b, which):
        """(address, count, keepalive) of a block's id array: straight from the sampler's
        output when the block still has it (no torch view is created), else the tensor."""
        raw = getattr(b, "_raw", None)      # MFGBlock.raw_ids(), inlined (three calls per step)
        if raw is not None:
            d = b._srcdata if which == "src" else b._edata
            if d is None or not dict.__contains__(d, 'ID'):
                return (raw[1], b._num_src, b) if which == "src" else (raw[4], b._num_edges, b)
        t = self._ids((b.srcdata if which == "src" else b.edata)['ID'])
        return t.data_ptr(), int(t.shape[0]), t

    def _submit(self, mfgs, jobs, n_node, n_cached, upd, async_enqueue, aliases=(), ann=None):
        # one output allocation for the whole call; every block's rows start 16-byte aligned
        offs, total, alg, moved = [], 0, 0, 0
        for job in jobs:
            n, dim = job[2], job[4]
            offs.append(total)
            total += (n * dim + 3) & ~3
            alg += n * (8 + 8 * dim)
            moved += n
        out_all, out_base, out_ptr = self._out_buffer(total)
        stats_pos = self._stats_rows(n_cached)
        stats_ptr = self._stats_ring.data_ptr() + 64 * stats_pos
        nj = len(jobs)
        self.algorithmic_bytes += alg
        self.rows_moved += moved
        descs, cdescs, cdescs_ref = self._desc_buf(nj)
        box = [0]     # this submission's ticket, for the thunks below
        pack = _DESC.pack_into
        for i, (kind, ids_ptr, n, _keep, dim, b, which, key) in enumerate(jobs):
            off = offs[i]
            pack(descs, i * _DESC.size, kind, upd, ids_ptr or 0, n, out_ptr + 4 * off,
                 stats_ptr + 64 * i if (kind != 2 and n) else 0)

            def rows(off=out_base + off, n=n, dim=dim, sync=async_enqueue):
                if sync:
                    self.wait_enqueued(box[0])
                return out_all[off:off + n * dim].view(n, dim)
            if b is None:
                self._target_edge_thunk = rows      # cache.py:411 `edge_feats[eid]`
                self._target_edge_features = None
            elif hasattr(b, "set_lazy"):
                b.set_lazy(which, key, rows)
            else:
                (b.srcdata if which == "src" else b.edata)[key] = rows(sync=False)
        for b, owner, n in aliases:
            def prefix_rows(off=out_base + offs[owner], n=n, dim=jobs[owner][4],
                            sync=async_enqueue):
                if sync:
                    self.wait_enqueued(box[0])
                return out_all[off:off + n * dim].view(n, dim)
            b.set_lazy("e", "f", prefix_rows)
        node_h = self._node.h if self._node is not None else None
        edge_h = self._edge.h if self._edge is not None else None
        self.num_gather_launches += max(n_node, n_cached - n_node, 1)
        if async_enqueue:
            word = self.__dict__.get("_ticket_word")
            if word is None:
                t = C.c_uint64(0)
                word = self._ticket_word = (t, C.byref(t))
            ticket = word[0]
            if ann is not None:
                rc = self._lib.gf_cache_fetch_announce_async(
                    node_h, edge_h, cdescs_ref, nj, self._stream(), ann[0], ann[1], ann[2], ann[3],
                    ann[4], ann[5], word[1])
            else:
                rc = self._lib.gf_cache_fetch_blocks_async(
                    node_h, edge_h, cdescs_ref, nj, self._stream(), word[1])
            if rc:
                _capi.check(rc)
            box[0] = ticket.value
            # ids / outputs / descriptors must outlive the enqueue
            self._tickets.append((ticket.value, (jobs, descs, cdescs, mfgs, out_all,
                                                 self._stats_ring, ann)))
        else:
            _capi.check(self._lib.gf_cache_fetch_blocks(
                node_h, edge_h, cdescs_ref, nj, self._stream()))
        # (aliased blocks count as all hits — of a cache that exists)
        self._stats_span = (stats_pos, n_node, n_cached, self._stats_ring,
                            len(aliases) if self.edge_capacity else 0)
        return mfgs

    def _out_buffer(self, total):
        """Output rows of one fetch_feature() call: (slab tensor, first float of this call's
        share, its device address).  Carved out of a slab that is allocated
        once per 8 calls with a power-of-two size per call: block sizes differ from batch to
        batch, and a fresh `torch.empty` of a never-seen size is a real hipMalloc in the
        caching allocator — during the first replays of a stream those serialised the enqueue
        thread (38 us of launch work per step instead of 26).  Uniform slabs are recycled by
        the allocator as soon as the MFGs of an old slab are gone; a slab is never reused
        while any of its views is alive."""
        slab = getattr(self, "_out_slab", None)
        if slab is None or total > slab[1] or slab[2] >= slab[3]:
            cap = 1 << max(int(total - 1).bit_length(), 16)
            if slab is not None:
                cap = max(cap, slab[1])
            count = max(1, min(8, (1 << 28) // cap))        # at most 1 GiB of floats per slab
            mem = torch.empty(cap * count, dtype=torch.float32, device=self.device)
            slab = self._out_slab = [mem, cap, 0, count, mem.data_ptr()]
        i = slab[2]
        slab[2] = i + 1
        base = i * slab[1]
        return slab[0], base, slab[4] + 4 * base

    def _desc_buf(self, n):
        """(bytearray, its ctypes view, a reference to pass) from a small ring (the native call copies the
        descriptors before it returns; the ring only saves building the view per call)."""
        ring = self.__dict__.get("_desc_ring")
        if ring is None:
            ring = self._desc_ring = []
            for _ in range(8):      # more than fetches can be queued (_MAX_QUEUED)
                buf = bytearray(_DESC.size * 64)
                view = _capi.GfFetchDesc.from_buffer(buf)
                ring.append((buf, view, C.byref(view)))
            self._desc_next = 0
        if n > 64:
            buf = bytearray(_DESC.size * n)
            view = _capi.GfFetchDesc.from_buffer(buf)
            return buf, view, C.byref(view)
        self._desc_next = (self._desc_next + 1) % len(ring)
        return ring[self._desc_next]
