"""Pins the numpy cache oracle (oracle/cache_oracle.py) to the outputs of the
reference's own Python cache (tests/golden/cache_reference.npz, produced by
tests/golden/make_cache_fixtures.py from /root/reference/gnnflow/cache)."""
import hashlib
import os

import numpy as np
import pytest

from oracle.cache_oracle import OracleLRUCache

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cache_reference.npz")


class Blk:
    def __init__(self, src, edge):
        self.srcdata = {"ID": src}
        self.edata = {"ID": edge}


def load():
    z = np.load(FIX)
    return z, [str(s) for s in z["scenarios"]]


def check_rows(z, key, arr):
    arr = np.ascontiguousarray(arr, np.float32)
    assert tuple(z[key + "/shape"]) == arr.shape, key
    digest = np.frombuffer(hashlib.sha256(arr.tobytes()).digest(), np.uint8)
    assert np.array_equal(digest, z[key + "/sha256"]), key
    if key in z.files:
        assert np.array_equal(arr, z[key]), key


def replay(z, name, make_cache, to_ids=lambda a: a, to_np=lambda a: a):
    """Replays one recorded scenario against a cache implementation; yields per batch
    (batch index, cache, blocks)."""
    N, E, dn, de, nb = (int(x) for x in z[name + "/meta"])
    ratio = float(z[name + "/ratio"][0])
    nf = z[name + "/node_feats"] if dn else None
    ef = z[name + "/edge_feats"] if de else None
    cache = make_cache(ratio, N, E, nf, ef, dn, de)
    cache.init_cache()
    for b in range(nb):
        blocks = [[Blk(to_ids(z["{}/b{}/src_ids{}".format(name, b, li)]),
                       to_ids(z["{}/b{}/edge_ids{}".format(name, b, li)]))] for li in range(2)]
        cache.fetch_feature(blocks, z["{}/b{}/eid".format(name, b)])
        if dn:
            check_rows(z, "{}/b{}/h".format(name, b), to_np(blocks[0][0].srcdata["h"]))
        if de:
            for li in range(2):
                check_rows(z, "{}/b{}/f{}".format(name, b, li), to_np(blocks[li][0].edata["f"]))
            check_rows(z, "{}/b{}/target".format(name, b), to_np(cache.target_edge_features))
        yield b, cache, blocks
        if b == int(z[name + "/reset_after"][0]):
            cache.reset()


def policy_of(z, name):
    return str(z[name + "/policy"][0])


def _mk(rule, policy="lru"):
    def make(ratio, N, E, nf, ef, dn, de):
        nf = None if nf is None else nf.astype(np.float32)   # cache.py:71-74 bool -> f32
        ef = None if ef is None else ef.astype(np.float32)
        return OracleLRUCache(ratio, ratio, N, E, nf, ef, dn, de, overflow_rule=rule,
                              policy=policy)
    return make


@pytest.mark.parametrize("name", load()[1])
@pytest.mark.parametrize("rule", ["smallest_ids", "first_seen"])
def test_features_match_reference(name, rule):
    """Fetched rows are the reference's, bit for bit, in every block of every batch,
    whatever the cache state (both overflow rules)."""
    z, _ = load()
    for _ in replay(z, name, _mk(rule, policy_of(z, name))):
        pass


@pytest.mark.parametrize("name", load()[1])
def test_first_batch_hit_ratio_and_occupancy(name):
    """Batch 0 starts from the init_cache() state on both sides, so the node ratio and
    the first edge block see identical caches: ratios must agree exactly.  The number
    of cached ids agrees after every batch (which ids depends on torch.topk's
    unspecified tie order, so only batch-0-before-update state is compared)."""
    z, _ = load()
    N, E, dn, de, nb = (int(x) for x in z[name + "/meta"])
    for b, cache, blocks in replay(z, name, _mk("smallest_ids", policy_of(z, name))):
        if dn:
            assert len(cache.node.cached_ids()) == len(z["{}/b{}/node_cached".format(name, b)])
            if b == 0:
                assert cache.cache_node_ratio == pytest.approx(
                    float(z["{}/b0/node_ratio".format(name)][0]), abs=1e-7)
        if de:
            assert len(cache.edge.cached_ids()) == len(z["{}/b{}/edge_cached".format(name, b)])


@pytest.mark.parametrize("name", ["cap1_tie_free", "lfu_cap1_tie_free", "fifo_small",
                                  "fifo_overflow_wrap"])
def test_tie_free_sequence_matches_reference_ratios(name):
    """Where torch.topk has no tie to break — a capacity-1 LRU/LFU cache, and FIFO at any
    capacity (its victims come from a pointer, fifo_cache.py:96-105) — the reference's
    whole hit-ratio sequence and cached-id sets are reproduced exactly."""
    z, _ = load()
    for b, cache, blocks in replay(z, name, _mk("smallest_ids", policy_of(z, name))):
        assert cache.cache_node_ratio == pytest.approx(
            float(z["{}/b{}/node_ratio".format(name, b)][0]), abs=1e-7), b
        assert cache.cache_edge_ratio == pytest.approx(
            float(z["{}/b{}/edge_ratio".format(name, b)][0]), abs=1e-7), b
        assert np.array_equal(cache.node.cached_ids(), z["{}/b{}/node_cached".format(name, b)])
        assert np.array_equal(cache.edge.cached_ids(), z["{}/b{}/edge_cached".format(name, b)])
