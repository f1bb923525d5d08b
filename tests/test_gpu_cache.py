"""GPU parity, part 3: fused HIP gather + LRU cache against (a) the reference's own
Python cache outputs (tests/golden/cache_reference.npz) and (b) the numpy oracle with
the same deterministic tie rule — fetched rows bit-exact, hit counts and cached-id sets
exact, over sequences of batches, duplicates, overflow, empty blocks, odd dims, and both
feature placements."""
import numpy as np
import pytest

from tests.test_oracle_cache import Blk, load, policy_of, replay

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=["list", "list2", "queue", "mixed"])
def lru_form(request, monkeypatch):
    """Every test of this module runs three times: with the LRU order kept as a list and
    updated in ONE launch (small caches, the default), as a list updated by the two launches
    list scan + list install (what caches beyond the one-launch kernel's tables use;
    GNNFLOW_LRU_FUSED=0), and as a queue with dead entries (what caches of >= 0.5 M slots use;
    forced here by lowering that bound to 1 slot, which also exercises its compaction every
    other update and the fall-back to the list form for blocks of more than capacity / 4
    rows).  "mixed": the bound at 200 slots, so that a test's larger cache is a queue and its
    smaller one a list updated on the round's side stream beside it.  All must make the
    oracle's decisions."""
    if request.param == "queue":
        monkeypatch.setenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", "1")
    elif request.param == "mixed":
        monkeypatch.setenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", "200")
    else:
        monkeypatch.delenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", raising=False)
    if request.param == "list2":
        monkeypatch.setenv("GNNFLOW_LRU_FUSED", "0")
    else:
        monkeypatch.delenv("GNNFLOW_LRU_FUSED", raising=False)
    return request.param


def _cls(policy):
    import gnnflow_amd.cache as caches
    return {"lru": caches.LRUCache, "lfu": caches.LFUCache, "fifo": caches.FIFOCache}[policy]


def _hip_cache(placement="device", policy="lru"):
    import torch

    def make(ratio, N, E, nf, ef, dn, de):
        return _cls(policy)(ratio, ratio, N, E, "cuda:0",
                        None if nf is None else torch.from_numpy(nf),
                        None if ef is None else torch.from_numpy(ef), dn, de,
                        feature_placement=placement)
    return make


def _to_ids(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, np.int64)).cuda()


def _to_np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("placement", ["device", "pinned"])
@pytest.mark.parametrize("name", load()[1])
def test_rows_match_reference_python(name, placement):
    z, _ = load()
    """Every fetched row equals the reference's own output (sha256 of each block,
    full arrays for batch 0); hit ratios are compared against the oracle below."""
    for _ in replay(z, name, _hip_cache(placement, policy_of(z, name)), _to_ids, _to_np):
        pass


def _run_against_oracle(N, E, dn, de, ratio, batches, seed, skew=1.0, policy="lru",
                        reset_after=()):
    import torch
    from oracle.cache_oracle import OracleLRUCache
    LRUCache = _cls(policy)
    rng = np.random.RandomState(seed)
    nf = rng.rand(N, dn).astype(np.float32) if dn else None
    ef = rng.rand(E, de).astype(np.float32) if de else None
    hip = LRUCache(ratio, ratio, N, E, "cuda:0",
                   None if nf is None else torch.from_numpy(nf),
                   None if ef is None else torch.from_numpy(ef), dn, de)
    ora = OracleLRUCache(ratio, ratio, N, E, nf, ef, dn, de, overflow_rule="first_seen",
                         policy=policy)
    hip.init_cache()
    ora.init_cache()

    def draw(high, n):
        if skew <= 0:
            return rng.randint(0, high, n).astype(np.int64)
        p = np.arange(1, high + 1, dtype=np.float64) ** (-skew)
        p /= p.sum()
        return rng.choice(high, size=n, p=p).astype(np.int64)

    for bi, (nsrc, nedges) in enumerate(batches):
        src = [draw(N, nsrc), draw(N, max(nsrc // 3, 1))]
        edg = [draw(E, ne) if ne else np.zeros(0, np.int64) for ne in nedges]
        hb = [[Blk(_to_ids(src[0]), _to_ids(edg[0]))], [Blk(_to_ids(src[1]), _to_ids(edg[1]))]]
        ob = [[Blk(src[0], edg[0])], [Blk(src[1], edg[1])]]
        eid = draw(E, 7)
        hip.fetch_feature(hb, eid)
        ora.fetch_feature(ob, eid)
        if dn:
            assert np.array_equal(_to_np(hb[0][0].srcdata["h"]), ob[0][0].srcdata["h"]), bi
            assert float(hip.cache_node_ratio) == pytest.approx(ora.cache_node_ratio, abs=1e-6), bi
            assert np.array_equal(np.sort(hip._node.slot_ids()[hip._node.slot_ids() >= 0]),
                                  ora.node.cached_ids()), bi
        if de:
            for li in range(2):
                if len(edg[li]):
                    assert np.array_equal(_to_np(hb[li][0].edata["f"]), ob[li][0].edata["f"]), bi
                else:
                    assert "f" not in hb[li][0].edata
            assert float(hip.cache_edge_ratio) == pytest.approx(ora.cache_edge_ratio, abs=1e-6), bi
            ids = hip._edge.slot_ids()
            assert np.array_equal(np.sort(ids[ids >= 0]), ora.edge.cached_ids()), bi
            assert np.array_equal(_to_np(hip.target_edge_features), ora.target_edge_features), bi
        if bi in reset_after:
            hip.reset()
            ora.reset()
    return hip, ora


@pytest.mark.parametrize("policy", ["lfu", "fifo"])
def test_policy_sequence_matches_oracle(policy):
    """LFUCache / FIFOCache (lfu_cache.py:134-210, fifo_cache.py:77-161) on the fused
    gather: rows, hit ratios and cached-id sets equal the oracle's after every batch,
    including a reset() in the middle (LFU: edge re-init with counts 0; FIFO: pointer
    rewind only) and blocks with more distinct misses than slots."""
    _run_against_oracle(N=3000, E=40000, dn=172, de=172, ratio=0.2,
                        batches=[(5000, (6000, 700))] * 6, seed=11, skew=0.8, policy=policy,
                        reset_after=(2,))
    _run_against_oracle(N=400, E=400, dn=4, de=8, ratio=0.05,
                        batches=[(300, (350, 60))] * 5, seed=12, skew=0.0, policy=policy,
                        reset_after=(1,))
    _run_against_oracle(N=500, E=3000, dn=7, de=13, ratio=0.3,
                        batches=[(300, (400, 0)), (10, (1, 50)), (640, (64, 65))] * 3,
                        seed=13, policy=policy)


@pytest.mark.parametrize("policy", ["lru", "lfu"])
def test_threshold_beyond_the_fine_histogram_bins(policy):
    """Slots left untouched for more than 2047 updates put the eviction threshold into a
    coarse level-1 bin: the level-2 histogram + last-workgroup path must pick exactly the
    oracle's victims (4096 slots, one new id per update, > 2048 updates; LFU: counts are
    far from the clamp, so this is its plain path over the same long sequence)."""
    import torch
    from oracle.cache_oracle import OracleLRUCache
    N, d, cap = 20000, 4, 4096
    rng = np.random.RandomState(31)
    nf = rng.rand(N, d).astype(np.float32)
    hip = _cls(policy)(0.0, cap / N, N, 16, "cuda:0", torch.from_numpy(nf), None, d, 0)
    ora = OracleLRUCache(0.0, cap / N, N, 16, nf, None, d, 0, overflow_rule="first_seen",
                         policy=policy)
    assert hip.node_capacity == cap == ora.node_capacity
    hip.init_cache()
    ora.init_cache()
    fresh = cap
    for bi in range(2300):
        # a few recently installed ids (hits), one or two never-seen ids (misses)
        k_new = 1 + (bi % 97 == 0)
        recent = np.arange(max(cap, fresh - 5), fresh)
        ids = np.concatenate([recent, np.arange(fresh, fresh + k_new)]).astype(np.int64)
        fresh += k_new
        hb = [[Blk(_to_ids(ids), _to_ids(np.zeros(0, np.int64)))]]
        ob = [[Blk(ids, np.zeros(0, np.int64))]]
        hip.fetch_feature(hb, None, target_edge_features=False)
        ora.fetch_feature(ob, None, target_edge_features=False)
        if bi % 100 == 99 or bi > 2040:
            assert np.array_equal(_to_np(hb[0][0].srcdata["h"]), ob[0][0].srcdata["h"]), bi
            assert float(hip.cache_node_ratio) == pytest.approx(ora.cache_node_ratio, abs=1e-6), bi
            got = hip._node.slot_ids()
            assert np.array_equal(np.sort(got[got >= 0]), ora.node.cached_ids()), bi


def test_gnnlab_static_cache_presamples_and_never_replaces():
    """GNNLabStaticCache (gnnlab_static_cache.py:87-182): the cached ids are the most often
    pre-sampled ones (counted per block as the reference's `count[ids] += 1` does), ties to
    the lowest id; fetches never change the cache."""
    import torch
    from gnnflow_amd import DynamicGraph, TemporalSampler
    from gnnflow_amd.cache import GNNLabStaticCache
    rng = np.random.RandomState(21)
    N, E = 300, 6000
    src = (rng.zipf(1.6, E) % N).astype(np.int64)
    dst = rng.randint(0, N, E).astype(np.int64)
    ts = np.sort(rng.rand(E).astype(np.float32))
    g = DynamicGraph(1 << 20, 64 << 20, "cuda", 16, 64, "insert", device=0)
    g.add_edges(src, dst, ts, add_reverse=True)
    sampler = TemporalSampler(g, fanouts=[5, 5], sample_strategy="recent")
    nf = rng.rand(N, 16).astype(np.float32)
    ef = rng.rand(E, 12).astype(np.float32)
    cache = GNNLabStaticCache(0.1, 0.1, N, E, "cuda:0", torch.from_numpy(nf),
                              torch.from_numpy(ef), 16, 12)
    train = {"src": src[:3000], "dst": dst[:3000], "time": ts[:3000]}
    cache.init_cache(sampler=sampler, train_df=train, pre_sampling_rounds=2, batch_size=500)

    node_cnt = np.zeros(N, np.int64)
    edge_cnt = np.zeros(E, np.int64)
    for _ in range(2):
        for lo in range(0, 3000, 500):
            roots = np.concatenate([src[lo:lo + 500], dst[lo:lo + 500]])
            t = np.concatenate([ts[lo:lo + 500], ts[lo:lo + 500]])
            mfgs = sampler.sample(roots, t)
            node_cnt[np.unique(_to_np(mfgs[0][0].srcdata["ID"]))] += 1
            for mfg in mfgs:
                for b in mfg:
                    if b.num_src_nodes() > b.num_dst_nodes():
                        edge_cnt[np.unique(_to_np(b.edata["ID"]))] += 1
    want_nodes = np.argsort(-node_cnt, kind="stable")[:cache.node_capacity]
    want_edges = np.argsort(-edge_cnt, kind="stable")[:cache.edge_capacity]
    assert np.array_equal(cache._node.slot_ids(), want_nodes)
    assert np.array_equal(cache._edge.slot_ids(), want_edges)

    roots = np.concatenate([src[3000:3600], dst[3000:3600]])
    t = np.concatenate([ts[3000:3600], ts[3000:3600]])
    for _ in range(2):
        mfgs = sampler.sample(roots, t)
        cache.fetch_feature(mfgs, np.arange(10))
        ids = _to_np(mfgs[0][0].srcdata["ID"])
        assert np.array_equal(_to_np(mfgs[0][0].srcdata["h"]), nf[ids])
        assert float(cache.cache_node_ratio) == pytest.approx(
            np.isin(ids, want_nodes).mean(), abs=1e-6)
        eids = _to_np(mfgs[0][0].edata["ID"])
        assert np.array_equal(_to_np(mfgs[0][0].edata["f"]), ef[eids])
        assert np.array_equal(cache._node.slot_ids(), want_nodes)
        assert np.array_equal(cache._edge.slot_ids(), want_edges)
    assert cache.name == "gnnlab"


def test_sequence_matches_oracle_reddit_like_dims():
    _run_against_oracle(N=3000, E=40000, dn=172, de=172, ratio=0.2,
                        batches=[(5000, (6000, 700))] * 6, seed=1, skew=0.8)


def test_sequence_matches_oracle_odd_dims_and_empty_blocks():
    # dim not a multiple of 4 -> scalar path; an empty edge block is skipped
    _run_against_oracle(N=500, E=3000, dn=7, de=13, ratio=0.3,
                        batches=[(300, (400, 0)), (10, (1, 50)), (640, (64, 65))] * 2, seed=2)


def test_overflow_more_unique_misses_than_slots():
    _run_against_oracle(N=400, E=400, dn=4, de=8, ratio=0.05,
                        batches=[(300, (350, 60))] * 4, seed=3, skew=0.0)


def test_all_hits_do_not_touch_lru_state():
    """lru_cache.py / cache.py:318: update_*_cache only runs when there is a miss."""
    hip, ora = _run_against_oracle(N=100, E=100, dn=4, de=4, ratio=1.0,
                                   batches=[(50, (50, 5))] * 3, seed=4)
    assert float(hip.cache_node_ratio) == 1.0


def test_zero_ratio_is_cache_free_gather():
    hip, _ = _run_against_oracle(N=64, E=64, dn=8, de=8, ratio=0.0,
                                 batches=[(40, (40, 4))] * 2, seed=5)
    assert float(hip.cache_edge_ratio) == 0.0


def test_reset_only_resets_edge_cache():
    hip, ora = _run_against_oracle(N=200, E=800, dn=4, de=4, ratio=0.2,
                                   batches=[(100, (200, 20))] * 3, seed=6)
    node_before = hip._node.slot_ids().copy()
    hip.reset()
    ora.reset()
    assert np.array_equal(hip._node.slot_ids(), node_before)
    assert np.array_equal(hip._edge.slot_ids(), np.arange(hip.edge_capacity))
    assert np.array_equal(np.sort(hip._edge.slot_ids()), ora.edge.cached_ids())


def test_rejects_cpu_device():
    import torch
    from gnnflow_amd.cache import LRUCache
    with pytest.raises(ValueError):
        LRUCache(0.2, 0.2, 10, 10, "cpu", torch.zeros(10, 4), None, 4, 0)


def test_large_block_round_trip_properties():
    """Full-size block (198 000 rows x 172 floats): size-independent checks — every row
    equals its table row (checksum of checksums), and a second fetch of the same ids is
    identical (idempotence) with a hit ratio that can only go up."""
    import torch
    from gnnflow_amd.cache import LRUCache
    E, d, n = 672447, 172, 198000
    g = torch.Generator(device="cuda").manual_seed(0)
    ef = torch.rand((E, d), generator=g, device="cuda")
    ids = torch.randint(0, E, (n,), generator=g, device="cuda")
    cache = LRUCache(0.2, 0.0, 10, E, "cuda:0", None, ef, 0, d)
    cache.init_cache()
    b1 = [[Blk(torch.zeros(1, dtype=torch.int64, device="cuda"), ids)]]
    cache.fetch_feature(b1)
    r1 = float(cache.cache_edge_ratio)
    f1 = b1[0][0].edata["f"]
    assert torch.equal(f1, ef[ids])
    row_sums = f1.double().sum(dim=1)
    assert torch.equal(row_sums, ef.double().sum(dim=1)[ids])
    b2 = [[Blk(torch.zeros(1, dtype=torch.int64, device="cuda"), ids)]]
    cache.fetch_feature(b2)
    assert torch.equal(b2[0][0].edata["f"], f1)
    assert float(cache.cache_edge_ratio) >= r1


@pytest.mark.parametrize("rows", [(0, 36000, 600), (25000, 0, 600), (25000, 11000, 0),
                                  (24832, 11264, 600), (300000, 5, 1), (40, 70000, 3)])
def test_direct_gather_one_row_grid_shapes(rows):
    """The direct gather (tables in HBM, no row mirror, 16-byte rows) launches ONE row of
    workgroups, the contexts' back to back, with tiles fitted to two workgroups per CU where that
    is possible: empty contexts in every position, the headline's sizes, a block beyond the
    1 024-workgroup cap of a context (grid-stride tiles), tiny blocks beside large ones — every
    fetched row equals its table row, twice (the second fetch hits)."""
    import torch
    from gnnflow_amd.cache import LRUCache
    n_node, n_edge, n_target = rows
    N, E, d = 20000, 400000, 172
    g = torch.Generator(device="cuda").manual_seed(sum(rows))
    nf = torch.rand((N, d), generator=g, device="cuda")
    ef = torch.rand((E, d), generator=g, device="cuda")
    cache = LRUCache(0.2, 0.2, N, E, "cuda:0", nf, ef, d, d)
    cache.init_cache()
    nid = torch.randint(0, N, (n_node,), generator=g, device="cuda")
    eid = torch.randint(0, E, (n_edge,), generator=g, device="cuda")
    tid = torch.randint(0, E, (n_target,), generator=g, device="cuda")
    for _ in range(2):
        b = [[Blk(nid, eid)]]
        cache.fetch_feature(b, eid=tid if n_target else None)
        assert torch.equal(b[0][0].srcdata["h"], nf[nid])
        if n_edge:
            assert torch.equal(b[0][0].edata["f"], ef[eid])
        if n_target:
            assert torch.equal(cache.target_edge_features, ef[tid])


def test_async_enqueue_matches_sync():
    """fetch_feature(async_enqueue=True) hands the launches to the library's enqueue thread;
    results, hit ratios and LRU state must be those of the synchronous call."""
    import torch
    import gnnflow_amd
    from gnnflow_amd.cache import LRUCache
    from tests import synth
    N, E, d = 600, 20000, 16
    src, dst, ts, eid = synth.powerlaw_graph(N, E, seed=3)
    g = gnnflow_amd.DynamicGraph(1 << 20, 1 << 28, "cuda", 16, 128, "insert")
    g.add_edges(src, dst, ts, eid)
    s = gnnflow_amd.TemporalSampler(g, [6, 6])
    rng = np.random.RandomState(1)
    nf = torch.from_numpy(rng.rand(N, d).astype(np.float32))
    ef = torch.from_numpy(rng.rand(E, d).astype(np.float32))
    caches = [LRUCache(0.1, 0.1, N, E, "cuda:0", nf, ef, d, d) for _ in range(2)]
    for c in caches:
        c.init_cache()
    for it in range(6):
        nodes, t = synth.random_roots(N, 500, 1000.0, seed=it)
        e_ids = rng.randint(0, E, 50)
        ma, mb = s.sample(nodes, t), s.sample(nodes, t)
        caches[0].fetch_feature(ma, e_ids)
        caches[1].fetch_feature(mb, e_ids, async_enqueue=True)
        for la, lb in zip(ma, mb):
            for ba, bb in zip(la, lb):
                if ba.num_edges():
                    assert torch.equal(ba.edata["f"], bb.edata["f"])
        assert torch.equal(ma[0][0].srcdata["h"], mb[0][0].srcdata["h"])
        assert torch.equal(caches[0].target_edge_features, caches[1].target_edge_features)
        assert float(caches[0].cache_edge_ratio) == float(caches[1].cache_edge_ratio)
        assert float(caches[0].cache_node_ratio) == float(caches[1].cache_node_ratio)
    caches[1].wait_enqueued()
    assert np.array_equal(caches[0]._edge.slot_ids(), caches[1]._edge.slot_ids())
    assert np.array_equal(caches[0]._node.slot_ids(), caches[1]._node.slot_ids())


def test_huge_block_takes_chained_scan_path():
    """More than 1024 row tiles (> 4 194 304 rows in one block) fall back to the chained
    single-workgroup scan; results must still equal the oracle's."""
    import torch
    from gnnflow_amd.cache import LRUCache
    from oracle.cache_oracle import OracleLRUCache
    N, E, d = 8, 3000, 4
    rng = np.random.RandomState(9)
    ef = rng.rand(E, d).astype(np.float32)
    nf = rng.rand(N, d).astype(np.float32)
    hip = LRUCache(0.1, 0.0, N, E, "cuda:0", torch.from_numpy(nf), torch.from_numpy(ef), 0, d)
    ora = OracleLRUCache(0.1, 0.0, N, E, None, ef, 0, d, overflow_rule="first_seen")
    hip.init_cache()
    ora.init_cache()
    for it in range(2):
        ids = rng.randint(0, E, 4300000).astype(np.int64)
        hb = [[Blk(_to_ids(np.zeros(1, np.int64)), _to_ids(ids))]]
        ob = [[Blk(np.zeros(1, np.int64), ids)]]
        hip.fetch_feature(hb)
        ora.fetch_feature(ob)
        assert np.array_equal(_to_np(hb[0][0].edata["f"]), ob[0][0].edata["f"])
        assert float(hip.cache_edge_ratio) == pytest.approx(ora.cache_edge_ratio, abs=1e-6)
        sid = hip._edge.slot_ids()
        assert np.array_equal(np.sort(sid[sid >= 0]), ora.edge.cached_ids())


@pytest.mark.parametrize("seed", range(16))
def test_fuzzed_cache_sequences_match_oracle(seed):
    """Random table sizes, row widths (vectorised and scalar paths), cache ratios down to a
    single slot, skews, block sizes incl. empty and overflowing ones, replacement policies and
    resets — rows, hit ratios and cached-id sets against the oracle after every batch."""
    rng = np.random.RandomState(500 + seed)
    N = int(rng.choice([8, 130, 2500]))
    E = int(rng.choice([16, 900, 30000]))
    dn = int(rng.choice([0, 3, 8, 172]))
    de = int(rng.choice([1, 4, 13, 172])) if dn == 0 or rng.randint(2) else 0
    if dn == 0 and de == 0:
        de = 4
    ratio = float(rng.choice([0.01, 0.07, 0.3, 1.0]))
    policy = str(rng.choice(["lru", "lru", "lfu", "fifo"]))
    nb = int(rng.randint(3, 9))
    batches = []
    for _ in range(nb):
        nsrc = int(rng.choice([1, 40, 700, 4000]))
        batches.append((nsrc, (int(rng.choice([0, 1, 65, 3000])), int(rng.choice([0, 7, 500])))))
    resets = tuple(int(x) for x in rng.choice(nb, size=int(rng.randint(0, 2)), replace=False))
    _run_against_oracle(N=N, E=E, dn=dn, de=de, ratio=ratio, batches=batches, seed=seed,
                        skew=float(rng.choice([0.0, 0.7, 1.4])), policy=policy, reset_after=resets)


@pytest.mark.parametrize("policy", ["lru", "lfu", "fifo"])
@pytest.mark.parametrize("second", [4096, 4097, 6000, 16385])
def test_two_large_blocks_per_fetch_around_the_row_tile_size(policy, second):
    """Two edge blocks per fetch whose second one sits at / just over one scan row tile (4096)
    and at several tiles: duplicates, overflow of the capacity, resets, all three policies."""
    _run_against_oracle(N=900, E=20000, dn=8, de=12, ratio=0.05,
                        batches=[(500, (7000, second)), (300, (50, second)), (900, (9000, 1))] * 2,
                        seed=40 + second % 7, skew=0.9, policy=policy, reset_after=(2,))


def test_resize_crosses_from_the_list_into_the_queue_form(monkeypatch):
    """The edge cache has 200 slots before the resize and 340 after it; with the queue form
    starting at 300 slots the resize also changes the form of the replacement order (and the
    node cache, 40 -> 66 slots, stays a list)."""
    monkeypatch.setenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", "300")
    hip = _resize_scenario("lru")
    assert hip._edge.lru_state()["queue_form"] == 1 and hip._node.lru_state()["queue_form"] == 0


@pytest.mark.parametrize("policy", ["lru", "lfu", "fifo"])
def test_resize_grows_id_space_and_capacity_keeping_the_contents(policy):
    _resize_scenario(policy)


def _resize_scenario(policy):
    """Cache.resize (cache.py:197-221): after the graph has grown, cached ids still hit with
    the right rows, the new ids are served from the new tables and get cached, the capacity
    follows the ratio, and the replacement order continues (new slots are empty and oldest /
    unused / next on the FIFO ring).  The reference leaves the new elements uninitialised
    (torch resize_); the oracle holds the deterministic completion."""
    import torch
    from oracle.cache_oracle import OracleLRUCache
    rng = np.random.RandomState(71)
    N0, E0, N1, E1, dn, de = 200, 1000, 330, 1700, 8, 12
    nf = rng.rand(N1, dn).astype(np.float32)
    ef = rng.rand(E1, de).astype(np.float32)
    hip = _cls(policy)(0.2, 0.2, N0, E0, "cuda:0", torch.from_numpy(nf[:N0]),
                       torch.from_numpy(ef[:E0]), dn, de)
    ora = OracleLRUCache(0.2, 0.2, N0, E0, nf[:N0], ef[:E0], dn, de,
                         overflow_rule="first_seen", policy=policy)
    hip.init_cache()
    ora.init_cache()

    def batch(N, E, bi):
        src = rng.randint(0, N, 150).astype(np.int64)
        e0 = rng.randint(0, E, 260).astype(np.int64)
        e1 = rng.randint(0, E, 40).astype(np.int64)
        eid = rng.randint(0, E, 9).astype(np.int64)
        hb = [[Blk(_to_ids(src), _to_ids(e0))], [Blk(_to_ids(src[:30]), _to_ids(e1))]]
        ob = [[Blk(src, e0)], [Blk(src[:30], e1)]]
        hip.fetch_feature(hb, eid)
        ora.fetch_feature(ob, eid)
        assert np.array_equal(_to_np(hb[0][0].srcdata["h"]), ob[0][0].srcdata["h"]), bi
        for li in range(2):
            assert np.array_equal(_to_np(hb[li][0].edata["f"]), ob[li][0].edata["f"]), bi
        assert float(hip.cache_node_ratio) == pytest.approx(ora.cache_node_ratio, abs=1e-6), bi
        assert float(hip.cache_edge_ratio) == pytest.approx(ora.cache_edge_ratio, abs=1e-6), bi
        for kind, okind in ((hip._node, ora.node), (hip._edge, ora.edge)):
            ids = kind.slot_ids()
            assert np.array_equal(np.sort(ids[ids >= 0]), okind.cached_ids()), bi

    for bi in range(4):
        batch(N0, E0, bi)
    hip.node_feats, hip.edge_feats = torch.from_numpy(nf), torch.from_numpy(ef)
    before = hip.get_mem_size()
    hip.resize(N1, E1)
    ora.resize(N1, E1, nf, ef)
    assert (hip.num_nodes, hip.num_edges) == (N1, E1)
    assert (hip.node_capacity, hip.edge_capacity) == (int(0.2 * N1), int(0.2 * E1)) == \
        (ora.node_capacity, ora.edge_capacity)
    assert hip.get_mem_size() > before
    for bi in range(4, 10):
        batch(N1, E1, bi)
    with pytest.raises(ValueError):        # the caller must supply tables that cover the new ids
        hip.resize(N1 + 50, E1)
    return hip


def _edge_only_pair(E, cap_ratio, d=4, seed=11):
    import torch
    from gnnflow_amd.cache import LRUCache
    from oracle.cache_oracle import OracleLRUCache
    ef = np.random.RandomState(seed).rand(E, d).astype(np.float32)
    hip = LRUCache(cap_ratio, 0.0, 1, E, "cuda:0", None, torch.from_numpy(ef), 0, d)
    ora = OracleLRUCache(cap_ratio, 0.0, 1, E, None, ef, 0, d, overflow_rule="first_seen")
    hip.init_cache()
    ora.init_cache()
    return hip, ora, ef


def _fetch_both(hip, ora, ef, ids, step):
    ids = np.ascontiguousarray(ids, np.int64)
    none = np.zeros(0, np.int64)
    hb = [[Blk(_to_ids(none), _to_ids(ids))]]
    ob = [[Blk(none, ids)]]
    hip.fetch_feature(hb, None, target_edge_features=False)
    ora.fetch_feature(ob, None)
    assert np.array_equal(_to_np(hb[0][0].edata["f"]), ef[ids]), step
    assert float(hip.cache_edge_ratio) == pytest.approx(ora.cache_edge_ratio, abs=1e-7), step
    got = hip._edge.slot_ids()
    assert np.array_equal(np.sort(got[got >= 0]), ora.edge.cached_ids()), step


def test_dead_entries_right_behind_the_head(lru_form):
    """Queue form: a block that hits ~14000 of the oldest entries but not the very oldest one
    leaves 14000 dead entries right behind the new head; the next, small block needs more
    victims than the three chunks (12288 entries) behind the head hold, so one workgroup has
    to walk on alone.
    (List form: the same sequence, same decisions.)"""
    cap = 65536
    hip, ora, ef = _edge_only_pair(E=4 * cap, cap_ratio=0.25)
    assert hip.edge_capacity == cap
    step = 0
    # initial order = slot order = ids 0..cap-1
    _fetch_both(hip, ora, ef, np.concatenate([np.arange(10, 14000), [cap + 0]]), step); step += 1
    # 50 misses, 50 hits far from the head: victims = entries 1..9 and then beyond the dead run
    _fetch_both(hip, ora, ef, np.concatenate([np.arange(cap + 1, cap + 51),
                                              np.arange(60000, 60050)]), step); step += 1
    st = hip._edge.lru_state()
    assert st["queue_form"] == (1 if lru_form in ("queue", "mixed") else 0)
    if lru_form in ("queue", "mixed"):
        assert st["lone_walks"] == 1 and st["head"] > 14000 and st["tail"] > cap
    rng = np.random.RandomState(5)
    for _ in range(6):
        ids = np.concatenate([rng.randint(0, 4 * cap, 300), rng.randint(14000, 14200, 100)])
        _fetch_both(hip, ora, ef, ids, step); step += 1


def test_many_updates_cross_several_compactions(lru_form):
    """400 updates of ~1000 rows on a 8192-slot cache: in the queue form the tail passes the
    allocation (capacity * 3/2) every few updates and the queue is compacted; blocks of more
    than capacity / 4 rows take the list form in between."""
    cap = 8192
    hip, ora, ef = _edge_only_pair(E=8 * cap, cap_ratio=0.125, seed=12)
    assert hip.edge_capacity == cap
    rng = np.random.RandomState(8)
    for step in range(400):
        n = 3000 if step % 37 == 36 else int(rng.randint(1, 2000))
        lo = int(rng.randint(0, 7 * cap))
        ids = np.concatenate([rng.randint(lo, lo + cap, n), rng.randint(0, 8 * cap, n // 8 + 1)])
        _fetch_both(hip, ora, ef, ids, step)
    st = hip._edge.lru_state()
    if lru_form in ("queue", "mixed"):
        assert st["queue_form"] == 1 and st["queue_entries"] == cap + cap // 2 + 64
        assert st["compactions"] >= 20 and st["list_form_updates"] >= 10
        assert st["tail"] - st["head"] >= cap
    else:
        assert st == dict(queue_form=0, queue_entries=cap, head=0, tail=cap, compactions=0,
                          list_form_updates=0, lone_walks=0)


def test_blocks_without_a_miss_leave_no_marks_behind(lru_form):
    """cache.py:318-323 / lru_cache.py: update only runs for a block that misses, so the hits of
    an all-hit block refresh nothing.  The queue form marks hit entries in a bitmap DURING the
    gather: those marks must be gone before the next block, or it would spare their slots."""
    cap = 4096
    hip, ora, ef = _edge_only_pair(E=8 * cap, cap_ratio=0.125, seed=21)
    rng = np.random.RandomState(9)
    step = 0
    first = rng.permutation(cap)[:3000]
    _fetch_both(hip, ora, ef, first, step); step += 1           # identity-filled cache: all hits
    _fetch_both(hip, ora, ef, np.arange(cap, cap + 900), step); step += 1   # misses: 900 evictions
    for _ in range(3):
        # the most recently installed ids again — every row hits, nothing may move ...
        _fetch_both(hip, ora, ef, np.arange(cap, cap + 900)[rng.permutation(900)[:500]], step); step += 1
        # ... then blocks that miss and evict in the order the oracle's untouched list gives
        _fetch_both(hip, ora, ef, rng.randint(2 * cap, 8 * cap, 700), step); step += 1
        _fetch_both(hip, ora, ef, np.concatenate([rng.randint(0, cap, 300),
                                                  rng.randint(2 * cap, 8 * cap, 300)]), step); step += 1
    if lru_form in ("queue", "mixed"):
        assert hip._edge.lru_state()["queue_form"] == 1


def test_queue_form_bitmap_spanning_several_tiles_and_groups(monkeypatch):
    """A 300 k-slot queue-form cache: its hit bitmap spans several tiles of 131 072 queue positions;
    GNNFLOW_LRU_QUEUE_GROUP=2 makes the install kernel's LDS prefix hold one entry per TWO tiles
    (what caches beyond 89 M slots get), so a hit entry's rank also sums the tiles of its group."""
    monkeypatch.setenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", "1")
    monkeypatch.setenv("GNNFLOW_LRU_QUEUE_GROUP", "2")
    monkeypatch.delenv("GNNFLOW_LRU_FUSED", raising=False)
    E, cap = 1_500_000, 300_000
    hip, ora, ef = _edge_only_pair(E=E, cap_ratio=0.2, seed=33)
    assert hip.edge_capacity == cap
    rng = np.random.RandomState(13)
    for step in range(10):
        lo = cap + step * 9000
        ids = np.concatenate([rng.randint(lo, lo + 60000, 15000),      # recent ids: hits near the tail
                              rng.randint(0, cap, 6000),                # old residents: hits anywhere
                              rng.randint(0, E, 4000)])                 # cold ids
        _fetch_both(hip, ora, ef, ids, step)
    st = hip._edge.lru_state()
    assert st["queue_form"] == 1 and st["tail"] > 2 * 131072 and st["list_form_updates"] == 0

