"""GPU parity of the path bench.py TIMES: gnnflow_amd.pipeline.ReplayPipeline.run — side-stream
sample_async with the library's enqueue thread, record_stream, fetch_feature(async_enqueue) —
on BASELINE config 2 itself (REDDIT-shaped stream, minimum block 62, fanout [10,10] most-recent,
LRUCache 0.2, 172-d node and edge features), against the CPU oracle:

  * every MFG array of every block bit-for-bit (OracleSampler),
  * every `h` / `f` / target-edge row equal to feats[ids] (and to the oracle cache's output),
  * hit counts of every block at every step, cached-id sets (OracleLRUCache),

over the whole chronological replay (1120 consecutive batches, 70 roll-overs of the sampler's
16-call output slab).  Reference loop: scripts/offline_edge_prediction.py:343-346,397-405.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

D = 172
MiB = 1 << 20


class _World:
    """Config 2 on both sides: HIP graph / sampler / cache and their oracles."""

    def __init__(self, first_batch, num_batches, policy="recent", fanouts=(10, 10),
                 cache_ratio=0.2, prefix_alias=True, placement="device", staging=None,
                 dims=(D, D)):
        import torch
        import gnnflow_amd
        from gnnflow_amd import synthetic
        from gnnflow_amd.cache import LRUCache
        from oracle import oracle as O
        from oracle.cache_oracle import OracleLRUCache
        self.torch = torch
        g = synthetic.reddit_like(seed=42)
        N, E = g["num_nodes"], g["num_edges"]
        self.graph = gnnflow_amd.DynamicGraph(20 * MiB, 1000 * MiB, "cuda", 62, 1024, "insert")
        self.ograph = O.OracleGraph(minimum_block_size=62, insertion_policy="insert")
        for lo in range(0, E, 100000):
            hi = lo + 100000
            for gr in (self.graph, self.ograph):
                gr.add_edges(g["src"][lo:hi], g["dst"][lo:hi], g["ts"][lo:hi], g["eid"][lo:hi])
        self.sampler = gnnflow_amd.TemporalSampler(self.graph, list(fanouts), policy, seed=1234)
        self.osampler = O.OracleSampler(self.ograph, list(fanouts), policy, seed=1234, threads=8)
        rng = np.random.RandomState(7)
        dn, de = dims
        self.efeat = rng.rand(E, de).astype(np.float32)
        self.nfeat = rng.rand(N, dn).astype(np.float32)
        dev = torch.device("cuda", 0)
        self.cache = LRUCache(cache_ratio, cache_ratio, N, E, dev, torch.from_numpy(self.nfeat),
                              torch.from_numpy(self.efeat), dn, de, feature_placement=placement,
                              staging=staging)
        self.cache.prefix_alias = prefix_alias
        self.cache.init_cache()
        self.ocache = OracleLRUCache(cache_ratio, cache_ratio, N, E, self.nfeat, self.efeat, dn, de)
        self.ocache.init_cache()
        all_batches = list(synthetic.replay_batches(g, 600, seed=42))
        self.host_batches = all_batches[first_batch:first_batch + num_batches]
        self.dev_batches = [(torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev),
                             torch.from_numpy(e).to(dev)) for r, t, e in self.host_batches]
        self.dev = dev


def _snapshot(cache, mfgs):
    """Device-side copies of everything one step produced (stream-ordered on the current
    stream, no host synchronisation), so the sampler's output slab is free to roll over."""
    blocks = []
    for mfg in mfgs:
        for b in mfg:
            col, row = b.edges()
            rec = dict(nsrc=b.num_src_nodes(), ndst=b.num_dst_nodes(), ne=b.num_edges(),
                       ID=b.srcdata["ID"].clone(), ts=b.srcdata["ts"].clone(),
                       eid=b.edata["ID"].clone(), dt=b.edata["dt"].clone(),
                       col=col.clone(), row=row.clone(),
                       f=b.edata["f"].clone() if b.num_edges() else None,
                       h=b.srcdata["h"].clone() if "h" in b.srcdata else None)
            blocks.append(rec)
    return dict(blocks=blocks, target=cache.target_edge_features.clone(),
                span=cache._stats_span)


def _hits_of(span):
    """Integer hit counts (node blocks, edge blocks) and #aliased blocks from the stats ring."""
    pos, n_node, n_cached, ring, n_alias = span
    rows = ring[pos:pos + n_cached].cpu().numpy().astype(np.int64)
    hits = rows[:, 0::2].sum(axis=1)
    return hits[:n_node], hits[n_node:], rows[:, 1], n_alias


def _check_step(w, bi, snap):
    """Runs the oracle on batch `bi` (advancing its cache) and compares the snapshot."""
    r, t, e = w.host_batches[bi]
    om = w.osampler.sample(r, t)
    oblocks = [b for mfg in om for b in mfg]
    assert len(oblocks) == len(snap["blocks"])
    # oracle cache: block by block in the reference's order, recording per-block hits
    node_hits, edge_hits = [], []
    for b in om[0]:
        out, hits, n = w.ocache.node.fetch(b.srcdata["ID"], True)
        b.srcdata["h"] = out
        node_hits.append(hits)
    for mfg in om:
        for b in mfg:
            if len(b.edata["ID"]):
                out, hits, n = w.ocache.edge.fetch(b.edata["ID"], True)
                b.edata["f"] = out
                edge_hits.append((hits, n))
    for k, (s, ob) in enumerate(zip(snap["blocks"], oblocks)):
        where = "batch {} block {}".format(bi, k)
        assert (s["nsrc"], s["ndst"], s["ne"]) == (ob.num_src_nodes(), ob.num_dst_nodes(),
                                                   ob.num_edges()), where
        assert np.array_equal(s["ID"].cpu().numpy(), ob.srcdata["ID"]), where
        assert np.array_equal(s["ts"].cpu().numpy(), ob.srcdata["ts"]), where
        assert np.array_equal(s["eid"].cpu().numpy(), ob.edata["ID"]), where
        assert np.array_equal(s["dt"].cpu().numpy(), ob.edata["dt"]), where
        ocol, orow = ob.edges()
        assert np.array_equal(s["col"].cpu().numpy(), ocol), where
        assert np.array_equal(s["row"].cpu().numpy(), orow), where
        if s["ne"]:
            f = s["f"].cpu().numpy()
            assert np.array_equal(f, w.efeat[ob.edata["ID"]]), where + " f"
            assert np.array_equal(f, ob.edata["f"]), where + " f vs oracle cache"
        if k < len(om[0]):
            h = s["h"].cpu().numpy()
            assert np.array_equal(h, w.nfeat[ob.srcdata["ID"]]), where + " h"
        else:
            assert s["h"] is None
    assert np.array_equal(snap["target"].cpu().numpy(), w.efeat[e]), "batch {} target".format(bi)
    hn, he, counts, n_alias = _hits_of(snap["span"])
    assert list(hn) == node_hits, "batch {} node hits".format(bi)
    # blocks served as a prefix of the previous block's rows are all hits by construction
    got = [(int(h), int(c)) for h, c in zip(he, counts[len(hn):])]
    assert got == edge_hits[:len(got)], "batch {} edge hits".format(bi)
    assert len(got) + n_alias == len(edge_hits)
    for hits, n in edge_hits[len(got):]:
        assert hits == n, "batch {}: aliased block is not all hits in the oracle".format(bi)


def _cached_sets_equal(w):
    assert np.array_equal(np.sort(w.cache._edge.slot_ids()), w.ocache.edge.cached_ids())
    assert np.array_equal(np.sort(w.cache._node.slot_ids()), w.ocache.node.cached_ids())


@pytest.mark.parametrize("first_batch", [0, 700])
def test_pipelined_replay_bit_exact_on_config2(first_batch):
    """The free-running pipeline (exactly bench.py's loop): step i's results are copied on
    the device while step i+1 is already in flight; compared in chunks of 32 steps."""
    from gnnflow_amd.pipeline import ReplayPipeline
    # from batch 0: the WHOLE chronological replay (1120 of its 1121 batches, 70 roll-overs of
    # the 16-call output slab), i.e. bench.py's timed window; from batch 700: 224 late batches
    steps, chunk = int(os.environ.get("GNNFLOW_PARITY_STEPS", 1120 if first_batch == 0 else 224)), 32
    steps -= steps % chunk
    w = _World(first_batch, steps)
    pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=True)
    assert pipe.pipelined
    for c0 in range(0, steps, chunk):
        snaps, held = [], []

        def on_step(i, mfgs):
            # snapshot the PREVIOUS step: its fetch was enqueued long ago, so no extra host
            # wait is introduced into the loop being tested
            if held:
                pi, pm = held.pop()
                snaps.append((pi, _snapshot(w.cache_prev, pm)))
            held.append((i, mfgs))
            w.cache_prev = _StepView(w.cache)

        pipe.run(c0, chunk, on_step)
        pi, pm = held.pop()
        snaps.append((pi, _snapshot(w.cache_prev, pm)))
        w.torch.cuda.synchronize()
        assert [i for i, _ in snaps] == list(range(c0, c0 + chunk))
        for i, snap in snaps:
            _check_step(w, i, snap)
        _cached_sets_equal(w)
    if first_batch == 700 and steps <= 400:
        # late batches: the roots' layer really is a prefix and really was aliased
        assert w.cache.prefix_alias and snaps[-1][1]["span"][4] == 1


class _StepView:
    """What Cache exposes about the fetch_feature() call that was just issued, frozen so it
    can be read one step later (the next call replaces these attributes)."""

    def __init__(self, cache):
        self._cache = cache
        self._thunk = cache._target_edge_thunk
        self._value = cache._target_edge_features
        self._stats_span = cache._stats_span

    @property
    def target_edge_features(self):
        if self._value is None:
            self._value = self._thunk()
        return self._value


@pytest.mark.parametrize("lru_form", ["list", "list2", "queue", "mixed"])
def test_stepwise_cached_sets_and_alias_equivalence(lru_form, monkeypatch):
    """Same loop with a host check after every step (cached-id sets at EVERY step), once with
    the prefix alias and once without: both must agree with the oracle, hence with each
    other.  Also with the LRU order kept as a queue (the form of caches of >= 2 M slots,
    forced here): the edge cache then appends, the node cache — blocks larger than a quarter
    of its 2196 slots — goes through the list form and re-indexes, in the same launches.
    "mixed": the bound between the two caches' sizes — the edge cache a queue, the node cache a
    list updated in ONE launch on the round's side stream beside the queue-form launches (what a
    GDELT-scale step does)."""
    from gnnflow_amd.pipeline import ReplayPipeline
    if lru_form == "queue":
        monkeypatch.setenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", "1")
    elif lru_form == "mixed":
        monkeypatch.setenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", "10000")
    else:
        monkeypatch.delenv("GNNFLOW_LRU_QUEUE_MIN_CAPACITY", raising=False)
    if lru_form == "list2":     # list scan + list install instead of the one-launch update
        monkeypatch.setenv("GNNFLOW_LRU_FUSED", "0")
    else:
        monkeypatch.delenv("GNNFLOW_LRU_FUSED", raising=False)
    for alias in (True, False):
        w = _World(900, 48, prefix_alias=alias)
        pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=True)

        def on_step(i, mfgs):
            snap = _snapshot(w.cache, mfgs)
            w.torch.cuda.synchronize()
            _check_step(w, i, snap)
            _cached_sets_equal(w)
            assert snap["span"][4] == (1 if alias else 0)

        pipe.run(0, 48, on_step)
        if lru_form == "mixed":
            assert w.cache._edge.lru_state()["queue_form"] == 1
            assert w.cache._node.lru_state()["queue_form"] == 0


@pytest.mark.parametrize("fanouts,policy", [((10, 10, 10), "recent"), ((10, 5), "recent"),
                                            ((5, 10), "recent"), ((10, 10), "uniform")])
def test_prefix_property_only_where_it_holds(fanouts, policy):
    """3 equal layers alias twice; unequal fanouts and uniform sampling never alias; rows,
    hits and cached sets match the oracle in every case (small cache: blocks larger than
    the capacity must fall back to a real lookup)."""
    from gnnflow_amd.pipeline import ReplayPipeline
    for ratio in (0.2, 0.002):
        w = _World(1000, 6, policy=policy, fanouts=fanouts, cache_ratio=ratio)
        if policy == "uniform":
            # uniform draws differ from the oracle's only in nothing: same Philox stream
            pass
        pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=False)
        seen = []

        def on_step(i, mfgs):
            snap = _snapshot(w.cache, mfgs)
            w.torch.cuda.synchronize()
            _check_step(w, i, snap)
            _cached_sets_equal(w)
            seen.append(snap["span"][4])

        pipe.run(0, 6, on_step)
        if policy == "uniform" or fanouts in ((10, 5), (5, 10)):
            assert set(seen) == {0}
        elif ratio == 0.2:
            assert set(seen) == {len(fanouts) - 1}


def _run_chunks(w, pipe, steps, chunk):
    """bench.py's free-running loop in chunks, every step's snapshot checked against the oracle."""
    for c0 in range(0, steps, chunk):
        snaps, held = [], []

        def on_step(i, mfgs):
            if held:
                pi, pm = held.pop()
                snaps.append((pi, _snapshot(w.cache_prev, pm)))
            held.append((i, mfgs))
            w.cache_prev = _StepView(w.cache)

        pipe.run(c0, chunk, on_step)
        pi, pm = held.pop()
        snaps.append((pi, _snapshot(w.cache_prev, pm)))
        w.torch.cuda.synchronize()
        assert [i for i, _ in snaps] == list(range(c0, c0 + chunk))
        for i, snap in snaps:
            _check_step(w, i, snap)
        _cached_sets_equal(w)


@pytest.mark.parametrize("staging", ["auto", (8, 64), None])
def test_pinned_tables_pipelined_replay_bit_exact(staging):
    """The reference's feature placement — tables in (pinned) host memory, LRU 0.2 on the GPU
    (gnnflow/cache/cache.py:288-313,381-388) — through the pipelined loop: batch i+1's coming
    misses are pulled into the staging ring on a side stream while batch i is fetched.  Rows,
    hit counts and cached-id sets must equal the oracle's whatever the ring does: "auto" (the
    default), a ring of 8 x 64 rows (nearly every generation overflows: most missed rows fall
    back to the host table inside the gather), and no ring at all."""
    from gnnflow_amd.pipeline import ReplayPipeline
    steps, chunk = int(os.environ.get("GNNFLOW_PARITY_STEPS_PINNED", 320)), 32
    steps -= steps % chunk
    for first_batch in (0, 800):
        w = _World(first_batch, steps, placement="pinned", staging=staging or 0)
        assert w.cache.staging == (staging is not None)
        pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=True)
        _run_chunks(w, pipe, steps, chunk)
        if staging is not None:
            st = w.cache.staging_state()
            for kind in ("node", "edge"):
                # (an "auto" ring that grew with the blocks started counting again)
                assert st[kind]["issued"] + st[kind]["dropped"] <= steps, st
                assert st[kind]["issued"] >= steps // 2, st     # the loop keeps up with its ring
                assert st[kind]["rows_pulled"] > 0, st
            if staging == "auto":
                # the batch's own 600 target edges are pulled once and serve the next batches'
                # edge misses: far fewer rows cross the link than the fetches missed
                assert st["edge"]["rows_pulled"] < 1300 * steps, st
                print("staging state after", steps, "steps from batch", first_batch, st)
            else:
                assert st["edge"]["rows_per_generation"] == 64


@pytest.mark.parametrize("dims", [(13, 186), (3, 5)])
def test_pinned_tables_odd_row_widths(dims):
    """The same with row widths that are no multiple of four floats (GDELT's edge rows are 186-d,
    gnnflow/config.py:157-167): the pull moves them as 16-byte vectors at 4-byte alignment, the last
    one ending with the row; rows under four floats go float by float."""
    from gnnflow_amd.pipeline import ReplayPipeline
    steps, chunk = 96, 32
    w = _World(850, steps, placement="pinned", staging="auto", dims=dims)
    pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=True)
    _run_chunks(w, pipe, steps, chunk)
    st = w.cache.staging_state()
    assert st["edge"]["rows_pulled"] > 0 and st["node"]["rows_pulled"] > 0, st


def test_pinned_tables_prefetch_is_only_a_hint(monkeypatch):
    """prefetch_feature() by hand around the plain loop: announcing a batch twice, never, for
    another batch's MFGs, with generations that are dropped at once (GNNFLOW_STAGE_SPIN_US=0 and
    no fetch in between), or after invalidate_staging() changes nothing in what is fetched."""
    monkeypatch.setenv("GNNFLOW_STAGE_SPIN_US", "0")
    w = _World(900, 24, placement="pinned", staging=(8, 4096))
    samp, cache = w.sampler, w.cache
    mf = [None] * 24

    def sample(i):
        if mf[i] is None:
            r, t, _ = w.dev_batches[i]
            mf[i] = samp.sample(r, t)
        return mf[i]

    for i in range(24):
        mfgs = sample(i)
        e = w.dev_batches[i][2]
        if i % 4 == 0:
            assert cache.prefetch_feature(mfgs, e)
            cache.prefetch_feature(mfgs, e)                     # twice
        elif i % 4 == 3:
            # announced two batches ahead, fetched with a lag of one (what the pipelined loop does)
            if i + 1 < 24:
                cache.prefetch_feature(mfgs, e)
                cache.prefetch_feature(sample(i + 1), w.dev_batches[i + 1][2])
                cache.set_staging_lag(1)
        elif i % 4 == 1 and i + 1 < 24:
            cache.prefetch_feature(sample(i + 1), w.dev_batches[i + 1][2])   # someone else's
        elif i % 4 == 2:
            for _ in range(12):                                 # more generations than regions
                cache.prefetch_feature(mfgs, e)
        if i == 13:
            cache.invalidate_staging()
        cache.fetch_feature(mfgs, e)
        cache.set_staging_lag(0)
        snap = _snapshot(cache, mfgs)
        w.torch.cuda.synchronize()
        _check_step(w, i, snap)
        _cached_sets_equal(w)
    st = cache.staging_state()
    assert st["edge"]["dropped"] > 0 and st["edge"]["issued"] > 0, st
    # a cache over device-resident tables has no ring and says so
    w2 = _World(900, 1)
    assert not w2.cache.staging
    assert w2.cache.prefetch_feature(w2.sampler.sample(*w2.dev_batches[0][:2]),
                                     w2.dev_batches[0][2]) is False


@pytest.mark.parametrize("fanouts", [(10, 10), (5, 10, 3)])
def test_uniform_sampling_on_two_lanes_reproduces_the_one_samplers_draws(fanouts):
    """Uniform sampling through the pipelined loop: consecutive batches go to two samplers (the
    sampler and a clone, a stream each), every sample begun with the call number it would have had
    on the one sampler — the MFGs must equal the oracle's, whose draws come from ONE sampler's
    counter (philox keyed by (seed, slot, call); sampling_kernels.cu:109-273 draws from curand
    states instead).  Then the plain loop continues on the primary sampler: its counter must stand
    where the oracle's does."""
    from gnnflow_amd.pipeline import ReplayPipeline
    w = _World(1000, 40, policy="uniform", fanouts=fanouts)
    pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=True)
    assert pipe.pipelined and len(pipe.lanes) == 2
    _run_chunks(w, pipe, 32, 16)
    assert w.sampler.call_counter() == 32 * len(fanouts)
    # the sampler alone, pipelined without a cache (config 3's loop): same stream of draws
    got = []
    pipe2 = ReplayPipeline(w.sampler, None, w.dev_batches, w.dev, pipelined=True)
    assert len(pipe2.lanes) == 2
    pipe2.run(32, 4, lambda i, mfgs: got.append((i, [[(b.edata["ID"].cpu().numpy(),
                                                        b.srcdata["ID"].cpu().numpy())
                                                       for b in mfg] for mfg in mfgs])))
    for i, mf in got:
        r, t, _ = w.host_batches[i]
        om = w.osampler.sample(r, t)
        for a, b in zip(mf, om):
            for (eid, nid), ob in zip(a, b):
                assert np.array_equal(eid, ob.edata["ID"]) and np.array_equal(nid, ob.srcdata["ID"])
    # ... and back on the primary sampler through the plain call
    for i in range(36, 40):
        r, t, _ = w.dev_batches[i]
        hm = w.sampler.sample(r, t)
        om = w.osampler.sample(*w.host_batches[i][:2])
        for a, b in zip(hm, om):
            for hb, ob in zip(a, b):
                assert np.array_equal(hb.edata["ID"].cpu().numpy(), ob.edata["ID"])


def test_fused_lru_update_fallback_paths():
    """The one-launch list update never waits for ever: a look-back that has not seen its granule
    recomputes the value from the launch's inputs.  With the polls' budget forced to 0
    (GNNFLOW_LRU_FUSE_SPINS, read when a process creates its first cache — hence a child process)
    EVERY wait takes that path; rows, hit counts and cached-id sets must still equal the oracle's
    at every step, and the recount counter must show that the path ran."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GNNFLOW_LRU_FUSE_SPINS="0")
    for k in ("GNNFLOW_LRU_FUSED", "GNNFLOW_LRU_QUEUE_MIN_CAPACITY"):   # (this module's arrangements)
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(root, "tests", "lru_fallback_child.py")],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    last = [l for l in p.stdout.splitlines() if l.startswith("recounts")][-1]
    assert int(last.split()[1]) > 100, last


@pytest.mark.parametrize("mirror", ["0", "1"])
def test_device_tables_with_and_without_the_row_mirror(mirror, monkeypatch):
    """With the tables in HBM the cache keeps no second copy of the cached rows by default
    (gf_cache_set_row_mirror: slots hold ids, every row is read from the table, an install moves
    nothing); GNNFLOW_CACHE_ROW_MIRROR=1 keeps the reference's buffer (cache.py:84-99).  Rows, hit
    counts and cached-id sets are the oracle's either way; only the memory differs."""
    from gnnflow_amd.pipeline import ReplayPipeline
    monkeypatch.setenv("GNNFLOW_CACHE_ROW_MIRROR", mirror)
    w = _World(900, 64)
    rows = w.cache.edge_capacity * D * 4 + w.cache.node_capacity * D * 4
    size = w.cache.get_mem_size()
    index = 4 * (w.cache.num_nodes + w.cache.num_edges) + 16 * (w.cache.edge_capacity + w.cache.node_capacity)
    assert size == index + (rows if mirror == "1" else 0)
    pipe = ReplayPipeline(w.sampler, w.cache, w.dev_batches, w.dev, pipelined=True)
    _run_chunks(w, pipe, 64, 32)
